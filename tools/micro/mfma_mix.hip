// mfma_mix.hip -- the budget VERDICT r4 (next #3 i) asked for: what the matrix pipe sustains under the socket power cap as a function of the
// instruction mix around it -- R `ds_read_b128` per MFMA (fresh random operands out of a 64-KB LDS image) x V independent fp32 VALU instructions per MFMA --
// at the occupancy of the large convolutions (four-wave workgroups, two or three per CU).  8 MFMAs (`v_mfma_f32_32x32x16_f16`) per group on 8 accumulators,
// random operand data (zero data clocks higher: MI355X_MICROARCH.md), no barriers, no global traffic: an upper bound for ANY kernel of that mix.
// Second version (round 5, last session).  The first one wrote its VALU filler as C++ `fmaf` chains and its fragment addresses as `(idx + 67) & mask`: hipcc
// packed the chains into `v_pk_fma_f32` pairs (V = 4 came out as 2 VALU instructions per MFMA) and spent three address instructions per `ds_read`
// (2.25 per MFMA at R = 0.75), so the VALU axis of its table was not what its header said.  Here the filler is `v_fma_f32` in inline asm (counted: the
// harness checks nothing, the ISA does -- `hipcc -S` shows V * 8 + 2 VALU per group) and the fragment reads use immediate offsets from ONE base address
// that moves once per group (2 VALU per 8 MFMAs).
// Two more axes at fixed mixes: the MFMA SHAPE / operand KIND (32x32x16 f16 against two 16x16x32 f16 per unit of flops -- twice the operand registers,
// half the accumulator registers moved per flop -- and 32x32x16 bf16) and the DATA (a fraction of zero elements in the B (pixel) operand, as behind a ReLU).
//   hipcc --offload-arch=gfx950 -O3 -o build_variants/mfma_mix tools/micro/mfma_mix.hip && build_variants/mfma_mix [seconds per point]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

enum { S32_F16 = 0, S16_F16 = 1, S32_BF16 = 2 };

// R8 = ds_read_b128 per 8 units (a unit = 32768 flops = one 32x32x16 MFMA or two 16x16x32 MFMAs); V = VALU per unit; LDS_KB caps the workgroups per CU
// (64 -> 2, 48 -> 3); ZB = zero elements of every 8 in the B operand pieces
template <int SHAPE, int R8, int V, int LDS_KB, int ZB>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned seed) {
  __shared__ uint4 lds[LDS_KB * 64];
  const int tid = threadIdx.x;
  unsigned s = seed ^ (tid * 2654435761u) ^ (blockIdx.x * 40503u);
  // pairs in [0.5, 1): f16 exponent 0x38, bf16 exponent 0x3f0
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return SHAPE == S32_BF16 ? (((s >> 9) | 0x3f003f00u) & 0x3f7f3f7fu) : (((s >> 9) | 0x38003800u) & 0x3bff3bffu); };
  auto rnd_b = [&]() {                                  // a B piece: ZB of its 8 elements zero, positions drawn per piece
    uint4 v = make_uint4(rnd(), rnd(), rnd(), rnd());
    unsigned m = 0, need = ZB;
    while (need) { s = s * 1664525u + 1013904223u; const unsigned b = (s >> 13) & 7u; if (!((m >> b) & 1u)) { m |= 1u << b; --need; } }
    unsigned* w = reinterpret_cast<unsigned*>(&v);
    for (int e = 0; e < 8; ++e) if ((m >> e) & 1u) w[e >> 1] &= (e & 1) ? 0x0000FFFFu : 0xFFFF0000u;
    return v;
  };
  for (int i = tid; i < LDS_KB * 64; i += 256) lds[i] = (i & 1) ? rnd_b() : make_uint4(rnd(), rnd(), rnd(), rnd());      // odd pieces: B operands
  __syncthreads();
  constexpr int NF = R8 < 2 ? 1 : (R8 / 2 > 6 ? 6 : R8 / 2);          // fragments of each operand refreshed per group
  uint4 fa[6], fb[6];
  for (int i = 0; i < 6; ++i) { fa[i] = make_uint4(rnd(), rnd(), rnd(), rnd()); fb[i] = rnd_b(); }
  f32x16_t acc[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float v[4] = {1.f, 2.f, 3.f, 4.f};
  const float c1 = 0.9999f, c2 = 1e-4f;
  // fragment reads: immediate offsets (multiples of 1952 B = 122 pieces: A pieces even, B pieces odd through the +16) from one base in the lower half of the image
  constexpr unsigned HALF_MASK = (LDS_KB * 1024 / 2 - 1) & ~31u;
  static_assert(12 * 1952 + 32 <= LDS_KB * 1024 / 2, "offsets stay inside the upper half");
  unsigned base = (unsigned)(tid * 32) & HALF_MASK;
  const char* lb = reinterpret_cast<const char*>(lds);
  for (int it = 0; it < iters; ++it) {
    base = (base + 4128u) & HALF_MASK;                                     // 2 VALU per group
#pragma unroll
    for (int r = 0; r < R8 / 2; ++r) {
      fa[r] = *reinterpret_cast<const uint4*>(lb + base + (2 * r) * 1952);
      fb[r] = *reinterpret_cast<const uint4*>(lb + base + (2 * r + 1) * 1952 + 16);
    }
    if constexpr (R8 & 1) fa[NF - 1] = *reinterpret_cast<const uint4*>(lb + base + (R8 - 1) * 1952);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int q = 0; q < V; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(u + q) & 3]) : "v"(c1), "v"(c2));     // independent chains: issue slots, not latency
      const uint4 a = fa[u % NF], b = fb[(u * 5 + 3) % NF];
      if constexpr (SHAPE == S32_F16) {
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc[u], 0, 0, 0);
      } else if constexpr (SHAPE == S32_BF16) {
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[u], 0, 0, 0);
      } else {
        // two 16x16x32 instructions on two quads of the same accumulator block (independent chains, as the sixteen registers of the 32x32 form)
        f32x4_t q0 = {acc[u][0], acc[u][1], acc[u][2], acc[u][3]}, q1 = {acc[u][4], acc[u][5], acc[u][6], acc[u][7]};
        q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), q0, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), q1, 0, 0, 0);
        acc[u][0] = q0[0]; acc[u][1] = q0[1]; acc[u][2] = q0[2]; acc[u][3] = q0[3];
        acc[u][4] = q1[0]; acc[u][5] = q1[1]; acc[u][6] = q1[2]; acc[u][7] = q1[3];
      }
    }
    if ((it & 63) == 63) for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] *= 1e-30f;
  }
  float r = v[0] + v[1] + v[2] + v[3];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
  if (r == 123.456f) out[0] = r;
}

template <int SHAPE, int R8, int V, int LDS_KB, int ZB> static double run(float* out, double secs) {
  const int iters = 4000, blocks = 256 * (LDS_KB == 64 ? 2 : 3) * 4;
  hipLaunchKernelGGL((k<SHAPE, R8, V, LDS_KB, ZB>), dim3(blocks), dim3(256), 0, 0, out, iters, 1234u);
  (void)hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  int n = 0; double el = 0;
  while (el < secs) { hipLaunchKernelGGL((k<SHAPE, R8, V, LDS_KB, ZB>), dim3(blocks), dim3(256), 0, 0, out, iters, 1234u); (void)hipDeviceSynchronize(); ++n;
                      el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  return (double)n * blocks * 4 * iters * 8.0 * 32768.0 / el * 1e-12;
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 1.0;
  float* out; (void)hipMalloc(&out, 4);
  printf("f16 MFMA 32x32x16, random operands, 4-wave workgroups; TFLOP/s (fraction of 2500); VALU = counted v_fma_f32 per MFMA (+ 0.25 per MFMA of addressing)\n");
  printf("%-38s %14s %14s %14s %14s %14s\n", "ds_read_b128 per MFMA \\ VALU per MFMA", "0", "2", "3", "4", "6");
#define ROW(R8, KB, label) { double z = run<S32_F16, R8, 0, KB, 0>(out, secs), a = run<S32_F16, R8, 2, KB, 0>(out, secs), b = run<S32_F16, R8, 3, KB, 0>(out, secs), \
                                    c = run<S32_F16, R8, 4, KB, 0>(out, secs), d = run<S32_F16, R8, 6, KB, 0>(out, secs); \
    printf("%-38s %8.0f (%.2f) %8.0f (%.2f) %8.0f (%.2f) %8.0f (%.2f) %8.0f (%.2f)\n", label, z, z / 2500, a, a / 2500, b, b / 2500, c, c / 2500, d, d / 2500); fflush(stdout); }
  ROW(0, 64, "0     (registers), 2 WG/CU")
  ROW(4, 64, "0.5,  2 WG/CU")
  ROW(6, 64, "0.75, 2 WG/CU (conv2 today)")
  ROW(8, 64, "1.0,  2 WG/CU")
  ROW(12, 64, "1.5,  2 WG/CU")
  ROW(8, 48, "1.0,  3 WG/CU (conv3 8x32 today)")
  printf("\nshape / kind / data at three mixes (ds_read_b128 / VALU per 32768 flops), 2 WG/CU\n%-52s %14s %14s %14s\n", "", "0 / 2", "0.75 / 4", "1.0 / 4");
#define ROW2(SH, ZB, label) { double a = run<SH, 0, 2, 64, ZB>(out, secs), b = run<SH, 6, 4, 64, ZB>(out, secs), c = run<SH, 8, 4, 64, ZB>(out, secs); \
    printf("%-52s %8.0f (%.2f) %8.0f (%.2f) %8.0f (%.2f)\n", label, a, a / 2500, b, b / 2500, c, c / 2500); fflush(stdout); }
  ROW2(S32_F16, 0, "f16 32x32x16, dense random operands")
  ROW2(S16_F16, 0, "f16 16x16x32 x2, dense random operands")
  ROW2(S32_BF16, 0, "bf16 32x32x16, dense random operands")
  ROW2(S32_F16, 4, "f16 32x32x16, half of the B elements zero")
  ROW2(S16_F16, 4, "f16 16x16x32 x2, half of the B elements zero")
  ROW2(S32_F16, 6, "f16 32x32x16, 6 of 8 B elements zero")
  ROW2(S32_F16, 0, "f16 32x32x16, dense (again: drift check)")
  return 0;
}
