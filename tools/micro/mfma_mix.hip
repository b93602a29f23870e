// mfma_mix.hip -- the budget VERDICT r4 (next #3 i) asked for: what the f16 matrix pipe sustains under the socket power cap as a function of the
// instruction mix around it -- R `ds_read_b128` per MFMA (fresh random operands out of a 64-KB LDS image) x V dependent-free fp32 VALU instructions per MFMA --
// at the occupancy of the large convolutions (four-wave workgroups, two or three per CU).  `v_mfma_f32_32x32x16_f16`, 8 MFMAs per group on 8 accumulators,
// random operand data (zero data clocks higher: MI355X_MICROARCH.md), no barriers, no global traffic: an upper bound for ANY kernel of that mix.
//   hipcc --offload-arch=gfx950 -O3 -o build_variants/mfma_mix tools/micro/mfma_mix.hip && build_variants/mfma_mix [seconds per point]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int R8, int V, int LDS_KB>      // R8 = ds_read_b128 per 8 MFMAs; V = VALU per MFMA; LDS_KB caps the workgroups per CU (64 -> 2, 48 -> 3)
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned seed) {
  __shared__ uint4 lds[LDS_KB * 64];
  const int tid = threadIdx.x;
  unsigned s = seed ^ (tid * 2654435761u) ^ (blockIdx.x * 40503u);
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 9) | 0x38003800u) & 0x3bff3bffu; };      // f16 pairs in [0.5, 1)
  for (int i = tid; i < LDS_KB * 64; i += 256) lds[i] = make_uint4(rnd(), rnd(), rnd(), rnd());
  __syncthreads();
  uint4 fr[12];
  for (int i = 0; i < 12; ++i) fr[i] = make_uint4(rnd(), rnd(), rnd(), rnd());
  f32x16_t acc[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float v[4] = {1.f, 2.f, 3.f, 4.f};
  int idx = tid;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < R8; ++r) { idx = (idx + 67) & (LDS_KB * 64 - 1); fr[r] = lds[idx]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int q = 0; q < V; ++q) v[(u + q) & 3] = fmaf(v[(u + q) & 3], 0.9999f, 1e-4f);        // independent chains: issue slots, not latency
      acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, fr[u % (R8 < 2 ? 2 : (R8 > 12 ? 12 : R8))]),
                                                       __builtin_bit_cast(f16x8_t, fr[(u * 5 + 3) % (R8 < 2 ? 2 : (R8 > 12 ? 12 : R8))]), acc[u], 0, 0, 0);
    }
    if ((it & 63) == 63) for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] *= 1e-30f;
  }
  float r = v[0] + v[1] + v[2] + v[3];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
  if (r == 123.456f) out[0] = r;
}

template <int R8, int V, int LDS_KB> static double run(float* out, double secs) {
  const int iters = 4000, blocks = 256 * (LDS_KB == 64 ? 2 : 3) * 4;
  hipLaunchKernelGGL((k<R8, V, LDS_KB>), dim3(blocks), dim3(256), 0, 0, out, iters, 1234u);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  int n = 0; double el = 0;
  while (el < secs) { hipLaunchKernelGGL((k<R8, V, LDS_KB>), dim3(blocks), dim3(256), 0, 0, out, iters, 1234u); hipDeviceSynchronize(); ++n;
                      el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  return (double)n * blocks * 4 * iters * 8.0 * 32768.0 / el * 1e-12;
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 1.0;
  float* out; hipMalloc(&out, 4);
  printf("f16 MFMA 32x32x16, random operands, 4-wave workgroups; TFLOP/s (fraction of 2500)\n");
  printf("%-34s %14s %14s %14s %14s\n", "ds_read_b128 per MFMA \\ VALU per MFMA", "2", "3", "4", "6");
#define ROW(R8, KB, label) { double a = run<R8, 2, KB>(out, secs), b = run<R8, 3, KB>(out, secs), c = run<R8, 4, KB>(out, secs), d = run<R8, 6, KB>(out, secs); \
    printf("%-34s %8.0f (%.2f) %8.0f (%.2f) %8.0f (%.2f) %8.0f (%.2f)\n", label, a, a / 2500, b, b / 2500, c, c / 2500, d, d / 2500); fflush(stdout); }
  ROW(0, 64, "0     (registers), 2 WG/CU")
  ROW(4, 64, "0.5,  2 WG/CU")
  ROW(6, 64, "0.75, 2 WG/CU (conv2 today)")
  ROW(8, 64, "1.0,  2 WG/CU")
  ROW(12, 64, "1.5,  2 WG/CU")
  ROW(4, 48, "0.5,  3 WG/CU")
  ROW(8, 48, "1.0,  3 WG/CU (conv3 8x32 today)")
  ROW(12, 48, "1.5,  3 WG/CU")
  return 0;
}
