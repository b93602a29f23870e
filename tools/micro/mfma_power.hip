// mfma_power.hip -- what does the matrix pipe sustain under the socket power cap with non-trivial operand data?
//   mode 0: MFMA only (operands stay in registers)            mode 1: + one ds_read_b128 per MFMA (fresh operands from LDS)
//   mode 2: as 1 plus ~6 VALU per MFMA (the mix of the conv kernels)
//   mode 3: the mix a Winograd F(2x2,3x3) inner loop would have: 1 ds_read_b128 and ~20 VALU per MFMA,
//           one ds_write_b128 per 2 MFMAs (the transformed input written to LDS)
// usage: mfma_power <mode> <seconds> <zero-data 0|1> <waves per workgroup: 4|8>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters, unsigned seed, int zero) {
  __shared__ uint4 lds[4096];                 // 64 KB
  const int tid = threadIdx.x;
  unsigned s = seed ^ (tid * 2654435761u) ^ (blockIdx.x * 40503u);
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return zero ? 0u : ((s >> 9) | 0x3c003c00u) & 0x3fff3fffu; };   // bf16 pairs in [0.5, 2)
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = make_uint4(rnd(), rnd(), rnd(), rnd());
  __syncthreads();
  uint4 a[2], b[2];
  for (int i = 0; i < 2; ++i) { a[i] = make_uint4(rnd(), rnd(), rnd(), rnd()); b[i] = make_uint4(rnd(), rnd(), rnd(), rnd()); }
  f32x16_t acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float v0 = 1.f, v1 = 2.f, v2 = 3.f;
  int idx = tid;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE >= 1) {
        idx = (idx + 67) & 4095;
        const uint4 f = lds[idx];
        if (u & 1) a[u >> 1] = f; else b[u >> 1] = f;
      }
      if (MODE >= 2) {
        v0 = fmaf(v0, 1.0001f, v1); v1 = fmaxf(v1 * 0.999f, v2); v2 = v2 + v0 * 1e-9f;
        v0 = fminf(v0, 3.f); v1 = fmaf(v1, 0.5f, 0.25f); v2 = fminf(v2, 5.f);
      }
      if (MODE >= 3) {
#pragma unroll
        for (int r = 0; r < 7; ++r) { v0 = v0 + v1; v1 = v1 - v2; }          // 14 more adds: the transform's additions
        if (u & 1) lds[(idx + 2048) & 4095] = make_uint4(__float_as_uint(v0), __float_as_uint(v1), a[0].z, b[0].w);
      }
      acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[u & 1]), __builtin_bit_cast(bf16x8_t, b[u >> 1]), acc[u], 0, 0, 0);
    }
    if ((it & 63) == 63) for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] *= 1e-30f;   // keep values finite
  }
  float r = v0 + v1 + v2;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
  if (r == 123.456f) out[0] = r;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double secs = argc > 2 ? atof(argv[2]) : 3.0;
  const int zero = argc > 3 ? atoi(argv[3]) : 0;
  const int waves = argc > 4 ? atoi(argv[4]) : 8;
  float* out; hipMalloc(&out, 4);
  const int iters = 20000, blocks = 256 * (waves == 8 ? 1 : 2) * 4;
  auto launch = [&]() {
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(waves * 64), 0, 0, out, iters, 1234u, zero);
    else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(waves * 64), 0, 0, out, iters, 1234u, zero);
    else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(waves * 64), 0, 0, out, iters, 1234u, zero);
    else hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(waves * 64), 0, 0, out, iters, 1234u, zero);
  };
  launch(); hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  int n = 0; double el = 0;
  while (el < secs) { launch(); hipDeviceSynchronize(); ++n; el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  const double flops = (double)n * blocks * waves * iters * 4.0 * 32768.0;
  printf("mode %d zero %d waves/wg %d: %.1f TFLOP/s bf16 (%d launches in %.2f s)\n", mode, zero, waves, flops / el * 1e-12, n, el);
  return 0;
}
