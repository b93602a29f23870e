// lds_fill_rate.hip -- how fast can a CU fill LDS from an L2-resident image, by LDS-DMA (global_load_lds_dwordx4, the kernels' weight path) and by
// register staging (global_load_dwordx4 -> ds_write_b128)?  Round 6: the timing ablation puts the in-loop weight DMA at a quarter of conv2 / conv3
// (profiles/r06_experiments.md section 3) while a lean issue sequence buys nothing -- is the LDS-DMA return path itself the limit?
// Persistent four-wave workgroups, two per CU (78 KB of LDS each), a 295-KB image shared by all (the packed weights of conv2 / conv3); per round every
// wave fetches P pieces of 1 KiB into a ring slot, waits for them, and the workgroup meets at a barrier (the kernels' stage structure without MFMAs).
//   hipcc --offload-arch=gfx950 -O3 -I diffusiondepth_amd/csrc -o build_variants/lds_fill_rate tools/micro/lds_fill_rate.hip && build_variants/lds_fill_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dd_gcn.h"

constexpr int IMG = 294912;

template <int MODE, int P>      // MODE 0 = LDS-DMA, 1 = registers + ds_write_b128, 2 = loads only (into registers, xor-folded), 3 = LDS-DMA with 2 x P pieces in flight (two slots ahead)
__global__ void __launch_bounds__(256, 2) fill(const char* __restrict__ img, float* out, int rounds) {
  DD_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = DD_LDS_BASE(smem);
  constexpr int STAGE = 4 * P * 1024;            // bytes per round and workgroup
  uint4 fold = make_uint4(0u, 0u, 0u, 0u);
  int off = (blockIdx.x * 7919) % (IMG / STAGE) * STAGE;
#pragma unroll 1
  for (int r = 0; r < rounds; ++r) {
    const int slot = r & 1;
    if constexpr (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int c = 0; c < P; ++c) {
        const int piece = c * 4 + wave;
        const char* src = img + off + piece * 1024 + lane * 16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot * STAGE + piece * 1024);
        DD_LDS_DMA16(smem, src, dst);
      }
      if constexpr (MODE == 3) { if (r > 0) DD_WAIT_VM(P); } else DD_WAIT_VM(0);
    } else {
      uint4 v[P];
#pragma unroll
      for (int c = 0; c < P; ++c) v[c] = *reinterpret_cast<const uint4*>(img + off + (c * 4 + wave) * 1024 + lane * 16);
      if constexpr (MODE == 1) {
#pragma unroll
        for (int c = 0; c < P; ++c) *reinterpret_cast<uint4*>(smem + slot * STAGE + (c * 4 + wave) * 1024 + lane * 16) = v[c];
      } else {
#pragma unroll
        for (int c = 0; c < P; ++c) { fold.x ^= v[c].x; fold.y ^= v[c].y; fold.z ^= v[c].z; fold.w ^= v[c].w; }
      }
      DD_WAIT_LGKM0();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    off += STAGE;
    if (off + STAGE > IMG) off = 0;
  }
  DD_WAIT_VM(0);
  __syncthreads();
  const uint4 l = *reinterpret_cast<const uint4*>(smem + tid * 16);
  if ((l.x ^ fold.x ^ fold.y ^ fold.z ^ fold.w) == 0x12345678u) out[0] = 1.f;
}

template <int MODE, int P> static void run(const char* img, float* out, const char* label) {
  const int lds = 78 * 1024, blocks = 512, rounds = 4000;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fill<MODE, P>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((fill<MODE, P>), dim3(blocks), dim3(256), lds, 0, img, out, 200);
  (void)hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL((fill<MODE, P>), dim3(blocks), dim3(256), lds, 0, img, out, rounds);
  (void)hipDeviceSynchronize();
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const double bytes = (double)blocks * rounds * 4.0 * P * 1024.0;
  printf("%-78s P=%d  %7.1f GB/s per CU  %6.2f TB/s chip  %6.3f us per round\n", label, P, bytes / el / 256 * 1e-9, bytes / el * 1e-12, el / rounds * 1e6);
  fflush(stdout);
}

int main() {
  std::vector<unsigned> h(IMG / 4);
  unsigned s = 7u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
  char* img; float* out;
  (void)hipMalloc(&img, IMG); (void)hipMalloc(&out, 4);
  (void)hipMemcpy(img, h.data(), IMG, hipMemcpyHostToDevice);
  printf("LDS fill rate from an L2-resident 295-KB image: 512 four-wave workgroups (2 per CU), every wave fetches P KiB per round, wait, barrier\n");
  run<0, 4>(img, out, "LDS-DMA (global_load_lds_dwordx4), wait vmcnt(0) per round: conv2's stage");
  run<0, 2>(img, out, "LDS-DMA, half stages");
  run<0, 8>(img, out, "LDS-DMA, double stages");
  run<3, 4>(img, out, "LDS-DMA, the previous round's pieces waited for (one round of lookahead)");
  run<1, 4>(img, out, "registers: global_load_dwordx4 -> ds_write_b128");
  run<1, 8>(img, out, "registers, double stages");
  run<2, 4>(img, out, "loads only (no LDS write)");
  run<2, 8>(img, out, "loads only, double stages");
  run<0, 4>(img, out, "LDS-DMA (again)");
  return 0;
}
