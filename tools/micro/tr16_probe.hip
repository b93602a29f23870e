// What does ds_read_b64_tr_b16 return?  LDS holds u16 element i = i; lane l supplies byte address base + l * stride; prints every lane's 4 values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(unsigned short* out, int stride_bytes) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + threadIdx.x * stride_bytes));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main(int argc, char** argv) {
  int stride = argc > 1 ? atoi(argv[1]) : 8;
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
  unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("stride %d bytes: lane -> 4 element indices (lane l points at elements %d*l ..)\n", stride, stride / 2);
  for (int l = 0; l < 64; ++l) printf("l%02d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l % 4 == 3) ? "\n" : "   ");
  return 0;
}
