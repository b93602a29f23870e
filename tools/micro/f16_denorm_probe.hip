// f16_denorm_probe.hip -- does v_mfma_f32_32x32x16_f16 honour SUBNORMAL f16 inputs on gfx950, and does v_cvt_f16_f32 produce them?
// (the split-f16 mode of dd_igemm2.hip carries operands as hi + lo f16 pairs; lo halves of small values are subnormal.  The kernels
// pre-scale operands so that this does not matter for ordinary values; this probe records what the hardware does.)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/f16_denorm_probe.hip -o /tmp/f16_denorm_probe && /tmp/f16_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__global__ void probe(float a_val, float b_val, float* out, uint16_t* cvt) {
  // A[i][k] = a_val for k == 0 (lanes 0..31, element 0), B[k][j] = b_val likewise: D[i][j] = a_val * b_val
  const int lane = threadIdx.x;
  f16x8_t a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  const _Float16 ah = (_Float16)a_val, bh = (_Float16)b_val;      // v_cvt_f16_f32
  if (lane < 32) { a[0] = ah; b[0] = bh; }
  f32x16_t acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (lane == 0) { out[0] = acc[0]; uint16_t u; __builtin_memcpy(&u, &ah, 2); cvt[0] = u; }
}

int main() {
  float* d; uint16_t* c;
  hipMalloc(&d, 4); hipMalloc(&c, 2);
  const float cases[][2] = {{1.0f, 1.0f}, {3.0e-6f, 1024.0f}, {1024.0f, 3.0e-6f}, {5.9604645e-8f, 65504.0f}, {6.0e-5f, 6.0e-5f}};
  for (auto& cs : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, cs[0], cs[1], d, c);
    float r; uint16_t u;
    hipMemcpy(&r, d, 4, hipMemcpyDeviceToHost); hipMemcpy(&u, c, 2, hipMemcpyDeviceToHost);
    const _Float16 ah = (_Float16)cs[0], bh = (_Float16)cs[1];
    printf("a=%.6g (f16 bits 0x%04x, host cvt %.6g) b=%.6g : mfma %.9g   expected %.9g   %s\n", cs[0], u, (float)ah, cs[1], r, (float)ah * (float)bh,
           (r == (float)ah * (float)bh) ? "subnormals honoured / exact" : "MISMATCH (flushed?)");
  }
  return 0;
}
