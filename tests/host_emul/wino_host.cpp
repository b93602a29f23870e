// tests/host_emul/wino_host.cpp -- TEST INFRASTRUCTURE: a small C interface (ctypes, tests/test_wino_host_emulation.py) to the Winograd kernels'
// launchers (launch_conv_wino_layer, launch_conv_wino_raw, launch_wino_gn_table) and weight packer (wino_pack_u) of
// diffusiondepth_amd/csrc/dd_wino.hip inside the host-emulated library (ddepth_host.cpp).  Nothing here restates the product's code.
#include "dd_elem.h"

namespace {
uint16_t cvt_f16(float f) { return (uint16_t)dd::f32_to_f16(f); }
uint16_t cvt_bf16(float f) { return (uint16_t)dd::f32_to_bf16(f); }
}  // namespace

extern "C" {

long long emu_wino_pack_bytes(int cout, int cin) { return (long long)dd::wino_pack_bytes(cout, cin); }
void emu_wino_pack(const float* w_oihw, int cout, int cin, int ek, uint16_t* out) {
  dd::wino_pack_u(w_oihw, cout, cin, ek == dd::EK_BF16 ? cvt_bf16 : cvt_f16, out);
}

// (a, b, e) table of the GroupNorm (+ time embedding) prologue: tab [B][C][4] floats
int emu_wino_table(const double* stats, const float* gamma, const float* beta, const float* emb, const long long* tvec, int t_base,
                   int t_bstride, int B, int h, int w, int C, float* tab) {
  dd::ConvParams p{};
  p.stats_in = stats; p.gn_gamma = gamma; p.gn_beta = beta; p.emb = emb; p.tvec = tvec; p.t_base = t_base; p.t_bstride = t_bstride;
  p.B = B; p.h = h; p.w = w;
  return dd::launch_wino_gn_table(p, C, tab, emb != nullptr, nullptr);
}

// version 1: the single-buffered convB kernel (layer must be 6); version 2: the generalised double-buffered kernel
int emu_wino_layer(int version, int layer, int ek, int packed_f16, int dma, const void* in, const void* cond, const void* wpack,
                   const float* bias, const float* tab, void* out, double* stats_out, int B, int h, int w) {
  dd::ConvParams p{};
  p.in = in; p.cond = cond; p.wpack = wpack; p.bias = bias; p.wino_tab = tab; p.out = out; p.stats_out = stats_out;
  p.B = B; p.h = h; p.w = w;
  p.wino_flags = dma ? 1 : 0;
  if (version == 1) return layer == 6 ? dd::launch_conv_wino_raw(ek, p, nullptr, 1) : hipErrorInvalidValue;
  return dd::launch_conv_wino_layer(layer, ek, p, nullptr, packed_f16 != 0);
}

}  // extern "C"
