// tests/host_emul/igemm2_host.cpp -- TEST INFRASTRUCTURE: a small C interface (ctypes, tests/test_igemm2_host_emulation.py) to the launchers of the
// production convolution kernels (diffusiondepth_amd/csrc/dd_igemm2.hip) inside the host-emulated library (ddepth_host.cpp).  The
// kernels have hundreds of green GPU parity tests; what the GPU cannot show is that they are free of RACES that the hardware's timing happens
// not to trigger: here a wave runs ahead as far as the workgroup barriers allow, and an LDS-DMA lands either at issue or as late as the
// s_waitcnt arithmetic permits (dd_gcn.h) -- the results must not depend on any of it.
#include "dd_kernels.h"

extern "C" {

// cin, cout, cout_pad, ck, tg, nt, th, ks of (layer, element kind): the packed-weight geometry the host side packs with
void emu_geom2(int layer, int ek, int* out8) {
  const dd::PackGeom g = dd::conv_pack_geom2(layer, ek);
  out8[0] = g.cin; out8[1] = g.cout; out8[2] = g.cout_pad; out8[3] = g.ck; out8[4] = g.tg; out8[5] = g.nt; out8[6] = g.th; out8[7] = g.ks;
}

int emu_conv2(int layer, int ek, const void* in, const void* wpack, const float* bias, void* out, double* stats_out, const double* stats_in,
              const float* gn_gamma, const float* gn_beta, const void* cond, const float* emb, const long long* tvec, int t_base, int t_bstride,
              const float* y4, float* xout, const float* c1c2, int step, int B, int h, int w, const float* cadd, const float* etab,
              const void* addend) {
  dd::ConvParams p{};
  p.in = in; p.wpack = wpack; p.bias = bias; p.out = out; p.stats_out = stats_out; p.stats_in = stats_in; p.gn_gamma = gn_gamma;
  p.gn_beta = gn_beta; p.cond = cond; p.emb = emb; p.tvec = tvec; p.t_base = t_base; p.t_bstride = t_bstride; p.y4 = y4; p.xout = xout;
  p.c1c2 = c1c2; p.step = step; p.B = B; p.h = h; p.w = w; p.cadd = cadd; p.etab = etab; p.addend = addend;
  const dd::PackGeom g = dd::conv_pack_geom2(layer, ek);
  p.tiles_x = (w + 31) / 32;
  p.tiles_y = (h + g.th - 1) / g.th;
  return dd::launch_conv_igemm2(layer, ek, p, nullptr);
}

}  // extern "C"
