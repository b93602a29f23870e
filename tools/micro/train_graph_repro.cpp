// train_graph_repro.cpp -- the 16-bit training step's intermittent non-finite gradients (profiles/r05_experiments.md section 4), OUTSIDE torch:
// a plain C++ driver on the C ABI (include/ddepth.h) that runs, `iters` times,
//     dd_denoise(keep_trajectory)   -- the trajectory-keeping forward of a training step: a replayed hipGraph with option train_graphs = 1, eager with 0
//     dd_denoise_backward(ticket)   -- the ~600-launch eager backward burst that reads the per-step slots the forward wrote
// WITHOUT clearing the parameter gradients in between and WITHOUT looking at anything until the end (every observation made the fault go away in round
// 5): a NaN / Inf that any iteration produces stays in the accumulated gradients; one count at the very end is the verdict.  Memory is hipMalloc'd
// here, the stream is the legacy NULL stream (mode 0), a created blocking stream (1) or a created non-blocking stream (2).
//   g++ -O2 -std=c++17 -pthread -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/micro/train_graph_repro.cpp -L diffusiondepth_amd -lddepth_hip -L /opt/rocm/lib -lamdhip64 \
//       -Wl,-rpath,'$ORIGIN/../diffusiondepth_amd' -o build_variants/train_graph_repro
//   build_variants/train_graph_repro <train_graphs 0|1> <stream mode 0|1|2> [iters 40] [precision 2=bf16|3=f16] [B 4] [h 176] [w 608] [T 20] [streams 1] [torchlike 0|1]
// torchlike = 1: what a head's training step adds around the two calls -- ONE epsilon-network call with its own kept activations between the loop forward and the
// backward (ddim_loss: dd_denoise_once + dd_denoise_once_backward), and both backward calls issued from a SECOND host thread (torch's autograd engine runs
// backward nodes on its own worker thread), with a host synchronisation after the forward and after the backward (the loss / gradient checks of the round-5 harness)
#include <hip/hip_runtime_api.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "ddepth.h"

#define CK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, dd_last_error(h)); return 2; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float urand() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (float)((rng_state >> 40) & 0xFFFFFF) / 16777216.f; }
static float nrand() { float u = urand() + 1e-7f, v = urand(); return std::sqrt(-2.f * std::log(u)) * std::cos(6.2831853f * v); }

int main(int argc, char** argv) {
  const int train_graphs = argc > 1 ? atoi(argv[1]) : 1, smode = argc > 2 ? atoi(argv[2]) : 0, iters = argc > 3 ? atoi(argv[3]) : 40;
  const int prec = argc > 4 ? atoi(argv[4]) : DD_PREC_BF16, B = argc > 5 ? atoi(argv[5]) : 4, hh = argc > 6 ? atoi(argv[6]) : 176, ww = argc > 7 ? atoi(argv[7]) : 608;
  const int T = argc > 8 ? atoi(argv[8]) : 20, lanes = argc > 9 ? atoi(argv[9]) : 1, torchlike = argc > 10 ? atoi(argv[10]) : 0;
  dd_handle_t h = nullptr;
  if (dd_create(&h, 0, DD_VARIANT_RES) != 0) { fprintf(stderr, "dd_create failed\n"); return 2; }
  // the denoiser's parameters in the reference's shapes (torch's default initialisation scales, a non-trivial GroupNorm affine)
  struct W { const char* name; std::vector<int> shape; int kind; };      // kind 0 conv weight (fan_in = in * k * k), 1 bias of that conv, 2 gamma, 3 beta, 4 embedding
  const std::vector<W> ws = {
    {"model.noise_embedding.0.weight", {64, 16, 3, 3}, 0}, {"model.noise_embedding.0.bias", {64, 16 * 9}, 1}, {"model.noise_embedding.1.weight", {64}, 2}, {"model.noise_embedding.1.bias", {64}, 3},
    {"model.noise_embedding.3.weight", {256, 64, 3, 3}, 0}, {"model.noise_embedding.3.bias", {256, 64 * 9}, 1}, {"model.noise_embedding.4.weight", {256}, 2}, {"model.noise_embedding.4.bias", {256}, 3},
    {"model.time_embedding.weight", {1280, 256}, 4},
    {"model.pred.0.weight", {64, 256, 3, 3}, 0}, {"model.pred.0.bias", {64, 256 * 9}, 1}, {"model.pred.1.weight", {64}, 2}, {"model.pred.1.bias", {64}, 3},
    {"model.pred.3.weight", {16, 64, 3, 3}, 0}, {"model.pred.3.bias", {16, 64 * 9}, 1}, {"model.pred.4.weight", {16}, 2}, {"model.pred.4.bias", {16}, 3}};
  for (const W& w : ws) {
    long long n = 1; float bound = 1.f;
    if (w.kind == 0) { for (int d : w.shape) n *= d; bound = 1.f / std::sqrt((float)(w.shape[1] * 9)); }
    else if (w.kind == 1) { n = w.shape[0]; bound = 1.f / std::sqrt((float)w.shape[1]); }
    else for (int d : w.shape) n *= d;
    std::vector<float> v((size_t)n);
    for (auto& x : v) x = w.kind <= 1 ? (2.f * urand() - 1.f) * bound : w.kind == 2 ? 0.6f + 0.8f * urand() : w.kind == 3 ? -0.3f + 0.6f * urand() : nrand();
    CK(dd_set_weight(h, w.name, v.data(), n));
  }
  CK(dd_commit_weights(h, nullptr));
  std::vector<float> acp(1000);
  { double a = 1.0; for (int i = 0; i < 1000; ++i) { a *= 1.0 - (1e-4 + (0.02 - 1e-4) * i / 999.0); acp[i] = (float)a; } }
  CK(dd_set_schedule(h, acp.data(), 1000));
  CK(dd_set_option(h, "train_graphs", train_graphs));
  CK(dd_set_option(h, "streams", lanes));
  hipStream_t s = nullptr;
  if (smode == 1) HK(hipStreamCreate(&s));
  if (smode == 2) HK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t n16 = (size_t)B * 16 * hh * ww, n256 = (size_t)B * 256 * hh * ww;
  float *xT, *cond, *x0, *g0, *gc, *gw;
  HK(hipMalloc(&xT, n16 * 4)); HK(hipMalloc(&x0, n16 * 4)); HK(hipMalloc(&g0, n16 * 4)); HK(hipMalloc(&cond, n256 * 4)); HK(hipMalloc(&gc, n256 * 4));
  const long long nw = 64LL * 256 * 9;
  HK(hipMalloc(&gw, nw * 4));
  { std::vector<float> v(n256); for (size_t i = 0; i < n16; ++i) v[i] = nrand(); HK(hipMemcpy(xT, v.data(), n16 * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < n16; ++i) v[i] = nrand() * 1e-3f; HK(hipMemcpy(g0, v.data(), n16 * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < n256; ++i) v[i] = std::fabs(nrand()) * 0.5f; HK(hipMemcpy(cond, v.data(), n256 * 4, hipMemcpyHostToDevice)); }
  HK(hipDeviceSynchronize());
  CK(dd_zero_grad(h, s));
  float *eps = nullptr, *geps = nullptr; long long* tdev = nullptr;
  if (torchlike) {
    HK(hipMalloc(&eps, n16 * 4)); HK(hipMalloc(&geps, n16 * 4)); HK(hipMalloc(&tdev, B * 8));
    std::vector<long long> tt(B); for (int i = 0; i < B; ++i) tt[i] = (37 + 211 * i) % 1000;
    HK(hipMemcpy(tdev, tt.data(), B * 8, hipMemcpyHostToDevice)); HK(hipMemcpy(geps, g0, n16 * 4, hipMemcpyDeviceToDevice));
  }
  int thread_rc = 0;
  for (int it = 0; it < iters; ++it) {
    CK(dd_set_option(h, "keep_trajectory", 1));
    CK(dd_denoise(h, xT, cond, x0, B, hh, ww, hh, ww, T, prec, s));
    int64_t ticket = 0, ticket1 = 0;
    CK(dd_get_counter(h, "trajectory_ticket", &ticket));
    if (torchlike) {
      CK(dd_denoise_once(h, x0, reinterpret_cast<const int64_t*>(tdev), cond, eps, B, hh, ww, hh, ww, prec, s));
      CK(dd_get_counter(h, "trajectory_ticket", &ticket1));
    }
    CK(dd_set_option(h, "keep_trajectory", 0));
    if (!torchlike) {
      CK(dd_set_option(h, "use_trajectory", ticket));
      CK(dd_denoise_backward(h, xT, cond, g0, nullptr, gc, B, hh, ww, hh, ww, T, prec, s));
      continue;
    }
    HK(hipStreamSynchronize(s));                    // (the harness looked at the loss here)
    std::thread bw([&]() {
      (void)hipSetDevice(0);
      if (dd_set_option(h, "use_trajectory", ticket1) || dd_denoise_once_backward(h, x0, reinterpret_cast<const int64_t*>(tdev), cond, geps, nullptr, gc, B, hh, ww, hh, ww, prec, s) ||
          dd_set_option(h, "use_trajectory", ticket) || dd_denoise_backward(h, xT, cond, g0, nullptr, gc, B, hh, ww, hh, ww, T, prec, s)) thread_rc = 1;
    });
    bw.join();
    if (thread_rc) { fprintf(stderr, "backward thread failed: %s\n", dd_last_error(h)); return 2; }
    HK(hipStreamSynchronize(s));                    // (... and at the gradients here)
  }
  // the verdict: one look at the end
  CK(dd_get_grad(h, "model.pred.0.weight", gw, nw, s));
  HK(hipStreamSynchronize(s));
  HK(hipDeviceSynchronize());
  std::vector<float> hw((size_t)nw), hx(n16), hc(1 << 20);
  HK(hipMemcpy(hw.data(), gw, nw * 4, hipMemcpyDeviceToHost)); HK(hipMemcpy(hx.data(), x0, n16 * 4, hipMemcpyDeviceToHost)); HK(hipMemcpy(hc.data(), gc, hc.size() * 4, hipMemcpyDeviceToHost));
  long long bad_w = 0, bad_x = 0, bad_c = 0; double amax = 0;
  for (float v : hw) { if (!std::isfinite(v)) ++bad_w; else amax = std::fmax(amax, std::fabs(v)); }
  for (float v : hx) if (!std::isfinite(v)) ++bad_x;
  for (float v : hc) if (!std::isfinite(v)) ++bad_c;
  int64_t graphs = 0, eager = 0, reuse = 0;
  dd_get_counter(h, "graph_launches", &graphs); dd_get_counter(h, "eager_loops", &eager); dd_get_counter(h, "trajectory_reuses", &reuse);
  printf("train_graphs=%d stream_mode=%d iters=%d prec=%d B=%d %dx%d T=%d lanes=%d torchlike=%d: accumulated grad(pred.0.weight) non-finite %lld of %lld (max |g| %.3e), last x_0 non-finite %lld, last grad_cond non-finite %lld; "
         "graph launches %lld, eager loops %lld, trajectory reuses %lld -> %s\n", train_graphs, smode, iters, prec, B, hh, ww, T, lanes, torchlike, bad_w, nw, amax, bad_x, bad_c,
         (long long)graphs, (long long)eager, (long long)reuse, (bad_w || bad_x || bad_c) ? "FAILED" : "clean");
  dd_destroy(h);
  return (bad_w || bad_x || bad_c) ? 1 : 0;
}
