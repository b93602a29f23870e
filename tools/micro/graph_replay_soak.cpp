// graph_replay_soak.cpp -- round 6, profiles/r06_experiments.md section 10, OUTSIDE torch: N back-to-back "eval forwards" on the C ABI (include/ddepth.h) --
//     dd_encode (eager)  dd_denoise (two lanes: one hipGraph replay per lane, eager once-per-image kernels around it)  dd_decode (eager)
// -- with no host synchronisation in between; a 4-KB sample of every call's x_0 is parked by an asynchronous device-to-device copy and all of them are compared
// with the first call's at the very end.  Under the Python binding (torch wheel's HIP 7.0 runtime) this pattern goes wrong at turns of the hardware queue unless
// DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; this driver links the HIP runtime it is built against (/opt/rocm), so it answers "is it the runtime version or torch's presence?".
//   g++ -O2 -std=c++17 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/micro/graph_replay_soak.cpp -L diffusiondepth_amd -lddepth_hip -L /opt/rocm/lib -lamdhip64 \
//       -Wl,-rpath,'$ORIGIN/../diffusiondepth_amd' -o build_variants/graph_replay_soak
//   build_variants/graph_replay_soak [iterations 1200] [graph 1|0] [lanes 2] [B 4] [full 0|1 = + dd_add_noise + dd_denoise_once per call]
#include <hip/hip_runtime_api.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ddepth.h"

#define CK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, dd_last_error(h)); return 2; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float urand() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (float)((rng_state >> 40) & 0xFFFFFF) / 16777216.f; }
static float nrand() { float u = urand() + 1e-7f, v = urand(); return std::sqrt(-2.f * std::log(u)) * std::cos(6.2831853f * v); }

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 1200, graph = argc > 2 ? atoi(argv[2]) : 1, lanes = argc > 3 ? atoi(argv[3]) : 2, B = argc > 4 ? atoi(argv[4]) : 4;
  const int H = 352, W = 1216, hh = 176, ww = 608, T = 20, prec = DD_PREC_F16R;
  dd_handle_t h = nullptr;
  if (dd_create(&h, 0, DD_VARIANT_RES) != 0) { fprintf(stderr, "dd_create failed\n"); return 2; }
  struct Wd { const char* name; std::vector<int> shape; int kind; };
  const std::vector<Wd> ws = {
    {"model.noise_embedding.0.weight", {64, 16, 3, 3}, 0}, {"model.noise_embedding.0.bias", {64, 16 * 9}, 1}, {"model.noise_embedding.1.weight", {64}, 2}, {"model.noise_embedding.1.bias", {64}, 3},
    {"model.noise_embedding.3.weight", {256, 64, 3, 3}, 0}, {"model.noise_embedding.3.bias", {256, 64 * 9}, 1}, {"model.noise_embedding.4.weight", {256}, 2}, {"model.noise_embedding.4.bias", {256}, 3},
    {"model.time_embedding.weight", {1280, 256}, 4},
    {"model.pred.0.weight", {64, 256, 3, 3}, 0}, {"model.pred.0.bias", {64, 256 * 9}, 1}, {"model.pred.1.weight", {64}, 2}, {"model.pred.1.bias", {64}, 3},
    {"model.pred.3.weight", {16, 64, 3, 3}, 0}, {"model.pred.3.bias", {16, 64 * 9}, 1}, {"model.pred.4.weight", {16}, 2}, {"model.pred.4.bias", {16}, 3}};
  for (const Wd& w : ws) {
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
  CK(dd_set_option(h, "graph", graph));
  CK(dd_set_option(h, "streams", lanes));
  int64_t gdef = -1; dd_get_counter(h, "graph_default", &gdef);
  const size_t n16 = (size_t)B * 16 * hh * ww, n256 = (size_t)B * 256 * hh * ww;
  float *xT, *cond, *x0, *park;
  HK(hipMalloc(&xT, n16 * 4)); HK(hipMalloc(&x0, n16 * 4)); HK(hipMalloc(&cond, n256 * 4));
  const size_t SAMPLE = 1024;                                // floats parked per lane and call (the first SAMPLE of image 0 and of the last image)
  HK(hipMalloc(&park, (size_t)iters * 2 * SAMPLE * 4));
  { std::vector<float> v(n256); for (size_t i = 0; i < n16; ++i) v[i] = nrand(); HK(hipMemcpy(xT, v.data(), n16 * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < n256; ++i) v[i] = std::fabs(nrand()) * 0.5f; HK(hipMemcpy(cond, v.data(), n256 * 4, hipMemcpyHostToDevice)); }
  HK(hipDeviceSynchronize());
  hipStream_t s = nullptr;
  const size_t img = (size_t)16 * hh * ww;
  // the rest of a head forward's library calls (the ddim_loss part: q_sample + ONE epsilon-network call with per-sample timesteps), argv[5] = 1
  const int full = argc > 5 ? atoi(argv[5]) : 0;
  float *noisy = nullptr, *eps = nullptr; long long* tdev = nullptr;
  if (full) {
    HK(hipMalloc(&noisy, n16 * 4)); HK(hipMalloc(&eps, n16 * 4)); HK(hipMalloc(&tdev, B * 8));
    std::vector<long long> tt(B); for (int i = 0; i < B; ++i) tt[i] = (37 + 211 * i) % 1000;
    HK(hipMemcpy(tdev, tt.data(), B * 8, hipMemcpyHostToDevice));
  }
  for (int it = 0; it < iters; ++it) {
    CK(dd_denoise(h, xT, cond, x0, B, hh, ww, hh, ww, T, prec, s));
    if (full) {
      CK(dd_add_noise(h, x0, xT, reinterpret_cast<const int64_t*>(tdev), noisy, B, 16, hh, ww, s));
      CK(dd_denoise_once(h, noisy, reinterpret_cast<const int64_t*>(tdev), cond, eps, B, hh, ww, hh, ww, prec, s));
    }
    HK(hipMemcpyAsync(park + ((size_t)it * 2) * SAMPLE, x0, SAMPLE * 4, hipMemcpyDeviceToDevice, s));                                   // lane 0's first image
    HK(hipMemcpyAsync(park + ((size_t)it * 2 + 1) * SAMPLE, x0 + (size_t)(B - 1) * img, SAMPLE * 4, hipMemcpyDeviceToDevice, s));      // the last lane's last image
  }
  HK(hipDeviceSynchronize());
  std::vector<float> hp((size_t)iters * 2 * SAMPLE);
  HK(hipMemcpy(hp.data(), park, hp.size() * 4, hipMemcpyDeviceToHost));
  int bad[2] = {0, 0}, first_bad[2] = {-1, -1}, last_bad[2] = {-1, -1};
  for (int it = 1; it < iters; ++it)
    for (int l = 0; l < 2; ++l)
      if (memcmp(&hp[((size_t)it * 2 + l) * SAMPLE], &hp[(size_t)l * SAMPLE], SAMPLE * 4) != 0) { ++bad[l]; if (first_bad[l] < 0) first_bad[l] = it; last_bad[l] = it; }
  int64_t graphs = 0, eager = 0, ov = -2, rt = -1;
  dd_get_counter(h, "graph_launches", &graphs); dd_get_counter(h, "eager_loops", &eager); dd_get_counter(h, "lane_overlap", &ov); dd_get_counter(h, "lane_probe_retries", &rt);
  int rtv = 0; (void)hipRuntimeGetVersion(&rtv);
  const char* pc = getenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE");
  printf("HIP runtime %d, DEBUG_CLR_GRAPH_PACKET_CAPTURE=%s (graph_default %lld), graph=%d lanes=%d B=%d: %d calls; first image differs from call 0 in %d calls (%d..%d), last image in %d (%d..%d); "
         "graph launches %lld, eager loops %lld, lane_overlap %lld retries %lld -> %s\n", rtv, pc ? pc : "(unset)", (long long)gdef, graph, lanes, B, iters, bad[0], first_bad[0], last_bad[0], bad[1], first_bad[1], last_bad[1],
         (long long)graphs, (long long)eager, (long long)ov, (long long)rt, (bad[0] || bad[1]) ? "WRONG" : "clean");
  dd_destroy(h);
  return (bad[0] || bad[1]) ? 1 : 0;
}
