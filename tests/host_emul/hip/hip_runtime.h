// tests/host_emul/hip/hip_runtime.h -- TEST INFRASTRUCTURE, never part of the product.
//
// A stand-in for <hip/hip_runtime.h> that lets a kernel translation unit of diffusiondepth_amd/csrc be compiled FOR THE HOST (clang++, x86) and
// executed workgroup by workgroup, so that the index arithmetic, LDS images, barrier placement and buffer-reuse hazards of a kernel that has no
// GPU time yet can be checked against a NumPy / torch reference on the CPU (tests/test_wino_host_emulation.py).
//
// Execution model: one workgroup at a time; every work-item is a fiber (ucontext) on ONE OS thread.  A fiber runs until it blocks (__syncthreads
// or a wave-wide operation) or returns; the fibers of ONE WAVE are resumed round-robin until the whole wave waits at a workgroup barrier, then
// the next wave runs.  So wave 0 always gets as far ahead of the others as the workgroup barriers permit (order 0; order 1 = the last wave
// does): a missing barrier shows up as a wrong result deterministically, not as a race that may or may not fire.  LDS is one global array shared by the fibers of the workgroup (workgroups run one after the other).
// Wave-wide operations (64 consecutive work-items): __shfl_xor and the 32x32 MFMAs exchange their operands through a per-wave scratch with two
// wave-level rendezvous; the MFMA register layout is the one dd_elem.h / the CDNA4 ISA documents (A: row = lane % 32, k = 8 * (lane / 32) + e;
// B: column = lane % 32, same k; D register r: row 8 * (r / 4) + 4 * (lane / 32) + r % 4, column lane % 32) -- the layout itself was confirmed
// on the GPU by the kernels that already run there; the emulation only has to be consistent with it.
// Not modelled: timing, bank conflicts, the asynchrony of global loads / LDS-DMA (copies complete at issue), occupancy.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <deque>
#include <functional>
#include <vector>
#include <chrono>

#define __host__
#define __device__
#define __global__
#define __shared__ static          /* statically sized LDS arrays: one workgroup runs at a time; dynamic LDS is DD_DYN_SMEM (dd_gcn.h) */
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

using std::max;
using std::min;

struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(8) uint2 { uint32_t x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorLaunchFailure = 719;
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorLaunchFailure ? "host emulation: deadlock" : "host emulation: error"; }
typedef void* hipStream_t;
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d) { p->multiProcessorCount = 256; return d == 0 ? hipSuccess : hipErrorInvalidValue; }      // the MI355X's CU count
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }   // a 4-CU "device": persistent grids stay small
// calls after which the HOST would wait for the device (synchronise, blocking copy, allocation): counted so that tests can assert that a
// steady-state entry point never blocks the host (the host must stay ahead of the GPU)
namespace hostemu { inline unsigned long& blocking_calls() { static unsigned long n = 0; return n; } }
static inline hipError_t hipDeviceSynchronize() { ++hostemu::blocking_calls(); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { ++hostemu::blocking_calls(); return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = malloc(8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
// "device" memory is host memory, handed out filled with 0xFF (NaN as float / double): a kernel that reads what nobody wrote shows up
static inline hipError_t hipMalloc(void** p, size_t n) {
  ++hostemu::blocking_calls();
  *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256);
  if (!*p) return hipErrorOutOfMemory;
  memset(*p, 0xFF, n);
  return hipSuccess;
}
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
// free / total "device" memory: HOSTEMU_FREE_BYTES in the environment (read at every call: a test can shrink the device), default 1 TiB
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
  const char* e = getenv("HOSTEMU_FREE_BYTES");
  *free_b = e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)1 << 40);
  *total_b = (size_t)1 << 40;
  return hipSuccess;
}
static inline hipError_t hipFree(void* p) { ++hostemu::blocking_calls(); free(p); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
struct hostemu_event { double t; };
typedef hostemu_event* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hostemu_event{0.0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
// every emulated stream executes at the call (or inside a captured graph, in order): an event is always complete when another stream waits for it
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { ++hostemu::blocking_calls(); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)((b->t - a->t) * 1e3); return hipSuccess; }
struct hostemu_graph { std::vector<std::function<void()>> nodes; };
typedef hostemu_graph* hipGraph_t;
typedef hostemu_graph* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };

// Context switch between fibers: callee-saved registers + stack pointer, no signal-mask system call (glibc's swapcontext makes one per
// switch; a kernel with MFMAs switches millions of times).  x86-64 SysV only; defined once, in tests/host_emul/ddepth_host.cpp.
extern "C" void hostemu_switch(void** save_sp, void* load_sp);

namespace hostemu {

constexpr int WAVE = 64;
constexpr size_t FIBER_STACK = 512 * 1024;

struct Idx3 { unsigned x, y, z; };
struct Rendezvous { int count = 0; unsigned gen = 0; int live = 0; };     // live: members that have not returned from the kernel yet
struct Fiber { void* sp; char* stack; bool done; Rendezvous* wait; unsigned wait_gen; unsigned wop; };   // wop: wave-wide operations executed
// one entry of a work-item's VMEM queue (s_waitcnt vmcnt counts them in issue order): an ordinary global load (dst == nullptr: the value was
// taken at issue, only its slot in the queue matters) or a 16-byte LDS-DMA piece that lands when it is retired (dma_late) or landed at issue
struct VmEntry { char* dst; const char* src; };

struct State {
  std::vector<Fiber> fibers;
  void* sched_sp = nullptr;
  int cur = -1, nthreads = 0;
  const std::function<void()>* body = nullptr;
  Rendezvous block;
  std::vector<Rendezvous> wave;
  std::vector<float> wave_a, wave_b;              // per wave: MFMA operands [64][8] x 2
  std::vector<uint64_t> wave_x;                   // per wave: shuffle / permlane values [64][2]
  std::vector<std::deque<VmEntry>> vmq;           // per work-item
  int dma_late = 0;                               // 0: an LDS-DMA lands at issue; 1: only when an s_waitcnt retires it
  hostemu_graph* capturing = nullptr;             // stream capture: launches / copies are recorded instead of executed
  unsigned long n_launches = 0;
  unsigned long progress = 0;
  int order = 0;                                  // 0: ascending work-item order, 1: descending
  hipError_t last_error = hipSuccess;
  unsigned long n_block_barriers = 0, n_wave_ops = 0;
};
inline State& st() { static State s; return s; }

inline void yield() { State& s = st(); hostemu_switch(&s.fibers[s.cur].sp, s.sched_sp); }
// A barrier waits for the members that are still alive: a wave that has ended is not waited for (s_barrier semantics; the wave-specialised
// kernels rely on it), and lanes that have returned do not take part in wave-wide operations.
inline void rendezvous(Rendezvous& r) {
  State& s = st();
  const unsigned gen = r.gen;
  if (++r.count >= r.live) { r.count = 0; ++r.gen; ++s.progress; return; }
  Fiber& f = s.fibers[s.cur];
  f.wait = &r;                                    // the scheduler does not resume this fiber before the generation moves on
  f.wait_gen = gen;
  while (r.gen == gen) yield();
  f.wait = nullptr;
}
inline void member_left(Rendezvous& r) {          // a work-item returned: the others must not wait for it
  State& s = st();
  --r.live;
  if (r.live > 0 && r.count >= r.live) { r.count = 0; ++r.gen; ++s.progress; }
}

}  // namespace hostemu

// work-item / workgroup coordinates: plain globals, rewritten by the scheduler every time a fiber is resumed
extern hostemu::Idx3 threadIdx, blockIdx, blockDim, gridDim;

namespace hostemu {

inline int lane_id() { return st().cur % WAVE; }        // waves are formed from the linearised work-item id (x fastest)
inline int wave_id() { return st().cur / WAVE; }
inline int wave_width() {                       // the last wave of a workgroup may be partial
  State& s = st();
  return std::min(WAVE, s.nthreads - wave_id() * WAVE);
}
inline void wave_sync() { State& s = st(); ++s.n_wave_ops; rendezvous(s.wave[wave_id()]); }
// Exchange buffer of this lane's next wave-wide operation.  Every such operation is: write the own slot, ONE rendezvous, read the others' slots.
// Two buffers used alternately make a second rendezvous unnecessary: a lane can run at most one operation ahead of the slowest lane of its
// wave (the next rendezvous holds it), so when it writes buffer p again every lane has finished reading what buffer p held two operations ago.
inline size_t wave_buf() { State& s = st(); return (size_t)wave_id() * 2 + (s.fibers[s.cur].wop++ & 1u); }
inline void fiber_entry() {
  State& s = st();
  (*s.body)();
  s.fibers[s.cur].done = true;
  ++s.progress;
  member_left(s.wave[wave_id()]);
  member_left(s.block);
  hostemu_switch(&s.fibers[s.cur].sp, s.sched_sp);      // never resumed
  abort();
}

inline hipError_t run_grid(dim3 grid, dim3 block, const std::function<void()>& body) {
  State& s = st();
  ++s.n_launches;
  const int n = (int)(block.x * block.y * block.z);
  s.nthreads = n;
  s.body = &body;
  blockDim = Idx3{block.x, block.y, block.z};
  gridDim = Idx3{grid.x, grid.y, grid.z};
  const int nw = (n + WAVE - 1) / WAVE;
  s.wave.assign(nw, Rendezvous());
  s.wave_a.assign((size_t)nw * 2 * WAVE * 8, 0.f);      // x 2: exchange buffers alternate between consecutive wave-wide operations
  s.wave_b.assign((size_t)nw * 2 * WAVE * 8, 0.f);
  s.wave_x.assign((size_t)nw * 2 * WAVE * 2, 0);
  s.vmq.assign(n, std::deque<VmEntry>());
  if ((int)s.fibers.size() < n) {
    const size_t old = s.fibers.size();
    s.fibers.resize(n);
    for (size_t i = old; i < (size_t)n; ++i) s.fibers[i].stack = (char*)malloc(FIBER_STACK);
  }
  auto set_thread = [&](int i) {
    s.cur = i;
    threadIdx = Idx3{(unsigned)i % block.x, ((unsigned)i / block.x) % block.y, (unsigned)i / (block.x * block.y)};
  };
  for (unsigned bz = 0; bz < grid.z; ++bz)
  for (unsigned by = 0; by < grid.y; ++by)
  for (unsigned bx = 0; bx < grid.x; ++bx) {
    blockIdx = Idx3{bx, by, bz};
    s.block = Rendezvous();
    s.block.live = n;
    for (int w = 0; w < nw; ++w) { s.wave[w] = Rendezvous(); s.wave[w].live = std::min(WAVE, n - w * WAVE); }
    for (int i = 0; i < n; ++i) {
      Fiber& f = s.fibers[i];
      f.done = false;
      f.wait = nullptr;
      f.wop = 0;
      s.vmq[i].clear();
      // a fresh stack that hostemu_switch can "return" into: six callee-saved registers, then the entry point, then a null return address
      void** top = reinterpret_cast<void**>(reinterpret_cast<uintptr_t>(f.stack + FIBER_STACK) & ~uintptr_t(15));
      top[-1] = nullptr;
      top[-2] = reinterpret_cast<void*>(&fiber_entry);
      for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
      f.sp = top - 8;
    }
    // A WAVE is the unit that runs ahead: its work-items are resumed round-robin until all of them wait at a workgroup barrier (or are done);
    // only then does the next wave get the processor.  (Wave-wide operations make the lanes of one wave advance together anyway; scheduling
    // all waves round-robin would also keep the WAVES in step from MFMA to MFMA and hide every missing workgroup barrier.)
    int remaining = n;
    while (remaining > 0) {
      const unsigned long before = s.progress;
      for (int wk = 0; wk < nw; ++wk) {
        const int wv = s.order == 0 ? wk : nw - 1 - wk;
        const int lo = wv * WAVE, hi = std::min(n, lo + WAVE);
        unsigned long p0;
        do {
          p0 = s.progress;
          for (int k = lo; k < hi; ++k) {
            const int i = s.order == 0 ? k : hi - 1 - (k - lo);
            Fiber& f = s.fibers[i];
            if (f.done || (f.wait && f.wait->gen == f.wait_gen)) continue;      // finished, or still blocked: nothing to run
            set_thread(i);
            hostemu_switch(&s.sched_sp, f.sp);
          }
        } while (s.progress != p0);
      }
      remaining = 0;
      for (int i = 0; i < n; ++i) remaining += s.fibers[i].done ? 0 : 1;
      if (remaining > 0 && s.progress == before) {     // every live work-item waits and nothing was released: divergent barrier
        fprintf(stderr, "hostemu: deadlock in workgroup (%u,%u,%u) (%d work-items blocked)\n", bx, by, bz, remaining);
        s.last_error = hipErrorLaunchFailure;
        return s.last_error;
      }
    }
  }
  return hipSuccess;
}
template <class Body>
inline hipError_t launch(dim3 grid, dim3 block, Body&& body_) {
  State& s = st();
  const std::function<void()> body = body_;
  if (s.capturing) {                               // stream capture: the launch becomes a graph node (arguments bound by value, as in HIP)
    s.capturing->nodes.push_back([grid, block, body]() { (void)run_grid(grid, block, body); });
    return hipSuccess;
  }
  return run_grid(grid, block, body);
}

// D = A (32 x K) . B (K x 32) + C with K = 16 (2-byte operands, 8 per lane) or K = 2 (fp32, 1 per lane)
typedef __attribute__((ext_vector_type(16))) float v16f;
inline v16f mfma_32x32(const float* a, const float* b, int per_lane, v16f c) {
  State& s = st();
  const int l = lane_id();
  const size_t buf = wave_buf();
  float* wa = &s.wave_a[buf * WAVE * 8];
  float* wb = &s.wave_b[buf * WAVE * 8];
  for (int e = 0; e < per_lane; ++e) { wa[l * 8 + e] = a[e]; wb[l * 8 + e] = b[e]; }
  wave_sync();
  const int col = l % 32, half = l / 32;
  for (int r = 0; r < 16; ++r) {
    const int row = 8 * (r / 4) + 4 * half + r % 4;
    float acc = 0.f;
    for (int kh = 0; kh < 2; ++kh)
      for (int e = 0; e < per_lane; ++e) acc += wa[(row + 32 * kh) * 8 + e] * wb[(col + 32 * kh) * 8 + e];
    c[r] += acc;
  }
  return c;
}
typedef __attribute__((ext_vector_type(8))) _Float16 v8h;
typedef __attribute__((ext_vector_type(8))) __bf16 v8b;
inline v16f mfma_f16(v8h a, v8h b, v16f c) {
  float fa[8], fb[8];
  for (int e = 0; e < 8; ++e) { fa[e] = (float)a[e]; fb[e] = (float)b[e]; }
  return mfma_32x32(fa, fb, 8, c);
}
inline v16f mfma_bf16(v8b a, v8b b, v16f c) {
  float fa[8], fb[8];
  const uint4 ua = __builtin_bit_cast(uint4, a), ub = __builtin_bit_cast(uint4, b);
  const uint32_t wa[4] = {ua.x, ua.y, ua.z, ua.w}, wb[4] = {ub.x, ub.y, ub.z, ub.w};
  for (int e = 0; e < 4; ++e) {
    fa[2 * e] = __builtin_bit_cast(float, wa[e] << 16); fa[2 * e + 1] = __builtin_bit_cast(float, wa[e] & 0xFFFF0000u);
    fb[2 * e] = __builtin_bit_cast(float, wb[e] << 16); fb[2 * e + 1] = __builtin_bit_cast(float, wb[e] & 0xFFFF0000u);
  }
  return mfma_32x32(fa, fb, 8, c);
}
inline v16f mfma_f32(float a, float b, v16f c) { return mfma_32x32(&a, &b, 1, c); }

template <class T> inline T shfl_from(T v, int src_lane) {         // value of lane src_lane (own value when out of range)
  static_assert(sizeof(T) <= 8, "one 64-bit slot per lane");
  State& s = st();
  uint64_t* wx = &s.wave_x[wave_buf() * WAVE * 2];
  uint64_t bits = 0;
  memcpy(&bits, &v, sizeof(T));
  wx[lane_id() * 2] = bits;
  wave_sync();
  const uint64_t rb = (src_lane >= 0 && src_lane < wave_width()) ? wx[src_lane * 2] : bits;
  T r;
  memcpy(&r, &rb, sizeof(T));
  return r;
}
template <class T> inline T shfl_xor(T v, int mask) { return shfl_from(v, lane_id() ^ mask); }
// v_permlane32_swap_b32 vdst, src0: lanes 32..63 of vdst <-> lanes 0..31 of src0; returns {vdst', src0'}
typedef __attribute__((ext_vector_type(2))) unsigned v2u;
inline v2u permlane32_swap(unsigned vdst, unsigned src0) {
  State& s = st();
  uint64_t* wx = &s.wave_x[wave_buf() * WAVE * 2];
  const int l = lane_id();
  wx[l * 2] = vdst;
  wx[l * 2 + 1] = src0;
  wave_sync();
  v2u r;
  if (l < 32) { r[0] = vdst; r[1] = (unsigned)wx[(l + 32) * 2]; }
  else { r[0] = (unsigned)wx[(l - 32) * 2 + 1]; r[1] = src0; }
  return r;
}

// ds_read_b64_tr_b16 (gfx950): every lane supplies the LDS address of 4 contiguous 16-bit elements; inside each group of 16 lanes the
// 16 x 4 matrix M[lane r][element c] comes back transposed in 4-row blocks: lane l = 4 r0 + i receives (M[r0][i], M[r0+4][i], M[r0+8][i],
// M[r0+12][i]) -- measured on an MI355X with tools/micro/tr16_probe.hip (the ISA manual is not in this image).
// v_cvt_pknorm_i16_f32: clamp to [-1, 1], times 32767, round to nearest even; lo half = a
inline unsigned cvt_pknorm_i16(float a, float b) {
  auto one = [](float x) -> unsigned {
    if (!(x == x)) return 0u;
    x = x < -1.f ? -1.f : (x > 1.f ? 1.f : x);
    return (unsigned)(unsigned short)(short)__builtin_nearbyintf(x * 32767.f);
  };
  return one(a) | (one(b) << 16);
}
typedef __attribute__((ext_vector_type(2))) unsigned tr16_v2u;
inline tr16_v2u lds_read_tr16(const char* addr) {
  State& s = st();
  uint64_t* wx = &s.wave_x[wave_buf() * WAVE * 2];
  const int l = lane_id();
  uint64_t mine;
  memcpy(&mine, addr, 8);
  wx[l * 2] = mine;
  wave_sync();
  const int gb = l & ~15, r0 = (l & 15) >> 2, i = l & 3;
  unsigned short e[4];
  for (int j = 0; j < 4; ++j) e[j] = (unsigned short)(wx[(gb + r0 + 4 * j) * 2] >> (16 * i));
  tr16_v2u r;
  r[0] = (unsigned)e[0] | ((unsigned)e[1] << 16);
  r[1] = (unsigned)e[2] | ((unsigned)e[3] << 16);
  return r;
}

// ---- the wave's VMEM queue (dd_gcn.h) ------------------------------------------------------------------------------------------------------------
inline void lds_dma16(char* smem, unsigned ldst, const char* gsrc) {
  State& s = st();
  char* dst = smem + ldst + lane_id() * 16;
  if (s.dma_late) { s.vmq[s.cur].push_back(VmEntry{dst, gsrc}); return; }
  memcpy(dst, gsrc, 16);
  s.vmq[s.cur].push_back(VmEntry{nullptr, nullptr});
}
inline void vmem_loads_issued(int n) {
  State& s = st();
  for (int i = 0; i < n; ++i) s.vmq[s.cur].push_back(VmEntry{nullptr, nullptr});
}
inline void wait_vm(int n) {                    // s_waitcnt vmcnt(n): at most n operations outstanding, retired in issue order
  std::deque<VmEntry>& q = st().vmq[st().cur];
  while ((int)q.size() > n) {
    const VmEntry e = q.front();
    q.pop_front();
    if (e.dst) memcpy(e.dst, e.src, 16);
  }
}

}  // namespace hostemu

// ---- copies, events, stream capture / graphs --------------------------------------------------------------------------------------------------
static inline hipError_t hipMemcpy(void* d, const void* s_, size_t n, hipMemcpyKind) { ++hostemu::blocking_calls(); memcpy(d, s_, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s_, size_t n, hipMemcpyKind, hipStream_t) {
  hostemu::State& s = hostemu::st();
  if (s.capturing) { s.capturing->nodes.push_back([d, s_, n]() { memcpy(d, s_, n); }); return hipSuccess; }
  memcpy(d, s_, n);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
  hostemu::State& s = hostemu::st();
  if (s.capturing) { s.capturing->nodes.push_back([d, v, n]() { memset(d, v, n); }); return hipSuccess; }
  memset(d, v, n);
  return hipSuccess;
}
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  auto stamp = [e]() { e->t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  hostemu::State& s = hostemu::st();
  if (s.capturing) s.capturing->nodes.push_back(stamp); else stamp();
  return hipSuccess;
}
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) {
  hostemu::State& s = hostemu::st();
  if (s.capturing) return hipErrorInvalidValue;
  s.capturing = new hostemu_graph();
  return hipSuccess;
}
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) {
  hostemu::State& s = hostemu::st();
  *g = s.capturing;
  s.capturing = nullptr;
  return *g ? hipSuccess : hipErrorInvalidValue;
}
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, unsigned long long) { *e = new hostemu_graph(*g); return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t g) { delete g; return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t g, hipStream_t) {
  for (auto& n : g->nodes) n();
  const hipError_t e = hostemu::st().last_error;
  return e;
}

static inline hipError_t hipGetLastError() { const hipError_t e = hostemu::st().last_error; hostemu::st().last_error = hipSuccess; return e; }
static inline void __syncthreads() { hostemu::State& s = hostemu::st(); ++s.n_block_barriers; hostemu::rendezvous(s.block); }
static inline float __shfl_xor(float v, int mask, int /*width*/ = 64) { return hostemu::shfl_xor(v, mask); }
static inline double __shfl_xor(double v, int mask, int /*width*/ = 64) { return hostemu::shfl_xor(v, mask); }
static inline int __shfl_xor(int v, int mask, int /*width*/ = 64) { return hostemu::shfl_xor(v, mask); }
template <class T> static inline T __shfl_down(T v, int delta, int /*width*/ = 64) {
  return hostemu::shfl_from(v, hostemu::lane_id() + delta);
}
static inline void __threadfence() {}      // one OS thread runs every work-item: memory is always coherent
static inline int atomicAdd(int* p, int v) { const int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { const unsigned o = *p; *p = o > v ? o : v; return o; }
static inline unsigned hostemu_alignbyte(unsigned hi, unsigned lo, unsigned n) { return (unsigned)(((((uint64_t)hi) << 32) | lo) >> (8 * (n & 3))); }
static inline unsigned hostemu_perm(unsigned a, unsigned b, unsigned sel) {     // v_perm_b32: selector bytes 0..3 -> b, 4..7 -> a, 0x0c -> 0x00
  const uint64_t src = (((uint64_t)a) << 32) | b;
  unsigned r = 0;
  for (int i = 0; i < 4; ++i) {
    const unsigned sb = (sel >> (8 * i)) & 0xFF;
    unsigned byte = sb < 8 ? (unsigned)((src >> (8 * sb)) & 0xFF) : (sb == 0x0c ? 0u : 0xFFu);
    r |= byte << (8 * i);
  }
  return r;
}
#define __builtin_amdgcn_alignbyte(hi, lo, n) hostemu_alignbyte((hi), (lo), (n))
#define __builtin_amdgcn_perm(a, b, sel) hostemu_perm((a), (b), (sel))
static inline double atomicAdd(double* p, double v) { const double o = *p; *p = o + v; return o; }     // one OS thread
static inline float atomicAdd(float* p, float v) { const float o = *p; *p = o + v; return o; }

// uniform-by-construction at every use in the kernels (wave index, LDS addresses): the identity is exact there
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_s_barrier() __syncthreads()
#define __builtin_amdgcn_sched_group_barrier(mask, count, sync) ((void)0)
#define __builtin_amdgcn_permlane32_swap(vdst, src0, fi, bc) hostemu::permlane32_swap((vdst), (src0))
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hostemu::mfma_f16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hostemu::mfma_bf16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hostemu::mfma_f32((a), (b), (c))

#define hipLaunchKernelGGL(kernel, grid, block, smem_bytes, stream, ...) \
  (hostemu::st().last_error = hostemu::launch((grid), (block), [=]() { (kernel)(__VA_ARGS__); }))
