// grid_sync_probe2.hip -- the two grid-barrier designs VERDICT r3 (weak #5, item 4) says grid_sync_probe.hip never measured:
//   (d) the DATA path of a fence-less barrier: producers publish with agent-scope (sc1, write-through) stores, consumers read with agent-scope (sc1)
//       loads -- no __threadfence(), i.e. no L2 write-back / invalidate -- ordered behind the barrier's own atomics by `s_waitcnt vmcnt(0)`;
//       CHECKED for stale tokens (variant (c) of the first probe exchanged data with PLAIN stores / loads, so of course it was stale);
//   (e) a two-level barrier: the workgroups of one XCD (s_getreg HW_REG_XCC_ID) meet on a counter that lives in THEIR L2 (workgroup-scope atomics:
//       executed at the XCD's own L2, no trip to the memory side), the last arriver of each XCD meets the other seven on ONE agent-scope counter
//       and then releases its XCD through a local flag.  8 spinners on the shared line instead of 418.
// Each with the same checked token exchange (block b publishes slot b, reads slot b + 1 after the barrier) and a timing pass of N barriers in a row.
// Every spin is bounded: a wrong barrier reports an error instead of hanging the GPU.
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-result tools/micro/grid_sync_probe2.hip -o build_variants/grid_sync_probe2 && build_variants/grid_sync_probe2
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr unsigned SPIN_MAX = 1u << 17;      // ~20 ms: a barrier that does not work ends the kernel quickly

// Agent scope: an sc1 load (misses the CU's vector L1, coherent across XCDs).  "Workgroup" scope is used here for the counters that live in ONE
// XCD's L2: a workgroup-scope LOAD may hit the CU's own L1 (the scope only promises coherence inside a workgroup), so the poll is an atomic
// read-modify-write of zero -- atomics always execute at the L2, which is shared by the XCD's CUs (first version of this probe polled with
// sc0 loads and timed out: profiles/history/r04_call2_grid_sync_probe2.txt).
template <int SCOPE> __device__ __forceinline__ bool spin_until(unsigned* ctr, unsigned target) {
  for (unsigned s = 0; s < SPIN_MAX; ++s) {
    const unsigned v = SCOPE == __HIP_MEMORY_SCOPE_WORKGROUP ? __hip_atomic_fetch_add(ctr, 0u, __ATOMIC_RELAXED, SCOPE) : __hip_atomic_load(ctr, __ATOMIC_RELAXED, SCOPE);
    if (v >= target) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

struct Ctl {                 // every counter on its own 256-byte line
  unsigned flat[64];         // [0] the one-level counter
  unsigned top[64];          // [0] the eight XCD leaders' counter
  unsigned reg[8][64];       // [x][0] workgroups registered on XCD x (agent scope, once)
  unsigned loc[8][64];       // [x][0] arrivals on XCD x (workgroup scope: lives in that XCD's L2)
  unsigned flag[8][64];      // [x][0] release flag of XCD x (workgroup scope)
};

// MODE 0: flat agent-scope counter (as probe 1 (c)), data through sc1 stores / loads     -> (d)
// MODE 1: two-level barrier, data through sc1 stores / loads                            -> (d) + (e)
// MODE 2: two-level barrier, PLAIN data stores / loads (expected stale: the control)
template <int MODE>
__global__ void __launch_bounds__(256) barrier_kernel(Ctl* c, unsigned* slots, int n, int* err, int check) {
  const unsigned nb = gridDim.x, b = blockIdx.x;
  unsigned xcc = 0, nloc = 0;
  __shared__ unsigned s_nloc;
  if (MODE != 0) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    if (threadIdx.x == 0) {
      // registration: how many workgroups run on my XCD (one flat barrier, once per kernel)
      __hip_atomic_fetch_add(&c->reg[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&c->flat[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!spin_until<__HIP_MEMORY_SCOPE_AGENT>(&c->flat[32], nb)) *err = 1;
      s_nloc = __hip_atomic_load(&c->reg[xcc][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    nloc = s_nloc;
  }
  auto barrier = [&](unsigned round, int which) {          // round = 1, 2, ...; `which` selects an independent counter set (0 / 1)
    __syncthreads();
    if (threadIdx.x == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my published data has left (sc1 stores are write-through)
      if (MODE == 0) {
        __hip_atomic_fetch_add(&c->flat[which], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!spin_until<__HIP_MEMORY_SCOPE_AGENT>(&c->flat[which], round * nb)) *err = 1;
      } else {
        const unsigned old = __hip_atomic_fetch_add(&c->loc[xcc][which], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (old + 1 == round * nloc) {                     // last arriver of this XCD: meet the other XCDs' leaders, then release mine
          __hip_atomic_fetch_add(&c->top[which], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // (XCDs that run no workgroup of this launch never arrive: count the XCDs that registered)
          unsigned nx = 0;
          for (int x = 0; x < 8; ++x) nx += __hip_atomic_load(&c->reg[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
          if (!spin_until<__HIP_MEMORY_SCOPE_AGENT>(&c->top[which], round * nx)) *err = 1;
          __hip_atomic_exchange(&c->flag[xcc][which], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (an atomic: executed at the L2)
        } else {
          if (!spin_until<__HIP_MEMORY_SCOPE_WORKGROUP>(&c->flag[xcc][which], round)) *err = 1;
        }
      }
    }
    __syncthreads();
  };
  for (int it = 0; it < n; ++it) {
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;      // somebody timed out: everybody leaves
    const unsigned tok = (unsigned)(it * 131 + b);
    if (threadIdx.x == 0) {
      if (MODE == 2) slots[b * 64] = tok;
      else __hip_atomic_store(&slots[b * 64], tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    barrier((unsigned)it + 1, 0);
    if (check) {
      if (threadIdx.x == 0) {
        const unsigned o = (b + 1) % nb;
        const unsigned got = (MODE == 2) ? slots[o * 64] : __hip_atomic_load(&slots[o * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (got != (unsigned)(it * 131 + o)) atomicAdd(err + 1, 1);
      }
      barrier((unsigned)it + 1, 1);        // nobody overwrites a slot before its reader has looked
    }
  }
}

template <int MODE> float run(int nb, int n, int check, int* herr) {
  Ctl* c; unsigned* slots; int* err;
  hipMalloc(&c, sizeof(Ctl)); hipMalloc(&slots, (size_t)nb * 256); hipMalloc(&err, 8);
  hipMemset(c, 0, sizeof(Ctl)); hipMemset(slots, 0, (size_t)nb * 256); hipMemset(err, 0, 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  void* args[] = {&c, &slots, &n, &err, &check};
  hipEventRecord(a, 0);
  hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<void*>(barrier_kernel<MODE>), dim3(nb), dim3(256), args, 0, 0);   // (co-residency guaranteed)
  hipEventRecord(b, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost);
  if (MODE == 1 && check) {                    // how the workgroups registered: workgroups per XCC_ID, local / top / flag counters of the last round
    Ctl hc; hipMemcpy(&hc, c, sizeof(Ctl), hipMemcpyDeviceToHost);
    printf("    registered per XCC_ID:");
    for (int x = 0; x < 8; ++x) printf(" %u", hc.reg[x][0]);
    printf("   loc:"); for (int x = 0; x < 8; ++x) printf(" %u", hc.loc[x][0]);
    printf("   top %u   flag:", hc.top[0]); for (int x = 0; x < 8; ++x) printf(" %u", hc.flag[x][0]);
    printf("\n"); fflush(stdout);
  }
  if (e != hipSuccess) { herr[0] = -1; printf("  launch error: %s\n", hipGetErrorString(e)); }
  hipFree(c); hipFree(slots); hipFree(err);
  return ms;
}

int main() {
  const char* names[3] = {"(d) flat agent-scope counter, data by sc1 store / sc1 load, no fences", "(d)+(e) two-level barrier (per-XCD L2 counter + 8 leaders), data by sc1",
                          "(e) two-level barrier, PLAIN data stores / loads (control: stale expected)"};
  for (int nb : {256, 418, 512}) {
    printf("== %d workgroups of 256 threads\n", nb); fflush(stdout);
    for (int mode = 0; mode < 3; ++mode) {
      int herr[2] = {0, 0};
      float msc = mode == 0 ? run<0>(nb, 50, 1, herr) : mode == 1 ? run<1>(nb, 50, 1, herr) : run<2>(nb, 50, 1, herr);
      const int e0 = herr[0], stale = herr[1];
      const int n = 400;
      float t0 = mode == 0 ? run<0>(nb, 0, 0, herr) : mode == 1 ? run<1>(nb, 0, 0, herr) : run<2>(nb, 0, 0, herr);
      float t1 = mode == 0 ? run<0>(nb, n, 0, herr) : mode == 1 ? run<1>(nb, n, 0, herr) : run<2>(nb, n, 0, herr);
      printf("  %-78s %7.2f us per barrier   (spin timeouts %d, stale tokens in 50 checked rounds %d; checked pass %.2f ms)\n", names[mode], (t1 - t0) * 1e3f / n,
             e0 | herr[0], stale, msc);
      fflush(stdout);
    }
  }
  return 0;
}
