// dd_gcn.h -- the gfx950 constructs of the convolution kernels that have no C++ meaning (LDS-DMA, s_waitcnt, the LDS base address), behind
// macros.  The device build gets exactly the instruction sequences the kernels carried inline before (the code objects are byte-identical);
// tests/host_emul compiles the same kernel sources for the HOST with -DDD_HOST_EMULATION and maps the macros onto a model of the wave's VMEM
// queue (tests/host_emul/hip/hip_runtime.h): s_waitcnt vmcnt(N) retires all but the newest N entries in issue order, an LDS-DMA lands either
// at issue or only when a wait retires it -- the two extremes the hardware's timing lies between.
#pragma once

#ifdef DD_HOST_EMULATION

#define DD_DYN_SMEM(name) extern __attribute__((aligned(16))) char name[]          /* the harness defines dd::smem */
#define DD_PIN_VGPR(x) ((void)(x))
#define DD_LDS_BASE(smem) 0u
#define DD_LDS_DMA16(smem, gsrc, ldst) hostemu::lds_dma16((smem), (ldst), (gsrc))
#define DD_WAIT_VM(n) hostemu::wait_vm(n)
#define DD_WAIT_VM_LGKM0(n) hostemu::wait_vm(n)
#define DD_WAIT_LGKM0() ((void)0)
#define DD_SCHED_FENCE() ((void)0)
#define DD_VMEM_LOADS_ISSUED(n) hostemu::vmem_loads_issued(n)
#define DD_GLOBAL_STORE16_UNTRACKED(ptr, v) (*reinterpret_cast<float4*>(ptr) = (v))
typedef hostemu::tr16_v2u dd_u32x2_t;
#define DD_CVT_PKNORM_I16(a, b) hostemu::cvt_pknorm_i16((a), (b))
#define DD_LDS_READ_TR16(smem, byte_off) hostemu::lds_read_tr16((smem) + (byte_off))

#else

// the workgroup's dynamic LDS (size given at launch)
#define DD_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
// "this value lives in a VGPR from here on": an empty asm the optimiser cannot look through (keeps loads above the selects that consume them)
#define DD_PIN_VGPR(x) asm volatile("" : "+v"(x))
#define DD_LDS_BASE(smem) ((unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem))
// one wave-instruction: lane l copies 16 bytes from its global address gsrc to LDS address ldst (wave-uniform, via M0) + 16 * l.
// Inline asm on purpose: with the __builtin form hipcc treats the DMA as an LDS write that may alias every later ds_read and drains
// vmcnt(0) right behind it.  hipcc does not count asm VMEM operations: every wait for a DMA is one of the explicit macros below.
#define DD_LDS_DMA16(smem, gsrc, ldst)                                                                                          \
  do {                                                                                                                          \
    unsigned keep_m0_;                                                                                                          \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"       \
                 : "=&s"(keep_m0_) : "v"(gsrc), "s"(ldst) : "memory");                                                          \
  } while (0)
#define DD_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define DD_WAIT_VM_LGKM0(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(n) : "memory")
#define DD_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// nothing is scheduled across this point: keeps a burst of independent loads in front of the first code that waits for one of them
#define DD_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// A 16-byte global store the compiler does NOT see (inline asm).  hipcc keeps ONE vmcnt for loads and stores; once a store is pending beside
// loads it can no longer count ("counter out of order") and waits vmcnt(0) for the next load it needs -- which drains every prefetched load
// of a software pipeline.  Hidden from it, the store only makes its counted waits for loads more conservative (the hardware counter is
// higher than it assumes); the data is in memory by the end of the kernel like any other store.  The trailing `s_nop 1` keeps the two wait
// states gfx940+ needs between a VMEM store of more than 64 bits and a VALU write to its data registers: the hazard recogniser does not
// look into asm, and the next loop trip may recompute the same VGPRs right behind the store.
typedef __attribute__((ext_vector_type(4))) float dd_f32x4_t;
#define DD_GLOBAL_STORE16_UNTRACKED(ptr, v)                                                                                     \
  do {                                                                                                                          \
    const float4 f4_ = (v);                                                                                                     \
    const dd_f32x4_t v_ = {f4_.x, f4_.y, f4_.z, f4_.w};                                                                         \
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(ptr), "v"(v_) : "memory");                                   \
  } while (0)
// v_cvt_pknorm_i16_f32: two floats in [-1, 1] -> two signed normalised 16-bit integers round(x * 32767), packed (lo = a)
#define DD_CVT_PKNORM_I16(a, b) __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_i16((a), (b)))
// bookkeeping for the host model only: n ordinary global loads were just issued by this wave (they count in vmcnt)
#define DD_VMEM_LOADS_ISSUED(n) ((void)0)
// ds_read_b64_tr_b16: the lane's 8-byte-aligned LDS address -> 4 x 16 bit, transposed inside each group of 16 lanes (see dd_wgrad2.hip)
typedef __attribute__((ext_vector_type(4))) short dd_s16x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned dd_u32x2_t;
#define DD_LDS_READ_TR16(smem, byte_off)                                                                                              \
  __builtin_bit_cast(dd_u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16(                                                             \
      (__attribute__((address_space(3))) dd_s16x4_t*)((__attribute__((address_space(3))) char*)(smem) + (byte_off))))

#endif
