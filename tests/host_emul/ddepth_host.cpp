// tests/host_emul/ddepth_host.cpp -- TEST INFRASTRUCTURE: the globals of the host emulation of the WHOLE library.  Every source file of
// diffusiondepth_amd/csrc (the C ABI in dd_api.cpp and dd_dcn.hip included) is compiled for the host with -DDD_HOST_EMULATION on top of
// hip/hip_runtime.h and linked with this file into build/host_emul/libddepth_hostemu_<hash>.so: the complete C ABI of include/ddepth.h and
// include/ddepth_dcn.h then runs on the CPU (device memory = host memory, streams execute at once, a captured graph replays its recorded
// launches), work-item by work-item under the adversarial schedules described in hip_runtime.h.  tests/test_library_host_emulation.py drives
// it through ctypes and compares with the oracle.  It is never loaded by the product (diffusiondepth_amd/backend.py accepts GPU tensors only).
#include <hip/hip_runtime.h>

hostemu::Idx3 threadIdx, blockIdx, blockDim, gridDim;

// void hostemu_switch(void** save_sp, void* load_sp): park the caller (callee-saved registers on its stack, stack pointer to *save_sp) and
// continue the context whose stack pointer is load_sp
asm(R"(
    .text
    .globl hostemu_switch
    .type hostemu_switch,@function
hostemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hostemu_switch, .-hostemu_switch
)");

namespace dd {
alignas(16) char smem[160 * 1024];      // the workgroup's dynamic LDS (DD_DYN_SMEM in dd_gcn.h binds to this)
}

extern "C" {
// 0 = the first wave runs ahead as far as the workgroup barriers allow, 1 = the last wave
void emu_set_order(int order) { hostemu::st().order = order; }
// 0 = an LDS-DMA lands when it is issued, 1 = only when an s_waitcnt retires it
void emu_set_dma_late(int late) { hostemu::st().dma_late = late; }
unsigned long emu_launch_count() { return hostemu::st().n_launches; }
// host-blocking runtime calls so far (hipDeviceSynchronize / hipStreamSynchronize / hipEventSynchronize / blocking hipMemcpy / hipMalloc / hipFree)
unsigned long emu_blocking_calls() { return hostemu::blocking_calls(); }
}
