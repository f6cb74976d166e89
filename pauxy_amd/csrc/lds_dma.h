// LDS-DMA (global_load_lds) and raw LDS read primitives shared by the work-group GEMM engine, the fused
// propagator and the Green's-function kernels.
//
// hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of any C++ LDS read that may alias an in-flight
// LDS-DMA, draining the operand ring every chunk; reads of DMA-filled slots are therefore issued from inline
// asm behind a counted `s_waitcnt vmcnt(N)`.  A fragment is stored in LDS in exactly the order the MFMA lanes
// consume it (lane l's element at byte l*16, or l*8 for a real operand), so the DMA instruction is a plain
// register load with an LDS destination (wave-uniform base + lane*16, which is all LDS-DMA can do) and the
// read back is a lane-linear, conflict-free ds_read_b128 / ds_read_b64.
#pragma once
#include "mfma_gemm.h"

typedef double d2_t __attribute__((ext_vector_type(2)));

__device__ inline void glds16(const void *g, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}
__device__ inline unsigned lds_addr(const void *p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)p;
}
__device__ inline d2_t lds_read_b128(unsigned addr) {
    d2_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ inline double lds_read_b64(unsigned addr) {
    double v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
