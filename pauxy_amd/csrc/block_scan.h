// Work-group prefix scan shared by the comb planners (k_small.hip, k_comm.hip).
#pragma once
#include "afq_internal.h"

// exclusive prefix sum of one value per thread over a 256-thread work-group (wave shuffles + 4 wave
// totals through LDS); *total receives the sum of all 256 values
template <class T> __device__ inline T block_excl_scan256(T v, T *wtot, T *total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();                 // wtot may still be read from a previous call
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    T base = 0, tot = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) base += wtot[w]; tot += wtot[w]; }
    *total = tot;
    return base + inc - v;
}
