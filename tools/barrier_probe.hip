// Cost of the synchronisation primitives the latency-bound kernels are built from (gfx950), in shader cycles (s_memtime):
//   a  __syncthreads() in a loop, every wave arriving together
//   b  one wave writes 16 B per lane to LDS, barrier, every wave reads it back and the value feeds the next write (the
//      column -> search -> row chain of the Gauss-Jordan kernels)
//   c  b without the barrier inside ONE wave (write, wave_barrier, read): the LDS round trip alone
//   d  dependent DPP maxima (6 stages + readlane), the pivot search
//   e  64 dependent-free fp64 FMAs per lane (issue rate)
// hipcc --offload-arch=gfx950 -O3 tools/barrier_probe.hip -o tools/barrier_probe && tools/barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(unsigned long long *out, int iters, double *sink) {
    __shared__ double2 buf[2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long t0, t1;
    double2 v = make_double2(lane * 0.5, 1.0);
    // a
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) __syncthreads();
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = (t1 - t0) / iters;
    // b
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (wave == (i & 7) % (blockDim.x >> 6)) buf[i & 1][lane] = v;
        __syncthreads();
        const double2 r = buf[i & 1][(lane + 1) & 63];
        v.x = v.x * 0.5 + r.x; v.y += r.y * 1e-9;
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[1] = (t1 - t0) / iters;
    // c
    __syncthreads();
    if (wave == 0) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            buf[0][lane] = v;
            __builtin_amdgcn_wave_barrier();
            const double2 r = buf[0][(lane + 1) & 63];
            __builtin_amdgcn_wave_barrier();
            v.x = v.x * 0.5 + r.x; v.y += r.y * 1e-9;
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) out[2] = (t1 - t0) / iters;
    }
    // d
    __syncthreads();
    if (wave == 0) {
        unsigned key = (unsigned)(v.x * 1000.0) + lane;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            unsigned x = key + i;
#define DPPMAX(ctrl, rmask) { const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rmask, 0xf, false); x = x > t ? x : t; }
            DPPMAX(0x111, 0xf) DPPMAX(0x112, 0xf) DPPMAX(0x114, 0xf) DPPMAX(0x118, 0xf) DPPMAX(0x142, 0xa) DPPMAX(0x143, 0xc)
            key = (unsigned)__builtin_amdgcn_readlane((int)x, 63) + lane;
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) out[3] = (t1 - t0) / iters;
        v.x += key * 1e-12;
    }
    // e
    __syncthreads();
    if (wave == 0) {
        double a[16];
        for (int j = 0; j < 16; ++j) a[j] = v.x + j;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) a[j] = fma(a[j], 0.999, v.y);
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) out[4] = (t1 - t0) / iters;
        for (int j = 0; j < 16; ++j) v.x += a[j];
    }
    sink[threadIdx.x] = v.x + v.y;
}

int main() {
    unsigned long long *out; double *sink;
    hipMalloc(&out, 8 * sizeof(*out)); hipMalloc(&sink, 1024 * sizeof(double));
    for (int nt : {64, 256, 512, 1024}) {
        hipMemset(out, 0, 8 * sizeof(*out));
        hipLaunchKernelGGL(probe, dim3(1), dim3(nt), 0, 0, out, 2000, sink);
        hipLaunchKernelGGL(probe, dim3(1), dim3(nt), 0, 0, out, 2000, sink);
        unsigned long long h[8];
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("%4d threads: barrier %llu | write+barrier+read chain %llu | one-wave LDS round trip %llu | DPP search %llu | 64 fp64 FMAs %llu cycles\n",
               nt, h[0], h[1], h[2], h[3], h[4]);
    }
    return 0;
}
