// Steady-state cost of the one-wave register Gauss-Jordan (gj_wave.h) for an n x n matrix, warm instruction cache:
// hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -Ipauxy_amd/csrc -Iinclude tools/gj_probe.hip -o tools/gj_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "gj_wave.h"

__global__ __launch_bounds__(64) void probe(const cplx *A, int n, int reps, unsigned long long *out, cplx *dets) {
    __shared__ cplx O[32 * 32], rowk[32], piv[32];
    __shared__ int prow[32];
    const int lane = threadIdx.x;
    for (int r = 0; r < reps; ++r) {
        for (int e = lane; e < n * n; e += 64) O[e] = A[e];
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        cplx ph; int la;
        gj_wave32(O, n, lane, true, rowk, piv, prow, ph, la);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[r] = t1 - t0; dets[r] = cmake(ldexp(ph.x, la), ldexp(ph.y, la)); }
        __syncthreads();
    }
}

int main() {
    for (int n : {7, 8, 16, 25, 32}) {
        std::vector<double> a(2 * n * n);
        unsigned s = 12345;
        for (auto &x : a) { s = s * 1664525u + 1013904223u; x = (double)(s >> 8) / (1 << 24) - 0.5; }
        for (int i = 0; i < n; ++i) a[2 * (i * n + i)] += 2.0;
        cplx *A, *dets; unsigned long long *out;
        const int reps = 64;
        hipMalloc(&A, sizeof(cplx) * n * n); hipMalloc(&dets, sizeof(cplx) * reps); hipMalloc(&out, 8 * reps);
        hipMemcpy(A, a.data(), sizeof(cplx) * n * n, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, A, n, reps, out, dets);
        std::vector<unsigned long long> h(reps);
        hipMemcpy(h.data(), out, 8 * reps, hipMemcpyDeviceToHost);
        unsigned long long mn = ~0ull;
        for (int r = 8; r < reps; ++r) mn = h[r] < mn ? h[r] : mn;
        printf("n = %2d: first call %llu cycles, warm %llu cycles = %llu per pivot step\n", n, h[0], mn, mn / n);
        hipFree(A); hipFree(dets); hipFree(out);
    }
    return 0;
}
