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

template <int WHICH>
__global__ __launch_bounds__(64) void probe16(const cplx *A, int n, int reps, unsigned long long *out, cplx *dets, cplx *inv) {
    __shared__ cplx O[16 * 16], rowk[32], piv[32];
    __shared__ int prow[32];
    const int lane = threadIdx.x;
    for (int r = 0; r < reps; ++r) {
        for (int e = lane; e < n * n; e += 64) O[e] = A[e];
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        cplx ph; int la;
        if (WHICH == 0) gj_wave16(O, n, lane, true, rowk, piv, prow, ph, la);
        else gj_wave16q(O, n, lane, true, rowk, piv, prow, ph, la);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[r] = t1 - t0; dets[r] = cmake(ldexp(ph.x, la), ldexp(ph.y, la)); }
        __syncthreads();
    }
    for (int e = lane; e < n * n; e += 64) inv[e] = O[e];
}

static void compare16() {
    for (int n : {1, 3, 4, 5, 8, 11, 15, 16}) {
        std::vector<double> a(2 * n * n);
        unsigned s = 777 + n;
        for (auto &x : a) { s = s * 1664525u + 1013904223u; x = (double)(s >> 8) / (1 << 24) - 0.5; }
        cplx *A, *dets, *inv; unsigned long long *out;
        const int reps = 32;
        hipMalloc(&A, sizeof(cplx) * n * n); hipMalloc(&dets, sizeof(cplx) * reps); hipMalloc(&out, 8 * reps);
        hipMalloc(&inv, sizeof(cplx) * n * n);
        hipMemcpy(A, a.data(), sizeof(cplx) * n * n, hipMemcpyHostToDevice);
        std::vector<double> i0(2 * n * n), i1(2 * n * n);
        double d0[2], d1[2];
        unsigned long long t[2];
        for (int w = 0; w < 2; ++w) {
            if (w == 0) hipLaunchKernelGGL(probe16<0>, dim3(1), dim3(64), 0, 0, A, n, reps, out, dets, inv);
            else hipLaunchKernelGGL(probe16<1>, dim3(1), dim3(64), 0, 0, A, n, reps, out, dets, inv);
            std::vector<unsigned long long> h(reps);
            hipMemcpy(h.data(), out, 8 * reps, hipMemcpyDeviceToHost);
            hipMemcpy(w ? i1.data() : i0.data(), inv, sizeof(cplx) * n * n, hipMemcpyDeviceToHost);
            hipMemcpy(w ? d1 : d0, dets + reps - 1, sizeof(cplx), hipMemcpyDeviceToHost);
            unsigned long long mn = ~0ull;
            for (int r = 8; r < reps; ++r) mn = h[r] < mn ? h[r] : mn;
            t[w] = mn;
        }
        double err = 0, ref = 0;
        for (int e = 0; e < 2 * n * n; ++e) { err = fmax(err, fabs(i0[e] - i1[e])); ref = fmax(ref, fabs(i0[e])); }
        printf("n = %2d: halves layout %llu cycles, quad layout %llu cycles (%llu per pivot); inverse differs by %.2e of %.2e; det %.15e%+.15ei vs %.15e%+.15ei\n",
               n, t[0], t[1], t[1] / n, err, ref, d0[0], d0[1], d1[0], d1[1]);
        hipFree(A); hipFree(dets); hipFree(out); hipFree(inv);
    }
}

int main() {
    compare16();
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
