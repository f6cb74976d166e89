// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate and in-kernel clock on gfx950
// (register operands only).  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, unsigned long long *clk, int iters, double a0, double b0) {
    d4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4_t){0, 0, 0, 0};
    double a = a0 * (1.0 + (threadIdx.x * 2654435761u % 1000) * 1e-3), b = b0 * (1.0 + (threadIdx.x * 40503u % 977) * 1e-3);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NACC> void run(int waves_per_simd, int iters, int reps, double a0) {
    const int nblk = 256 * waves_per_simd;
    double *out; hipMalloc(&out, sizeof(double) * nblk * 256);
    unsigned long long *clk; hipMalloc(&clk, 16 * nblk);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(nblk), dim3(256), 0, 0, out, clk, 10, a0, 1.0);
    hipDeviceSynchronize();
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(nblk), dim3(256), 0, 0, out, clk, iters, a0, 1.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * nblk);
        hipMemcpy(h.data(), clk, 16 * nblk, hipMemcpyDeviceToHost);
        std::vector<double> ghz(nblk), cyc(nblk);
        for (int i = 0; i < nblk; ++i) { ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; cyc[i] = (double)h[2 * i]; }
        std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
        const double flops = (double)nblk * 4 * iters * NACC * 2048.0;
        if (r == 0 || r == reps - 1 || r == reps / 2)
            printf("NACC=%2d w/SIMD=%d a0=%g rep %2d: %.3f ms %.1f TFLOP/s  clock %.2f GHz  %.1f shader-cycles per MFMA per SIMD\n",
                   NACC, waves_per_simd, a0, r, ms, flops / ms / 1e9, ghz[nblk / 2],
                   cyc[nblk / 2] / ((double)iters * NACC * waves_per_simd));
    }
    hipFree(out); hipFree(clk);
}
int main() {
    run<16>(1, 1500, 40, 1.0);      // ~1.4 ms bursts back to back
    run<16>(1, 1500, 40, 0.0);      // zero operands
    run<16>(2, 1500, 20, 1.0);
    run<16>(1, 150000, 3, 1.0);     // long
    return 0;
}
