// Microbenchmark: v_mfma_f64_16x16x4_f64 issue rate for the operand patterns of the library's kernels (register
// operands only, one wave per SIMD), in s_memtime ticks per MFMA.  Patterns:
//   0  16 accumulators, one A and one B register pair for every MFMA (tools/mfma_peak.hip)
//   1  HS-potential GEMM: 5 tiles x (re, im): acc[2j] += a.re * b[j], acc[2j+1] += a.im * b[j]      (shared B)
//   2  the same MFMAs ordered re-parts first, then im-parts                                              (B alternates)
//   3  3M complex tile group: P1 += ar*br, P2 += ai*bi, P3 += (ar+ai)*(br+bi), 4 tiles                   (prop_fused)
//   4  pattern 1 with one ds_read_b64 of the NEXT B operand and two integer VALU instructions between the MFMA pairs
//   5  pattern 1 with the B operands re-read from LDS every iteration, all reads ahead of the MFMAs (plain GEMM loop)
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_pattern.hip -o /tmp/mfma_pattern && /tmp/mfma_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
template <int PAT>
__global__ __launch_bounds__(256) void k(double *out, unsigned long long *clk, int iters, double s) {
    d4_t acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (d4_t){0, 0, 0, 0};
    double a[2][2], b[2][5];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) a[i][j] = s * (1 + threadIdx.x % 7 + i + 2 * j);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 5; ++j) b[i][j] = s * (2 + threadIdx.x % 5 + i + 3 * j);
    __shared__ double lds[2 * 5 * 64];
    for (int i = threadIdx.x; i < 2 * 5 * 64; i += blockDim.x) lds[i] = s * (1 + i % 13);
    __syncthreads();
    int junk = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        if (PAT == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = MF(a[0][0], b[0][0], acc[i]);
        } else if (PAT == 1) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                for (int j = 0; j < 5; ++j) { acc[2 * j] = MF(a[ss][0], b[ss][j], acc[2 * j]); acc[2 * j + 1] = MF(a[ss][1], b[ss][j], acc[2 * j + 1]); }
        } else if (PAT == 2) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[2 * j] = MF(a[ss][0], b[ss][j], acc[2 * j]);
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[2 * j + 1] = MF(a[ss][1], b[ss][j], acc[2 * j + 1]);
            }
        } else if (PAT == 4) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    acc[2 * j] = MF(a[ss][0], b[ss][j], acc[2 * j]); acc[2 * j + 1] = MF(a[ss][1], b[ss][j], acc[2 * j + 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    b[ss][j] = lds[(ss * 5 + j) * 64 + (threadIdx.x & 63)];
                    junk = junk * 3 + j; junk ^= ss;
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else if (PAT == 5) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                for (int j = 0; j < 5; ++j) b[ss][j] = lds[(ss * 5 + j) * 64 + (threadIdx.x & 63)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                for (int j = 0; j < 5; ++j) { acc[2 * j] = MF(a[ss][0], b[ss][j], acc[2 * j]); acc[2 * j + 1] = MF(a[ss][1], b[ss][j], acc[2 * j + 1]); }
        } else {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double ar = a[ss][0] + t, ai = a[ss][1] - t, br = b[ss][t], bi = b[ss][t + 1];
                    acc[3 * t] = MF(ar, br, acc[3 * t]);
                    acc[3 * t + 1] = MF(ai, bi, acc[3 * t + 1]);
                    acc[3 * t + 2] = MF(ar + ai, br + bi, acc[3 * t + 2]);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double r = 0;
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + junk;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int PAT> void run(int per_iter, double s) {
    const int nblk = 256, iters = 2000;
    double *out; hipMalloc(&out, sizeof(double) * nblk * 256);
    unsigned long long *clk; hipMalloc(&clk, 8 * nblk);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<PAT>, dim3(nblk), dim3(256), 0, 0, out, clk, iters, s);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d (operands %s): %.1f ticks per MFMA\n", PAT, s == 0.0 ? "zero" : "nonzero", (double)h[17] / ((double)iters * per_iter));
    hipFree(out); hipFree(clk);
}
int main() {
    run<0>(16, 1e-3); run<1>(20, 1e-3); run<2>(20, 1e-3); run<3>(24, 1e-3); run<4>(20, 1e-3); run<5>(20, 1e-3);
    run<0>(16, 0.0); run<1>(20, 0.0);
    return 0;
}
