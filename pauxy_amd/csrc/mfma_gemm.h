// fp64 MFMA tile engine for gfx950: one wavefront owns a TM x TN block of
// 16x16 output tiles and walks the contraction index 4 at a time with
// v_mfma_f64_16x16x4_f64.  Complex operands are interleaved (re, im) in
// memory and are split into real MFMAs in registers:
//   Cr += Ar*Br + (-Ai)*Bi ;  Ci += Ar*Bi + Ai*Br      (4 real MFMAs / fragment pair)
// Fragment maps (gfx950, f64 16x16x4):  A: lane l holds A[l&15][l>>4];
// B: lane l holds B[l>>4][l&15];  C/D register r of lane l is C[(l>>4)+4r][l&15].
#pragma once
#include "afq_internal.h"

typedef double d4_t __attribute__((ext_vector_type(4)));

__device__ inline d4_t mfma16(double a, double b, d4_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Problem concept P:
//   static constexpr bool A_CPLX, B_CPLX;
//   int batch, rows, cols, kdim;
//   __device__ bool active(int b);
//   __device__ cplx loadA(int b, int row, int k);   (row < rows, k < kdim guaranteed)
//   __device__ cplx loadB(int b, int k, int col);
//   __device__ void store(int b, int row, int col, double re, double im);
template <int TM, int TN, class P>
__global__ __launch_bounds__(256) void mfma_gemm_kernel(P p) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int tiles_m = (p.rows + 16 * TM - 1) / (16 * TM);
    const int tiles_n = (p.cols + 16 * TN - 1) / (16 * TN);
    const long ntask = (long)p.batch * tiles_m * tiles_n;
    const long task = (long)blockIdx.x * wpb + wave;
    if (task >= ntask) return;
    const int b = (int)(task / ((long)tiles_m * tiles_n));
    const int rem = (int)(task % ((long)tiles_m * tiles_n));
    // consecutive waves share the A panel (same tile row), walk tile columns
    const int tm = rem / tiles_n, tn = rem % tiles_n;
    if (!p.active(b)) return;
    const int row0 = tm * 16 * TM, col0 = tn * 16 * TN;
    const int lr = lane & 15, lk = lane >> 4;
    constexpr bool OUT_CPLX = P::A_CPLX || P::B_CPLX;

    d4_t accR[TM][TN], accI[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            accR[i][j] = (d4_t){0, 0, 0, 0};
            accI[i][j] = (d4_t){0, 0, 0, 0};
        }

    for (int k0 = 0; k0 < p.kdim; k0 += 4) {
        const int k = k0 + lk;
        const bool kok = k < p.kdim;
        cplx a[TM], bb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = row0 + i * 16 + lr;
            a[i] = (kok && row < p.rows) ? p.loadA(b, row, k) : cmake(0.0, 0.0);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + j * 16 + lr;
            bb[j] = (kok && col < p.cols) ? p.loadB(b, k, col) : cmake(0.0, 0.0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                accR[i][j] = mfma16(a[i].x, bb[j].x, accR[i][j]);
                if (P::A_CPLX && P::B_CPLX) accR[i][j] = mfma16(-a[i].y, bb[j].y, accR[i][j]);
                if (P::B_CPLX) accI[i][j] = mfma16(a[i].x, bb[j].y, accI[i][j]);
                if (P::A_CPLX) accI[i][j] = mfma16(a[i].y, bb[j].x, accI[i][j]);
            }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + i * 16 + lk + 4 * r;
                const int col = col0 + j * 16 + lr;
                if (row < p.rows && col < p.cols)
                    p.store(b, row, col, accR[i][j][r], OUT_CPLX ? accI[i][j][r] : 0.0);
            }
}

template <int TM, int TN, class P>
inline hipError_t launch_mfma_gemm(const P &p, hipStream_t stream, int waves_per_block = 4) {
    const long tiles_m = (p.rows + 16 * TM - 1) / (16 * TM);
    const long tiles_n = (p.cols + 16 * TN - 1) / (16 * TN);
    const long ntask = (long)p.batch * tiles_m * tiles_n;
    if (ntask == 0) return hipSuccess;
    const long nblk = (ntask + waves_per_block - 1) / waves_per_block;
    hipLaunchKernelGGL((mfma_gemm_kernel<TM, TN, P>), dim3((unsigned)nblk), dim3(64 * waves_per_block), 0,
                       stream, p);
    return hipGetLastError();
}
