// LDS-ring variant of the fp64 MFMA tile engine.
//
// Same tiling and lane->k map as mfma_gemm.h, but operand fragments are not
// loaded into registers one chunk ahead: each wave owns a private ring of D
// chunk slots in LDS which it fills with `global_load_lds_dwordx4` (LDS-DMA, no
// VGPR destination) D-1 chunks ahead of the MFMAs.  A fragment is stored in the
// ring in exactly the order the MFMA lanes consume it (lane l's element at
// byte l*16, or l*8 for a real operand), so
//   * the DMA instruction is the old register load with an LDS destination
//     (wave-uniform base + lane*16, which is all LDS-DMA can do), and
//   * the read back is a lane-linear, conflict-free ds_read_b128 / ds_read_b64.
// With D = 4 the loads of a chunk have ~3 chunks of MFMA time (>= 3 x 16 x 64
// cycles) to land, which covers the ~2 us loaded L2/HBM latency seen on MI355X
// without spending VGPRs on a second/third fragment set.
//
// hipcc (ROCm 7.2) would put `s_waitcnt vmcnt(0)` in front of any C++ LDS read
// that may alias an in-flight LDS-DMA, draining the ring every chunk; the reads
// are therefore issued from inline asm behind a counted `s_waitcnt vmcnt(N)`
// (N = (D-1) x loads-per-chunk: every chunk issues the same number of loads,
// chunks past the end load a zero page).  No barrier is needed: a wave only
// reads what it loaded itself.
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

// Problem concept: as in mfma_gemm.h (A_CPLX must be true), plus
//   __device__ const cplx   *ptrA(int b, int row, int k);
//   __device__ const cplx   *ptrB(int b, int k, int col);   when B_CPLX
//   __device__ const double *ptrB(int b, int k, int col);   when !B_CPLX (cols even, 16-byte aligned pairs)
template <int TM, int TN, int D, class P, int MAP>
__global__ __launch_bounds__(256) void mfma_gemm_ring_kernel(P p, const void *zero16) {
    static_assert(P::A_CPLX, "ring engine expects a complex A operand");
    static_assert(D == 2 || D == 4, "ring depth must be 2 or 4");
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int tiles_m = (p.rows + 16 * TM - 1) / (16 * TM);
    const int tiles_n = (p.cols + 16 * TN - 1) / (16 * TN);
    const long per_batch = (long)tiles_m * tiles_n;
    const long ntask = (long)p.batch * per_batch;
    int b, tm, tn;
    if (MAP == MAP_BATCH_XCD) {
        const int nb8 = p.batch < 8 ? p.batch : 8;
        const long wg_per_group = (per_batch * ((p.batch + nb8 - 1) / nb8) + wpb - 1) / wpb;
        const int grp = blockIdx.x % nb8;
        const long slot = blockIdx.x / nb8;
        if (slot >= wg_per_group) return;
        const long t = slot * wpb + wave;
        const long bb = t / per_batch;
        b = grp + (int)bb * nb8;
        if (b >= p.batch) return;
        const int rem = (int)(t % per_batch);
        tm = rem / tiles_n; tn = rem % tiles_n;
    } else {
        const long task = (long)blockIdx.x * wpb + wave;
        if (task >= ntask) return;
        b = (int)(task / per_batch);
        const int rem = (int)(task % per_batch);
        if (MAP == MAP_ROWS_FAST) { tn = rem / tiles_m; tm = rem % tiles_m; }
        else { tm = rem / tiles_n; tn = rem % tiles_n; }
    }
    if (!p.active(b)) return;
    const int row0 = tm * 16 * TM, col0 = tn * 16 * TN;
    const int lr = lane & 15, lk = lane >> 4;

    constexpr int A_BYTES = TM * 2 * 1024;
    constexpr int B_BYTES = P::B_CPLX ? TN * 2 * 1024 : TN * 1024;
    constexpr int CHUNK = A_BYTES + B_BYTES;
    constexpr int LPC = TM * 2 + (P::B_CPLX ? TN * 2 : TN);
    constexpr int NWAIT = (D - 1) * LPC;
    static_assert(NWAIT <= 63, "vmcnt field is 6 bits");
    unsigned char *ring = smem + (size_t)wave * (D * CHUNK);
    const unsigned ring_l = lds_addr(ring);
    const int nchunks = (p.kdim + 7) >> 3;

    // real-B pair map: DMA lane L = half*32 + lp carries the doubles of MFMA lanes 2lp, 2lp+1 of sub-step `half`
    const int b_half = lane >> 5, b_lp = lane & 31;
    const int b_kk = b_lp >> 3, b_cc = (b_lp & 7) * 2;

    auto issue = [&](int c, int slot) {
        unsigned char *dst = ring + slot * CHUNK;
        const int k0 = c * 8;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int k = k0 + 2 * lk + s, row = row0 + i * 16 + lr;
                const void *src = (k < p.kdim && row < p.rows) ? (const void *)p.ptrA(b, row, k) : zero16;
                glds16(src, dst + (i * 2 + s) * 1024);
            }
        if (P::B_CPLX) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int k = k0 + 2 * lk + s, col = col0 + j * 16 + lr;
                    const void *src = (k < p.kdim && col < p.cols) ? (const void *)p.ptrB(b, k, col) : zero16;
                    glds16(src, dst + A_BYTES + (j * 2 + s) * 1024);
                }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int k = k0 + 2 * b_kk + b_half, col = col0 + j * 16 + b_cc;
                const void *src = (k < p.kdim && col < p.cols) ? (const void *)p.ptrB(b, k, col) : zero16;
                glds16(src, dst + A_BYTES + j * 1024);
            }
        }
    };

    d4_t accR[TM][TN], accI[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            accR[i][j] = (d4_t){0, 0, 0, 0};
            accI[i][j] = (d4_t){0, 0, 0, 0};
        }

#pragma unroll
    for (int c = 0; c < D - 1; ++c) issue(c, c);
    for (int c = 0; c < nchunks; ++c) {
        issue(c + D - 1, (c + D - 1) & (D - 1));
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");
        const unsigned sl = ring_l + (c & (D - 1)) * CHUNK;
        d2_t a[TM][2];
        d2_t bc[TN][2];
        double br[TN][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) a[i][s] = lds_read_b128(sl + (i * 2 + s) * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (P::B_CPLX) bc[j][s] = lds_read_b128(sl + A_BYTES + (j * 2 + s) * 1024 + lane * 16);
                else br[j][s] = lds_read_b64(sl + A_BYTES + j * 1024 + s * 512 + lane * 8);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (P::B_CPLX) {
                        accR[i][j] = mfma16(a[i][s][0], bc[j][s][0], accR[i][j]);
                        accI[i][j] = mfma16(a[i][s][0], bc[j][s][1], accI[i][j]);
                        accR[i][j] = mfma16(-a[i][s][1], bc[j][s][1], accR[i][j]);
                        accI[i][j] = mfma16(a[i][s][1], bc[j][s][0], accI[i][j]);
                    } else {
                        accR[i][j] = mfma16(a[i][s][0], br[j][s], accR[i][j]);
                        accI[i][j] = mfma16(a[i][s][1], br[j][s], accI[i][j]);
                    }
                }
    }
    // drain the zero-page loads still in flight before LDS / the wave goes away
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + i * 16 + lk + 4 * r;
                const int col = col0 + j * 16 + lr;
                if (row < p.rows && col < p.cols) p.store(b, row, col, accR[i][j][r], accI[i][j][r]);
            }
}

template <int TM, int TN, bool B_CPLX>
constexpr size_t ring_bytes_per_wave(int D) {
    return (size_t)D * (TM * 2 * 1024 + (B_CPLX ? TN * 2 * 1024 : TN * 1024));
}

template <int TM, int TN, int D, class P, int MAP>
inline hipError_t launch_mfma_gemm_ring(const P &p, hipStream_t stream, int waves_per_block, const void *zero16) {
    const long tiles_m = (p.rows + 16 * TM - 1) / (16 * TM);
    const long tiles_n = (p.cols + 16 * TN - 1) / (16 * TN);
    const long per_batch = tiles_m * tiles_n;
    const long ntask = (long)p.batch * per_batch;
    if (ntask == 0) return hipSuccess;
    long nblk = (ntask + waves_per_block - 1) / waves_per_block;
    if (MAP == MAP_BATCH_XCD) {
        const int nb8 = p.batch < 8 ? p.batch : 8;
        const long wg_per_group = (per_batch * ((p.batch + nb8 - 1) / nb8) + waves_per_block - 1) / waves_per_block;
        nblk = wg_per_group * nb8;
    }
    const size_t lds = ring_bytes_per_wave<TM, TN, P::B_CPLX>(D) * waves_per_block;
    auto kern = mfma_gemm_ring_kernel<TM, TN, D, P, MAP>;
    static size_t lds_set = 0;              // one per template instantiation: set the cap once
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set = lds;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * waves_per_block), lds, stream, p, zero16);
    return hipGetLastError();
}
