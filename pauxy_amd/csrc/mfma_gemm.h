// fp64 MFMA tile engine for gfx950: one wavefront owns a TM x TN block of
// 16x16 output tiles and walks the contraction index 8 at a time with two
// v_mfma_f64_16x16x4_f64 per (tile, operand pair).  Complex operands are
// interleaved (re, im) in memory and are split into real MFMAs in registers:
//   Cr += Ar*Br + (-Ai)*Bi ;  Ci += Ar*Bi + Ai*Br      (4 real MFMAs / fragment pair)
// Fragment maps (gfx950, f64 16x16x4):  A: lane l holds A[l&15][l>>4];
// B: lane l holds B[l>>4][l&15];  C/D register r of lane l is C[(l>>4)+4r][l&15].
// Inside a chunk of 8 contraction indices lane group g = l>>4 takes k = k0+2g
// for the first MFMA and k0+2g+1 for the second, so that a lane's two A
// elements are adjacent in memory (row-major A): 16 rows x 128 contiguous bytes
// per wavefront load.  Fragments of chunk c+1 are loaded into a second register
// set before the MFMAs of chunk c issue (software prefetch), so the L2 latency
// hides under >= 16 x 64 MFMA cycles.
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
struct FragSet {
    cplx a[TM][2], b[TN][2];
    __device__ inline void load(const P &p, int bt, int row0, int col0, int k0, int lr, int lk) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k = k0 + 2 * lk + s;
            const bool kok = k < p.kdim;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = row0 + i * 16 + lr;
                a[i][s] = (kok && row < p.rows) ? p.loadA(bt, row, k) : cmake(0.0, 0.0);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = col0 + j * 16 + lr;
                b[j][s] = (kok && col < p.cols) ? p.loadB(bt, k, col) : cmake(0.0, 0.0);
            }
        }
    }
};

// Task -> (batch, tile row, tile column) maps.  Workgroups are dealt round-robin
// over the 8 XCDs (blockIdx % 8 labels the XCD, each with a private 4 MiB L2),
// so the map decides which operand panels share an L2 / an L1:
//   MAP_COLS_FAST  consecutive waves walk tile columns of one tile row (share the A panel)
//   MAP_ROWS_FAST  consecutive waves walk tile rows of one tile column (share the B panel)
//   MAP_BATCH_XCD  batch b runs on XCD b % 8: each batch's operand slices stay in one L2
//   MAP_COLPANEL_XCD  (work-group engine, batch == 1) column panel c and all its row tiles run on XCD c % 8
//   MAP_BATCH_XCD_ROWS  (work-group engine) as MAP_BATCH_XCD, tile rows fastest inside a batch: the row tiles of one
//                       B column panel run back to back on one XCD
//   MAP_COLTILE_SLOW  (work-group engine) the column tile is the SLOWEST index of the whole launch (then the batch, then the row
//                  tile): a problem that leaves out whole column tiles of most batches (BTILE_SKIP: the beta columns of
//                  closed-shell walkers) keeps its live work-groups contiguous, i.e. spread evenly over the XCDs
enum { MAP_COLS_FAST = 0, MAP_ROWS_FAST = 1, MAP_BATCH_XCD = 2, MAP_COLPANEL_XCD = 3, MAP_BATCH_XCD_ROWS = 4, MAP_COLTILE_SLOW = 5 };
// optional problem trait: static constexpr bool INACTIVE_COPY = true -- inactive_tile(b, row0, nrows, col0, ncols, t, nthr)
// is called by the nthr threads that would have computed the tile of an inactive batch (both GEMM engines)
template <class P, class = void> struct gemm_inactive_copy { static constexpr bool value = false; };
template <class P> struct gemm_inactive_copy<P, decltype((void)P::INACTIVE_COPY)> { static constexpr bool value = P::INACTIVE_COPY; };

template <int TM, int TN, class P, int MAP = MAP_COLS_FAST>
__global__ __launch_bounds__(512) void mfma_gemm_kernel(P p) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int tiles_m = (p.rows + 16 * TM - 1) / (16 * TM);
    const int tiles_n = (p.cols + 16 * TN - 1) / (16 * TN);
    const long per_batch = (long)tiles_m * tiles_n;
    const long ntask = (long)p.batch * per_batch;
    int b, tm, tn;
    if (MAP == MAP_BATCH_XCD) {
        // blockIdx = slot * nb8 + (b % nb8): with nb8 = min(batch, 8) consecutive
        // workgroups (consecutive XCDs) take different batches
        const int nb8 = p.batch < 8 ? p.batch : 8;
        const long wg_per_group = (per_batch * ((p.batch + nb8 - 1) / nb8) + wpb - 1) / wpb;
        const int grp = blockIdx.x % nb8;
        const long slot = blockIdx.x / nb8;
        if (slot >= wg_per_group) return;
        const long t = slot * wpb + wave;          // task index inside this XCD group
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
    if (!p.active(b)) {
        // optional problem trait INACTIVE_COPY: the output tile of an inactive batch is not left alone but copied through
        // (ping-pong buffers: a dead walker's columns have to arrive in the destination too)
        if constexpr (gemm_inactive_copy<P>::value) p.inactive_tile(b, tm * 16 * TM, 16 * TM, tn * 16 * TN, 16 * TN, lane, 64);
        return;
    }
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

    // Register prefetch one k-chunk ahead with two explicit fragment sets and the loop unrolled by two.  (Written
    // as cur/nxt with a copy at the end of the iteration, the compiler folds the copy away and loads, waits for and
    // consumes a chunk's fragments inside one iteration.)
    FragSet<TM, TN, P> fA, fB;
    auto mfmas = [&](const FragSet<TM, TN, P> &cur) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    accR[i][j] = mfma16(cur.a[i][s].x, cur.b[j][s].x, accR[i][j]);
                    if (P::A_CPLX && P::B_CPLX) accR[i][j] = mfma16(-cur.a[i][s].y, cur.b[j][s].y, accR[i][j]);
                    if (P::B_CPLX) accI[i][j] = mfma16(cur.a[i][s].x, cur.b[j][s].y, accI[i][j]);
                    if (P::A_CPLX) accI[i][j] = mfma16(cur.a[i][s].y, cur.b[j][s].x, accI[i][j]);
                }
        __builtin_amdgcn_sched_barrier(0);
    };
    fA.load(p, b, row0, col0, 0, lr, lk);
    for (int k0 = 0; k0 < p.kdim; k0 += 16) {
        if (k0 + 8 < p.kdim) fB.load(p, b, row0, col0, k0 + 8, lr, lk);
        mfmas(fA);
        if (k0 + 8 < p.kdim) {
            if (k0 + 16 < p.kdim) fA.load(p, b, row0, col0, k0 + 16, lr, lk);
            mfmas(fB);
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

template <int TM, int TN, class P, int MAP = MAP_COLS_FAST>
inline hipError_t launch_mfma_gemm(const P &p, hipStream_t stream, int waves_per_block = 4) {
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
    hipLaunchKernelGGL((mfma_gemm_kernel<TM, TN, P, MAP>), dim3((unsigned)nblk), dim3(64 * waves_per_block), 0,
                       stream, p);
    return hipGetLastError();
}

// wave-tasks a (TM, TN) tiling produces, for picking a shape that fills the
// 1024 SIMDs of the chip in whole rounds
inline long mfma_gemm_tasks(int batch, int rows, int cols, int TM, int TN) {
    return (long)batch * ((rows + 16 * TM - 1) / (16 * TM)) * ((cols + 16 * TN - 1) / (16 * TN));
}
