// Work-group-tiled fp64 MFMA GEMM with a shared LDS ring.
//
// WM x WN waves form one work-group; wave (wm, wn) owns TM x TN 16x16 tiles, so
// the work-group tile is (16 TM WM) x (16 TN WN).  Per chunk of 8 contraction
// indices every operand fragment of the work-group tile is brought into LDS
// ONCE (global_load_lds_dwordx4, fragment order, see lds_dma.h) and read
// by all the waves that need it: an A fragment by the WN waves of its tile row,
// a B fragment by the WM waves of its tile column.  That multiplies the flops
// per byte fetched from L2 by ~WM (resp. WN) over the one-wave-one-block
// engines, which is what the K-long, output-small contractions (force bias:
// 256 x 5000 x 500) need: they are L2-miss-bandwidth bound otherwise.
//
// Pipeline (ring of D slots, one raw s_barrier per chunk):
//   iteration c:  s_waitcnt vmcnt((D-2) * LPW)   own DMA of chunk c has landed
//                 s_barrier                      everyone's DMA of chunk c has landed and
//                                                everyone is done reading chunk c-1
//                 issue DMA of chunk c+D-1 into slot (c-1) % D
//                 ds_read fragments of chunk c (inline asm), s_waitcnt lgkmcnt(0)
//                 MFMAs
// Every wave issues exactly LPW loads per chunk (its share of the A and B
// fragments, padded with zero-page loads into a scratch KB) so the counted
// vmcnt is a compile-time constant.
#pragma once
#include "lds_dma.h"


// K3M: complex x complex products by the 3-multiplication (Karatsuba) form
//   P1 = Ar Br, P2 = Ai Bi, P3 = (Ar+Ai)(Br+Bi);  Cr = P1 - P2, Ci = P3 - P1 - P2
// (3 real MFMAs per fragment pair instead of 4; error is bounded normwise, ~1e-16 |A||B|).
// optional problem trait: static constexpr bool A_CONJ = true contracts with conj(A)
template <class P, class = void> struct gemm_conj_a { static constexpr bool value = false; };
template <class P> struct gemm_conj_a<P, decltype((void)P::A_CONJ)> { static constexpr bool value = P::A_CONJ; };

// optional problem trait: static constexpr bool INCR = true -- both operands are affine in the contraction index
// (baseA(b,row) + k kstepA(), baseB(b,col) + k kstepB(b), valid for k < klimit(b), rowok(b,row), colok(b,col)): the refill then
// advances one pointer per fragment instead of re-deriving address and bounds from (b, row, k) every chunk
template <class P, class = void> struct gemm_incr { static constexpr bool value = false; };
template <class P> struct gemm_incr<P, decltype((void)P::INCR)> { static constexpr bool value = P::INCR; };

#ifdef AFQ_TUNING
// tuning builds: when set (hipMemcpyToSymbol), work-groups 0-63 of every ring GEMM leave their s_memtime phases here
__device__ unsigned long long *afq_gemm_ts = nullptr;
// tuning builds: timing ablations of the plain chunk loop (WRONG results): bit 0 no MFMAs, bit 1 no ring refill,
// bit 2 no fragment reads, bit 3 no chunk barrier, bit 4 no output stores
__device__ int afq_gemm_abl = 0;
#define GEMM_UNLESS(bits) if (!(abl & (bits)))
#else
#define GEMM_UNLESS(bits)
#endif
// optional problem trait: static constexpr bool KCUT = true -- kcut(b, col0, ncols) is the contraction length the work-group
// tile at columns [col0, col0 + ncols) of batch b needs (<= kdim; B is zero beyond it for these columns: triangular B)
template <class P, class = void> struct gemm_kcut { static constexpr bool value = false; };
template <class P> struct gemm_kcut<P, decltype((void)P::KCUT)> { static constexpr bool value = P::KCUT; };
// optional problem trait: static constexpr bool ROWDOT = true -- the product itself is not stored; every output element is
// multiplied with dot_operand(b, row, col) and the sums over the 16 columns of a tile go to store_dot(b, row, tile, re, im)
// (a quadratic form x^T A x evaluated as sum_col (x^T A)[col] x[col] without the round trip of x^T A through memory)
template <class P, class = void> struct gemm_rowdot { static constexpr bool value = false; };
template <class P> struct gemm_rowdot<P, decltype((void)P::ROWDOT)> { static constexpr bool value = P::ROWDOT; };
// optional problem trait: static constexpr bool COLDOT = true -- besides being stored, every output element is multiplied
// with coldot_coef(b, row, col) and summed over the rows of the wave's tile block: store_coldot(b, part, col, re, im) with
// part = first row of the block / (16 TM) receives one partial sum per column (diag(C^T X) without reading X back)
template <class P, class = void> struct gemm_coldot { static constexpr bool value = false; };
template <class P> struct gemm_coldot<P, decltype((void)P::COLDOT)> { static constexpr bool value = P::COLDOT; };
template <int CTRL> __device__ inline double gemm_dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over each group of 16 consecutive lanes, left in all of them: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ inline double gemm_row16_sum(double v) {
    v += gemm_dpp_f64<0xB1>(v);
    v += gemm_dpp_f64<0x4E>(v);
    v += gemm_dpp_f64<0x141>(v);
    v += gemm_dpp_f64<0x140>(v);
    return v;
}
// optional problem trait: static constexpr bool A_REAL = true -- every imaginary part of A is exactly zero (the
// caller has checked): the products with it are not issued, 2 real multiplications per element pair instead of 3 / 4.
// A is still stored and staged as complex numbers.
template <class P, class = void> struct gemm_areal { static constexpr bool value = false; };
template <class P> struct gemm_areal<P, decltype((void)P::A_REAL)> { static constexpr bool value = P::A_REAL; };
// ... and the same for B: static constexpr bool B_REAL = true on a problem with B_CPLX (complex storage, zero imaginary parts)
template <class P, class = void> struct gemm_breal { static constexpr bool value = false; };
template <class P> struct gemm_breal<P, decltype((void)P::B_REAL)> { static constexpr bool value = P::B_REAL; };
// optional problem trait: static constexpr bool TILE_SKIP = true -- skip_tile(row0, col0) is true for work-group tiles whose
// output nobody wants (the strictly lower tiles of a symmetric result): such a work-group returns before its first barrier
template <class P, class = void> struct gemm_tileskip { static constexpr bool value = false; };
template <class P> struct gemm_tileskip<P, decltype((void)P::TILE_SKIP)> { static constexpr bool value = P::TILE_SKIP; };
// optional problem trait (with INCR): static constexpr bool INCR_SEG = true -- the operands are affine in the contraction
// index on [0, kseg()) and again on [kseg(), klimit): the refill re-bases its pointers from baseA2(b, row) / baseB2(b, col)
// (the sources of index kseg()) when it reaches kseg(), a multiple of 8
template <class P, class = void> struct gemm_incr_seg { static constexpr bool value = false; };
template <class P> struct gemm_incr_seg<P, decltype((void)P::INCR_SEG)> { static constexpr bool value = P::INCR_SEG; };
// optional problem trait: static constexpr bool BTILE_SKIP = true -- skip_tile_b(b, row0, col0) is true for work-group tiles of
// batch b whose output nobody wants (the beta columns of a closed-shell walker, whose propagated alpha block is copied over them
// afterwards: k_gemm.hip): such a work-group returns before its first barrier
template <class P, class = void> struct gemm_btileskip { static constexpr bool value = false; };
template <class P> struct gemm_btileskip<P, decltype((void)P::BTILE_SKIP)> { static constexpr bool value = P::BTILE_SKIP; };
template <class P, bool I> struct gemm_incr_types { using A = const void *; using B = const void *; };
template <class P> struct gemm_incr_types<P, true> {
    using A = decltype(((const P *)nullptr)->baseA(0, 0));
    using B = decltype(((const P *)nullptr)->baseB(0, 0));
};

// KC: k-chunks of 8 per ring slot / barrier (1 or 2).  With KC = 2 the fragments of the second half are
// read from LDS while the MFMAs of the first half run, and the barrier cost is paid once per 16 indices.
// WPE: waves per SIMD the register allocation must leave room for (amdgpu_waves_per_eu).  A work-group of 4 + 4 or 8 waves puts
// two waves on every SIMD; with WPE = 4 (at most 128 VGPRs) a second work-group is co-resident on the CU, whose MFMAs run
// while the first sits at its chunk barrier or in its epilogue.
template <int WM, int WN, int TM, int TN, int D, class P, int MAP, bool K3M = false, int KC = 1, int STAG = 0, int WPE = 1>
__global__ __launch_bounds__(WM *WN * 64 * ((STAG == 3 || STAG == 5) ? 2 : 1), WPE) void mfma_gemm_wg_kernel(P p, const void *zero16) {
#ifdef AFQ_TUNING
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
    static_assert(P::A_CPLX, "A operand must be complex");
    static_assert(D == 2 || D == 4 || D == 8, "ring depth must be 2, 4 or 8");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NW = WM * WN;
    constexpr int RT = WM * TM, CT = WN * TN;              // tile rows / cols of the work-group tile
    constexpr int NA = RT * 2;                             // A fragments per chunk (1 KB each)
    constexpr int NB = P::B_CPLX ? CT * 2 : CT;            // B fragments (1 KB each)
    constexpr int LPA = (NA + NW - 1) / NW, LPB = (NB + NW - 1) / NW, LPW = KC * (LPA + LPB);
    constexpr int SUB = (NA + NB) * 1024;                    // bytes of one 8-index sub-chunk
    constexpr int CHUNK = KC * SUB;
    constexpr int NWAIT = (D - 2) * LPW;
    static_assert(NWAIT <= 63, "vmcnt field is 6 bits");
    const int lane = threadIdx.x & 63;
    // STAG == 3: the work-group carries NW extra LOADER waves (one per compute wave, i.e. one per SIMD partner slot) that
    // do nothing but the ring refill; the compute waves run the pipelined loop without it.  An LDS-DMA instruction keeps
    // its wave's instruction issue busy for 100+ cycles and only two or three MFMAs queue up ahead of it, so a compute
    // wave that refills the ring itself idles the matrix pipe of its SIMD for most of that time when it is alone there.
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: LDS-DMA targets (M0) stay in SGPRs
    // (STAG == 5: the same loader waves; the compute waves run a lean loop with ONE set of operand fragments -- for kernels
    //  that are to fit 128 VGPRs so that two work-groups share a CU, see the end of the chunk loops)
    constexpr bool LOADERS = STAG == 3 || STAG == 5;
    const bool loader = LOADERS && wave_all >= WM * WN;
    const int wave = loader ? wave_all - WM * WN : wave_all;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_m = (p.rows + 16 * RT - 1) / (16 * RT);
    const int tiles_n = (p.cols + 16 * CT - 1) / (16 * CT);
    const long per_batch = (long)tiles_m * tiles_n;
    int b, tm, tn;
    if (MAP == MAP_BATCH_XCD || MAP == MAP_BATCH_XCD_ROWS) {
        const int nb8 = p.batch < 8 ? p.batch : 8;
        const int grp = blockIdx.x % nb8;
        const long t = blockIdx.x / nb8;
        b = grp + (int)(t / per_batch) * nb8;
        if (b >= p.batch) return;
        const int rem = (int)(t % per_batch);
        if (MAP == MAP_BATCH_XCD_ROWS) { tn = rem / tiles_m; tm = rem % tiles_m; }
        else { tm = rem / tiles_n; tn = rem % tiles_n; }
        // triangular B: the contraction grows with the column tile -- longest work-groups first, the short ones fill the tail
        if (gemm_kcut<P>::value) tn = tiles_n - 1 - tn;
    } else if (MAP == MAP_COLPANEL_XCD) {
        // single batch: work-groups are dealt round-robin to the 8 XCDs, so XCD x takes the column panels
        // x, x+8, ... with ALL their row tiles: a B panel (and its slice of the output) stays in one L2
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        b = 0;
        tn = (j / tiles_m) * 8 + x; tm = j % tiles_m;
        if (tn >= tiles_n) return;
    } else if (MAP == MAP_COLTILE_SLOW) {
        const long t = blockIdx.x, per_tn = (long)p.batch * tiles_m;
        tn = (int)(t / per_tn);
        const long rem = t % per_tn;
        b = (int)(rem / tiles_m); tm = (int)(rem % tiles_m);
    } else {
        const long t = blockIdx.x;
        b = (int)(t / per_batch);
        const int rem = (int)(t % per_batch);
        if (MAP == MAP_ROWS_FAST) { tn = rem / tiles_m; tm = rem % tiles_m; }
        else { tm = rem / tiles_n; tn = rem % tiles_n; }
    }
    if (!p.active(b)) {                                    // uniform over the work-group
        if constexpr (gemm_inactive_copy<P>::value) p.inactive_tile(b, tm * 16 * RT, 16 * RT, tn * 16 * CT, 16 * CT, (int)threadIdx.x, (int)blockDim.x);
        return;
    }
    const int row0 = tm * 16 * RT, col0 = tn * 16 * CT;
    if constexpr (gemm_tileskip<P>::value) { if (p.skip_tile(row0, col0)) return; }   // uniform over the work-group
    if constexpr (gemm_btileskip<P>::value) { if (p.skip_tile_b(b, row0, col0)) return; }
    const int lr = lane & 15, lk = lane >> 4;
    unsigned char *scratch = smem + (size_t)D * CHUNK + (size_t)wave * 1024;
    const unsigned ring_l = lds_addr(smem);
    int kdim_wg = p.kdim;
    if constexpr (gemm_kcut<P>::value) kdim_wg = p.kcut(b, col0, 16 * CT);
    const int nchunks = (kdim_wg + 8 * KC - 1) / (8 * KC);
    const int b_half = lane >> 5, b_lp = lane & 31;
    const int b_kk = b_lp >> 3, b_cc = (b_lp & 7) * 2;

    // incremental refill state (INCR problems): current source of every fragment this wave moves, its static validity
    // and the contraction index the next refill starts at
    using ptrA_t = typename gemm_incr_types<P, gemm_incr<P>::value>::A;
    using ptrB_t = typename gemm_incr_types<P, gemm_incr<P>::value>::B;
    ptrA_t curA[LPA];
    ptrB_t curB[LPB];
    bool okA[LPA], okB[LPB];
    int kcur = 0, klim = 0;
    if constexpr (gemm_incr<P>::value) {
        klim = p.klimit(b);
#pragma unroll
        for (int t = 0; t < LPA; ++t) {
            const int f = wave + t * NW, row = row0 + (f >> 1) * 16 + lr;
            okA[t] = f < NA && row < p.rows && p.rowok(b, row);
            curA[t] = p.baseA(b, okA[t] ? row : row0) + (long)(2 * lk + (f & 1)) * p.kstepA();
        }
#pragma unroll
        for (int t = 0; t < LPB; ++t) {
            const int f = wave + t * NW;
            const int col = P::B_CPLX ? col0 + (f >> 1) * 16 + lr : col0 + f * 16 + b_cc;
            const int kl = P::B_CPLX ? 2 * lk + (f & 1) : 2 * b_kk + b_half;
            okB[t] = f < NB && col < p.cols && p.colok(b, col);
            curB[t] = p.baseB(b, okB[t] ? col : col0) + (long)kl * p.kstepB(b);
        }
    }
    auto issue = [&](int c, int slot) {
        if constexpr (gemm_incr<P>::value) {
            // refills come strictly in chunk order (the call sites below), so `c` is implied by kcur
#pragma unroll
            for (int sub = 0; sub < KC; ++sub) {
                unsigned char *dst = smem + (size_t)slot * CHUNK + (size_t)sub * SUB;
                if constexpr (gemm_incr_seg<P>::value) {
                    if (kcur == p.kseg()) {                    // uniform: second affine segment of both operands
#pragma unroll
                        for (int t = 0; t < LPA; ++t) {
                            const int f = wave + t * NW, row = row0 + (f >> 1) * 16 + lr;
                            curA[t] = p.baseA2(b, okA[t] ? row : row0) + (long)(2 * lk + (f & 1)) * p.kstepA();
                        }
#pragma unroll
                        for (int t = 0; t < LPB; ++t) {
                            const int f = wave + t * NW;
                            const int col = P::B_CPLX ? col0 + (f >> 1) * 16 + lr : col0 + f * 16 + b_cc;
                            const int kl = P::B_CPLX ? 2 * lk + (f & 1) : 2 * b_kk + b_half;
                            curB[t] = p.baseB2(b, okB[t] ? col : col0) + (long)kl * p.kstepB(b);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < LPA; ++t) {
                    const int f = wave + t * NW;
                    const bool ok = okA[t] && kcur + 2 * lk + (f & 1) < klim;
                    glds16(ok ? (const void *)curA[t] : zero16, f < NA ? dst + f * 1024 : scratch);
                    curA[t] += 8 * p.kstepA();
                }
#pragma unroll
                for (int t = 0; t < LPB; ++t) {
                    const int f = wave + t * NW;
                    const int kl = P::B_CPLX ? 2 * lk + (f & 1) : 2 * b_kk + b_half;
                    const bool ok = okB[t] && kcur + kl < klim;
                    glds16(ok ? (const void *)curB[t] : zero16, f < NB ? dst + (NA + f) * 1024 : scratch);
                    curB[t] += 8 * p.kstepB(b);
                }
                kcur += 8;
            }
            return;
        }
#pragma unroll
        for (int sub = 0; sub < KC; ++sub) {
        unsigned char *dst = smem + (size_t)slot * CHUNK + (size_t)sub * SUB;
        const int k0 = (c * KC + sub) * 8;
#pragma unroll
        for (int t = 0; t < LPA; ++t) {
            const int f = wave + t * NW;                   // A fragment index: tile row f>>1, sub-step f&1
            const int k = k0 + 2 * lk + (f & 1), row = row0 + (f >> 1) * 16 + lr;
            const bool ok = f < NA && k < p.kdim && row < p.rows;
            const void *src = ok ? (const void *)p.ptrA(b, row, k) : zero16;
            glds16(src, f < NA ? dst + f * 1024 : scratch);
        }
#pragma unroll
        for (int t = 0; t < LPB; ++t) {
            const int f = wave + t * NW;
            const void *src = zero16;
            if (P::B_CPLX) {
                const int k = k0 + 2 * lk + (f & 1), col = col0 + (f >> 1) * 16 + lr;
                if (f < NB && k < p.kdim && col < p.cols) src = (const void *)p.ptrB(b, k, col);
            } else {
                const int k = k0 + 2 * b_kk + b_half, col = col0 + f * 16 + b_cc;
                if (f < NB && k < p.kdim && col < p.cols) src = (const void *)p.ptrB(b, k, col);
            }
            glds16(src, f < NB ? dst + (NA + f) * 1024 : scratch);
        }
        }
    };

    d4_t accR[TM][TN], accI[TM][TN], acc3[K3M ? TM : 1][K3M ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            accR[i][j] = (d4_t){0, 0, 0, 0};
            accI[i][j] = (d4_t){0, 0, 0, 0};
            if (K3M) acc3[i][j] = (d4_t){0, 0, 0, 0};
        }

    // operand fragments of ONE chunk (declared outside the chunk loop: the staggered waves carry them across a barrier)
    d2_t a[KC][TM][2];
    d2_t bc[KC][TN][2];
    double br[KC][TN][2];
    auto read_sub = [&](unsigned sl0, int sub) __attribute__((always_inline)) {
        const unsigned sl = sl0 + sub * SUB;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // (a real operand is read as its real half only: a 16-byte read whose upper half is never used lets the
                //  register allocator hand that half to another value while the asynchronous read is still in flight)
                if (gemm_areal<P>::value) a[sub][i][s][0] = lds_read_b64(sl + ((wm * TM + i) * 2 + s) * 1024 + lane * 16);
                else a[sub][i][s] = lds_read_b128(sl + ((wm * TM + i) * 2 + s) * 1024 + lane * 16);
            }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (P::B_CPLX && gemm_breal<P>::value) bc[sub][j][s][0] = lds_read_b64(sl + (NA + (wn * TN + j) * 2 + s) * 1024 + lane * 16);
                else if (P::B_CPLX) bc[sub][j][s] = lds_read_b128(sl + (NA + (wn * TN + j) * 2 + s) * 1024 + lane * 16);
                else br[sub][j][s] = lds_read_b64(sl + (NA + wn * TN + j) * 1024 + s * 512 + lane * 8);
            }
    };
    auto conj_sub = [&](int sub) __attribute__((always_inline)) {
        if (gemm_conj_a<P>::value) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) a[sub][i][s][1] = -a[sub][i][s][1];
        }
    };
    // the MFMAs of sub-step s (4 of the 8 contraction indices of a chunk)
    auto mfma_step = [&](int sub, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (P::B_CPLX && gemm_areal<P>::value) {
                    accR[i][j] = mfma16(a[sub][i][s][0], bc[sub][j][s][0], accR[i][j]);
                    accI[i][j] = mfma16(a[sub][i][s][0], bc[sub][j][s][1], accI[i][j]);
                } else if (P::B_CPLX && gemm_breal<P>::value) {
                    accR[i][j] = mfma16(a[sub][i][s][0], bc[sub][j][s][0], accR[i][j]);
                    accI[i][j] = mfma16(a[sub][i][s][1], bc[sub][j][s][0], accI[i][j]);
                } else if (P::B_CPLX && K3M) {
                    accR[i][j] = mfma16(a[sub][i][s][0], bc[sub][j][s][0], accR[i][j]);                          // P1
                    accI[i][j] = mfma16(a[sub][i][s][1], bc[sub][j][s][1], accI[i][j]);                          // P2
                    acc3[i][j] = mfma16(a[sub][i][s][0] + a[sub][i][s][1], bc[sub][j][s][0] + bc[sub][j][s][1], acc3[i][j]);  // P3
                } else if (P::B_CPLX) {
                    accR[i][j] = mfma16(a[sub][i][s][0], bc[sub][j][s][0], accR[i][j]);
                    accI[i][j] = mfma16(a[sub][i][s][0], bc[sub][j][s][1], accI[i][j]);
                    accR[i][j] = mfma16(-a[sub][i][s][1], bc[sub][j][s][1], accR[i][j]);
                    accI[i][j] = mfma16(a[sub][i][s][1], bc[sub][j][s][0], accI[i][j]);
                } else {
                    accR[i][j] = mfma16(a[sub][i][s][0], br[sub][j][s], accR[i][j]);
                    accI[i][j] = mfma16(a[sub][i][s][1], br[sub][j][s], accI[i][j]);
                }
            }
    };
    auto mfma_sub = [&](int sub) __attribute__((always_inline)) {
        conj_sub(sub);
        mfma_step(sub, 0);
        mfma_step(sub, 1);
    };

#ifdef AFQ_TUNING
    const int abl = afq_gemm_abl;
    const unsigned long long ts1 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
#endif
    // STAG == 4: the ring is refilled THROUGH REGISTERS (global_load_dwordx4, ds_write_b128 one chunk later) instead of by
    // LDS-DMA.  A global_load ... lds instruction keeps its wave's instruction issue busy for 150+ cycles and only a few
    // MFMAs queue up ahead of it, so with one wave per SIMD the four refill instructions of a chunk idle the matrix pipe
    // for ~600 of its 1900 cycles (VhsProb at C3: SQ_WAIT_INST_LDS 1.5 % and zero bank conflicts -- it is not the LDS).
    // A plain load and a plain LDS store issue in a few cycles each; the price is 16 bytes of staging registers per
    // fragment and chunk in flight -- and only ONE chunk of latency tolerance (the loads of chunk c + D are waited for one
    // iteration later, in program order ahead of half the MFMAs).  MEASURED NEGATIVE (round 3, VhsProb at C3, tuning knob
    // AFQ_VHS_RREG): correct, 67.9 us against 61.5 us with the DMA ring; 63 % of that kernel's L2 accesses miss
    // (TCC_MISS / (HIT + MISS), profiles/archive/r03_pmc_sq_tcp_bench_kernels.txt), so the three chunks the DMA ring keeps in
    // flight matter more than the issue slots it costs.  Kept for tuning builds only.
    if constexpr (STAG == 4) {
        static_assert(KC == 1 && gemm_incr<P>::value, "register-staged refill: incremental problems, one sub-chunk per slot");
        d2_t stgA[LPA], stgB[LPB];
        auto gload = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < LPA; ++t) {
                const int f = wave + t * NW;
                const bool ok = okA[t] && kcur + 2 * lk + (f & 1) < klim;
                stgA[t] = *(const d2_t *)(ok ? (const void *)curA[t] : zero16);
                curA[t] += 8 * p.kstepA();
            }
#pragma unroll
            for (int t = 0; t < LPB; ++t) {
                const int f = wave + t * NW;
                const int kl = P::B_CPLX ? 2 * lk + (f & 1) : 2 * b_kk + b_half;
                const bool ok = okB[t] && kcur + kl < klim;
                stgB[t] = *(const d2_t *)(ok ? (const void *)curB[t] : zero16);
                curB[t] += 8 * p.kstepB(b);
            }
            kcur += 8;
        };
        auto lstore = [&](int slot) __attribute__((always_inline)) {
            unsigned char *dst = smem + (size_t)slot * CHUNK + lane * 16;
#pragma unroll
            for (int t = 0; t < LPA; ++t) { const int f = wave + t * NW; if (f < NA) *(d2_t *)(dst + f * 1024) = stgA[t]; }
#pragma unroll
            for (int t = 0; t < LPB; ++t) { const int f = wave + t * NW; if (f < NB) *(d2_t *)(dst + (NA + f) * 1024) = stgB[t]; }
        };
#pragma unroll
        for (int c = 0; c < D - 1; ++c) { gload(); lstore(c); }
        gload();                                              // chunk D - 1 waits in registers
        for (int c = 0; c < nchunks; ++c) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's stores of chunk c + D - 2 have landed
            __builtin_amdgcn_s_barrier();                          // chunk c complete; everyone is done reading chunk c - 1
            read_sub(ring_l + (c & (D - 1)) * CHUNK, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            conj_sub(0);
            mfma_step(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            lstore((c + D - 1) & (D - 1));                         // chunk c + D - 1, loaded one iteration ago, into the slot of c - 1
            gload();                                               // chunk c + D
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(0, 1);
        }
    } else
    if (!LOADERS || loader) {
#pragma unroll
        for (int c = 0; c < D - 1; ++c) issue(c, c);
    }
    if constexpr (STAG == 4) {
    } else
#ifdef AFQ_TUNING
    if (STAG == 3 && (abl & 32) && !loader) __builtin_amdgcn_s_setprio(3);       // experiment: compute waves first
    if (STAG == 3 && (abl & 64) && loader) __builtin_amdgcn_s_setprio(3);        // experiment: loader waves first
#endif
    if (LOADERS && loader) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(D - 1, (D - 1) & (D - 1));
        for (int c = 0; c + 1 < nchunks; ++c) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");
            __builtin_amdgcn_s_barrier();
            issue(c + D, (c + D) & (D - 1));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // STAG: the second half of the waves (the SIMD partners of the first half when the work-group has 8 waves) cross
    // the chunk barrier in the MIDDLE of a chunk's MFMAs -- sub-step 1 of chunk c is multiplied right behind barrier
    // c + 1 from fragments already in registers, then the fragments of chunk c + 1 are read and its sub-step 0
    // multiplied -- so that the partner's LDS reads and ring refill run under this wave's MFMAs and vice versa.
    // STAG == 2: half-chunk pipeline with every non-MFMA instruction BETWEEN two MFMA groups of the wave (program order
    // pinned with sched_barrier), in the 64-cycle shadows of the MFMAs: the sub-step 0 MFMAs of chunk c run while the
    // sub-step 1 fragments of chunk c are read; the wave crosses barrier c + 1; the sub-step 1 MFMAs run while the ring is
    // refilled (address arithmetic included) and the sub-step 0 fragments of chunk c + 1 are read.  The plain loop below
    // does refill, reads and the wait for them in a block behind the barrier with the matrix pipe idle, which only
    // works out when several waves share a SIMD; the small-output contractions of this library run one wave per SIMD.
    if constexpr (STAG == 4) {
    } else
    if constexpr (STAG == 5) {
        // lean compute loop behind loader waves: per sub-step read ITS fragments (one set of registers for both sub-steps),
        // wait, multiply.  Nothing of the wave's own overlaps the LDS latency -- that is what the second work-group on the CU
        // is for (launch with WPE = 4).
        static_assert(KC == 1, "lean loop: one sub-chunk per slot");
        for (int c = 0; c < nchunks; ++c) {
            __builtin_amdgcn_s_barrier();
            const unsigned sl = ring_l + (c & (D - 1)) * CHUNK;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (gemm_areal<P>::value) a[0][i][0][0] = lds_read_b64(sl + ((wm * TM + i) * 2 + s2) * 1024 + lane * 16);
                    else a[0][i][0] = lds_read_b128(sl + ((wm * TM + i) * 2 + s2) * 1024 + lane * 16);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (P::B_CPLX && gemm_breal<P>::value) bc[0][j][0][0] = lds_read_b64(sl + (NA + (wn * TN + j) * 2 + s2) * 1024 + lane * 16);
                    else if (P::B_CPLX) bc[0][j][0] = lds_read_b128(sl + (NA + (wn * TN + j) * 2 + s2) * 1024 + lane * 16);
                    else br[0][j][0] = lds_read_b64(sl + (NA + wn * TN + j) * 1024 + s2 * 512 + lane * 8);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (gemm_conj_a<P>::value) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[0][i][0][1] = -a[0][i][0][1];
                }
                mfma_step(0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
    if ((STAG == 2 || STAG == 3) && KC == 1) {
        constexpr bool own_refill = STAG == 2;
        constexpr int NG = TM * TN, NR = TM + TN;
        constexpr int RPG = NG > 1 ? (NR + NG - 2) / (NG - 1) : NR;
        auto mfma_tile = [&](int i, int j, int s) __attribute__((always_inline)) {
            if (P::B_CPLX && gemm_areal<P>::value) {
                accR[i][j] = mfma16(a[0][i][s][0], bc[0][j][s][0], accR[i][j]);
                accI[i][j] = mfma16(a[0][i][s][0], bc[0][j][s][1], accI[i][j]);
            } else if (P::B_CPLX && gemm_breal<P>::value) {
                accR[i][j] = mfma16(a[0][i][s][0], bc[0][j][s][0], accR[i][j]);
                accI[i][j] = mfma16(a[0][i][s][1], bc[0][j][s][0], accI[i][j]);
            } else if (P::B_CPLX && K3M) {
                accR[i][j] = mfma16(a[0][i][s][0], bc[0][j][s][0], accR[i][j]);
                accI[i][j] = mfma16(a[0][i][s][1], bc[0][j][s][1], accI[i][j]);
                acc3[i][j] = mfma16(a[0][i][s][0] + a[0][i][s][1], bc[0][j][s][0] + bc[0][j][s][1], acc3[i][j]);
            } else if (P::B_CPLX) {
                accR[i][j] = mfma16(a[0][i][s][0], bc[0][j][s][0], accR[i][j]);
                accI[i][j] = mfma16(a[0][i][s][0], bc[0][j][s][1], accI[i][j]);
                accR[i][j] = mfma16(-a[0][i][s][1], bc[0][j][s][1], accR[i][j]);
                accI[i][j] = mfma16(a[0][i][s][1], bc[0][j][s][0], accI[i][j]);
            } else {
                accR[i][j] = mfma16(a[0][i][s][0], br[0][j][s], accR[i][j]);
                accI[i][j] = mfma16(a[0][i][s][1], br[0][j][s], accI[i][j]);
            }
        };
        auto read_one = [&](unsigned sl, int r, int s) __attribute__((always_inline)) {      // fragment r of sub-step s
            if (r < TM) {
                if (gemm_areal<P>::value) a[0][r][s][0] = lds_read_b64(sl + ((wm * TM + r) * 2 + s) * 1024 + lane * 16);
                else a[0][r][s] = lds_read_b128(sl + ((wm * TM + r) * 2 + s) * 1024 + lane * 16);
            }
            else if (P::B_CPLX && gemm_breal<P>::value) bc[0][r - TM][s][0] = lds_read_b64(sl + (NA + (wn * TN + r - TM) * 2 + s) * 1024 + lane * 16);
            else if (P::B_CPLX) bc[0][r - TM][s] = lds_read_b128(sl + (NA + (wn * TN + r - TM) * 2 + s) * 1024 + lane * 16);
            else br[0][r - TM][s] = lds_read_b64(sl + (NA + wn * TN + r - TM) * 1024 + s * 512 + lane * 8);
        };
        auto conj_set = [&](int s) __attribute__((always_inline)) {
            if (gemm_conj_a<P>::value) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[0][i][s][1] = -a[0][i][s][1];
            }
        };
        // MFMA groups of sub-step s; the reads of sub-step rs from slot `sl` (and the ring refill) in between
        auto half = [&](const int s, const unsigned sl, const int rs, const bool fetch, const int refill_c)
            __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                mfma_tile(g / TN, g % TN, s);
                __builtin_amdgcn_sched_barrier(0);
                // (dealing the refill's DMA instructions over the group boundaries instead was measured 2 x SLOWER:
                //  an LDS-DMA instruction issued between MFMAs stalls the wave for 300-900 cycles)
                if (g == 0 && refill_c >= 0) issue(refill_c, refill_c & (D - 1));
                if (fetch && (g < NG - 1 || NG == 1)) {
#pragma unroll
                    for (int q = 0; q < RPG; ++q)
                        if (g * RPG + q < NR) read_one(sl, g * RPG + q, rs);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (fetch) conj_set(rs);
        };
        if (own_refill) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");
        __builtin_amdgcn_s_barrier();
        if (own_refill) issue(D - 1, (D - 1) & (D - 1));
#pragma unroll
        for (int r = 0; r < NR; ++r) read_one(ring_l, r, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        conj_set(0);
        for (int c = 0; c < nchunks; ++c) {
            const bool more = c + 1 < nchunks;
            half(0, ring_l + (c & (D - 1)) * CHUNK, 1, true, -1);
            if (more) {
                if (own_refill) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            half(1, ring_l + ((c + 1) & (D - 1)) * CHUNK, 0, more, (own_refill && more) ? c + D : -1);
        }
    } else
    if (STAG == 1 && KC == 1 && wave >= NW / 2) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(D - 1, (D - 1) & (D - 1));
        read_sub(ring_l, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        conj_sub(0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        for (int c = 0; c < nchunks; ++c) {
            if (c + 1 < nchunks) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");
                __builtin_amdgcn_s_barrier();
                issue(c + D, (c + D) & (D - 1));
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < nchunks) {
                read_sub(ring_l + ((c + 1) & (D - 1)) * CHUNK, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                conj_sub(0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
    for (int c = 0; c < nchunks; ++c) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");
        GEMM_UNLESS(8)
        __builtin_amdgcn_s_barrier();
        GEMM_UNLESS(2)
        issue(c + D - 1, (c + D - 1) & (D - 1));
        const unsigned sl0 = ring_l + (c & (D - 1)) * CHUNK;
        GEMM_UNLESS(4)
        read_sub(sl0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (KC == 2) {
            read_sub(sl0, KC - 1);                         // in flight under the MFMAs of the first half
            __builtin_amdgcn_sched_barrier(0);
        }
        GEMM_UNLESS(1)
        mfma_sub(0);
        if (KC == 2) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_sub(KC - 1);
        }
    }
#ifdef AFQ_TUNING
    const unsigned long long ts2 = __builtin_amdgcn_s_memtime(), tr2 = __builtin_amdgcn_s_memrealtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (P::B_CPLX && K3M && !gemm_areal<P>::value && !gemm_breal<P>::value) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const d4_t p1 = accR[i][j], p2 = accI[i][j];
                accR[i][j] = p1 - p2;
                accI[i][j] = acc3[i][j] - p1 - p2;
            }
    }
    const int wrow0 = row0 + wm * TM * 16, wcol0 = col0 + wn * TN * 16;
    if constexpr (gemm_rowdot<P>::value) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wrow0 + i * 16 + lk + 4 * r;
                double sr = 0.0, si = 0.0;                   // this lane's column of each of the wave's TN tiles
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = wcol0 + j * 16 + lr;
                    if (row < p.rows && col < p.cols) {
                        const cplx g = p.dot_operand(b, row, col);
                        sr += accR[i][j][r] * g.x - accI[i][j][r] * g.y;
                        si += accR[i][j][r] * g.y + accI[i][j][r] * g.x;
                    }
                }
                // sum over the 16 lanes of one accumulator row (lr = lane & 15) with DPP moves; every lane takes part
                sr = gemm_row16_sum(sr); si = gemm_row16_sum(si);
                // one partial sum per wave and row, filed under the wave's first tile; its other tiles get a zero
                if (lr < TN && row < p.rows && wcol0 + lr * 16 < p.cols)
                    p.store_dot(b, row, (wcol0 >> 4) + lr, lr == 0 ? sr : 0.0, lr == 0 ? si : 0.0);
            }
    } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wrow0 + i * 16 + lk + 4 * r;
                const int col = wcol0 + j * 16 + lr;
                GEMM_UNLESS(16)
                if (row < p.rows && col < p.cols) p.store(b, row, col, accR[i][j][r], accI[i][j][r]);
            }
    if constexpr (gemm_coldot<P>::value) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wcol0 + j * 16 + lr;
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wrow0 + i * 16 + lk + 4 * r;
                    if (row < p.rows && col < p.cols) {
                        const cplx c = p.coldot_coef(b, row, col);
                        sr += accR[i][j][r] * c.x - accI[i][j][r] * c.y;
                        si += accR[i][j][r] * c.y + accI[i][j][r] * c.x;
                    }
                }
            // the four lanes lr, lr + 16, lr + 32, lr + 48 hold the rows lk + 4 r of this column
            sr += __shfl_xor(sr, 16); si += __shfl_xor(si, 16);
            sr += __shfl_xor(sr, 32); si += __shfl_xor(si, 32);
            if (lk == 0 && col < p.cols && wrow0 < p.rows) p.store_coldot(b, wrow0 / (16 * TM), col, sr, si);
        }
    }
    }
#ifdef AFQ_TUNING
    if (afq_gemm_ts && threadIdx.x == 0 && blockIdx.x < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
        unsigned long long *o = afq_gemm_ts + blockIdx.x * 4;
        o[0] = tr2 - tr1; o[1] = ts2 - ts1; o[2] = ts3 - ts2; o[3] = (unsigned long long)nchunks;
    }
#endif
}

// Flops the matrix pipe executes for one launch of the engine (what "issued" rooflines are priced on): every work-group
// tile multiplies its full (16 RT) x (16 CT) block over whole chunks of 8 contraction indices -- padding included --
// with 2 real products per element pair when B is real, 3 with the 3-multiplication complex product, else 4.
// `klen(b, col0, ncols)`: contraction length of a tile (kdim, or the KCUT trait restated on the host).
template <int WM, int WN, int TM, int TN, class P, bool K3M = false, int KC = 1, class KLen>
inline double mfma_gemm_wg_issued_flops(const P &p, KLen klen) {
    constexpr int RT = WM * TM, CT = WN * TN;
    const long tiles_m = (p.rows + 16 * RT - 1) / (16 * RT);
    const long tiles_n = (p.cols + 16 * CT - 1) / (16 * CT);
    const double mults = P::B_CPLX ? ((gemm_areal<P>::value || gemm_breal<P>::value) ? 2.0 : K3M ? 3.0 : 4.0) : 2.0;
    double f = 0.0;
    for (int b = 0; b < p.batch; ++b)
        for (long tn = 0; tn < tiles_n; ++tn) {
            const long k = klen(b, (int)(tn * 16 * CT), 16 * CT);
            const long kpad = (k + 8 * KC - 1) / (8 * KC) * (8 * KC);
            f += 2.0 * mults * (double)tiles_m * (16.0 * RT) * (16.0 * CT) * (double)kpad;
        }
    return f;
}

template <int WM, int WN, int TM, int TN, int D, class P, int MAP, bool K3M = false, int KC = 1, int STAG = 0, int WPE = 1>
inline hipError_t launch_mfma_gemm_wg(const P &p, hipStream_t stream, const void *zero16) {
    constexpr int RT = WM * TM, CT = WN * TN;
    constexpr int NA = RT * 2, NB = P::B_CPLX ? CT * 2 : CT;
    const long tiles_m = (p.rows + 16 * RT - 1) / (16 * RT);
    const long tiles_n = (p.cols + 16 * CT - 1) / (16 * CT);
    const long per_batch = tiles_m * tiles_n;
    long nblk = (long)p.batch * per_batch;
    if (nblk == 0) return hipSuccess;
    if (MAP == MAP_BATCH_XCD || MAP == MAP_BATCH_XCD_ROWS) {
        const int nb8 = p.batch < 8 ? p.batch : 8;
        nblk = per_batch * ((p.batch + nb8 - 1) / nb8) * nb8;
    }
    if (MAP == MAP_COLPANEL_XCD) nblk = 8 * tiles_m * ((tiles_n + 7) / 8);
    const size_t lds = (size_t)D * KC * (NA + NB) * 1024 + (size_t)WM * WN * 1024;
    static_assert((STAG != 3 && STAG != 5) || WM * WN <= 8, "compute + loader waves must fit one work-group");
    auto kern = mfma_gemm_wg_kernel<WM, WN, TM, TN, D, P, MAP, K3M, KC, STAG, WPE>;
    static size_t lds_set[AFQ_MAX_DEVICES] = {0};   // one per template instantiation and device: set the cap once
    {
        hipError_t e = afq_raise_lds((const void *)kern, lds, lds_set);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * WM * WN * ((STAG == 3 || STAG == 5) ? 2 : 1)), lds, stream, p, zero16);
    return hipGetLastError();
}
