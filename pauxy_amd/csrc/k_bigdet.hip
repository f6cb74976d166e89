// Green's function / overlap determinant for large electron counts (32 < N <= 128 per
// spin; BASELINE configs[3], 16x16 Hubbard: N = 128, M = 256), walkers/single_det.py:295-321.
//
//   1. O_s   = phi_s^T conj(psi_s)          batched fp64-MFMA GEMM (work-group LDS-ring engine, 3M)
//   2. O_s^-1 and det O_s                   register-resident Gauss-Jordan, one 512-thread work-group per
//                                           matrix: the 128 x 128 complex matrix (256 KB) lives in the
//                                           VGPRs of the 8 waves (32 elements per thread); per pivot step
//                                           only the pivot column and the scaled pivot row travel through
//                                           LDS (4 KB), two barriers per step, implicit row pivoting (no
//                                           row swaps: the permutation is undone when the inverse is stored)
//   3. Ghalf_s = O_s^-1 phi_s^T             batched fp64-MFMA GEMM
// This replaces the generic work-group LU + per-column substitution kernel, which is latency bound
// on global memory (14.7 ms per call at C4); the pieces above take ~0.1 + 0.3 + 0.15 ms.
#include "mfma_gemm_wg.h"
#include "gj_wave.h"
#include "weight_update.h"
#include <cstring>

#define GJ_N 128

// ------------------------------------------------------------------ 1. overlap matrices
// BR: the trial has no imaginary part (checked at upload): two real multiplications per element pair instead of three
template <bool BR>
struct OvlpProbT {
    static constexpr bool A_CPLX = true, B_CPLX = true, B_REAL = BR;
    int batch, rows, cols, kdim;     // 2 nw, nmax, nmax, M
    int nt, na, nb, ld;
    const cplx *phi;                 // [nw, M, nt]
    const cplx *psic;                // conj(psi) [M, nt]
    long psi_stride;                 // 0 or M*nt (per-walker trial)
    cplx *O;                         // [2 nw, ld * ld]
    const cplx *zero;
    __device__ bool active(int) const { return true; }
    __device__ cplx loadA(int, int, int) const { return cmake(0, 0); }
    __device__ cplx loadB(int, int, int) const { return cmake(0, 0); }
    __device__ const cplx *ptrA(int b, int row, int k) const {
        const int s = b & 1, ns = s ? nb : na;
        return row < ns ? phi + ((long)(b >> 1) * kdim + k) * nt + (s ? na : 0) + row : zero;
    }
    __device__ const cplx *ptrB(int b, int k, int col) const {
        const int s = b & 1, ns = s ? nb : na;
        return col < ns ? psic + (b >> 1) * psi_stride + (long)k * nt + (s ? na : 0) + col : zero;
    }
    static constexpr bool INCR = true;       // incremental refill of the ring engine
    __device__ int klimit(int) const { return kdim; }
    __device__ const cplx *baseA(int b, int row) const { return phi + (long)(b >> 1) * kdim * nt + ((b & 1) ? na : 0) + row; }
    __device__ const cplx *baseB(int b, int col) const { return psic + (b >> 1) * psi_stride + ((b & 1) ? na : 0) + col; }
    __device__ long kstepA() const { return nt; }
    __device__ long kstepB(int) const { return nt; }
    __device__ bool rowok(int b, int row) const { return row < ((b & 1) ? nb : na); }
    __device__ bool colok(int b, int col) const { return col < ((b & 1) ? nb : na); }
    __device__ void store(int b, int row, int col, double re, double im) const {
        O[(long)b * ld * ld + (long)row * ld + col] = cmake(re, im);
    }
};

typedef OvlpProbT<false> OvlpProb;

// ------------------------------------------------------------------ 3. Ghalf = Oinv phi^T
// CD: also leave diag(G_s)[n] = sum_i conj(psi[n, i]) Ghalf_s[i, n] behind, as one partial sum per block of 32 rows
// (gdiag[2 w + s][part][n]): the Hubbard force bias needs nothing else of the Green's function, and reading Ghalf back
// for it costs 268 MB per step at C4
template <bool CD>
struct GhalfProbT {
    static constexpr bool A_CPLX = true, B_CPLX = true, COLDOT = CD;
    int batch, rows, cols, kdim;     // 2 nw, nmax, M, nmax
    int nt, na, nb, ld, M;
    const cplx *psicT;               // conj(psi)^T [nt, M]
    cplx *gdiag;                     // [2 nw, nparts, M]
    int nparts;
    __device__ cplx coldot_coef(int b, int row, int col) const {
        const int s = b & 1, ns = s ? nb : na;
        return row < ns ? psicT[(long)((s ? na : 0) + row) * M + col] : cmake(0.0, 0.0);
    }
    __device__ void store_coldot(int b, int part, int col, double re, double im) const {
        gdiag[((long)b * nparts + part) * M + col] = cmake(re, im);
    }
    const cplx *Oinv;                // [2 nw, ld * ld]
    const cplx *phi;
    cplx *ghalf;                     // [nw, nt, M]
    int skip_store;                  // only the diagonal sums are wanted (afq_propagate_finish on an announced step)
    const cplx *zero;
    __device__ bool active(int) const { return true; }
    __device__ cplx loadA(int, int, int) const { return cmake(0, 0); }
    __device__ cplx loadB(int, int, int) const { return cmake(0, 0); }
    __device__ const cplx *ptrA(int b, int row, int k) const {
        const int ns = (b & 1) ? nb : na;
        return (row < ns && k < ns) ? Oinv + (long)b * ld * ld + (long)row * ld + k : zero;
    }
    __device__ const cplx *ptrB(int b, int k, int col) const {
        const int s = b & 1, ns = s ? nb : na;
        return k < ns ? phi + ((long)(b >> 1) * M + col) * nt + (s ? na : 0) + k : zero;
    }
    static constexpr bool INCR = true;       // incremental refill of the ring engine
    __device__ int klimit(int b) const { return (b & 1) ? nb : na; }
    __device__ const cplx *baseA(int b, int row) const { return Oinv + (long)b * ld * ld + (long)row * ld; }
    __device__ const cplx *baseB(int b, int col) const { return phi + ((long)(b >> 1) * M + col) * nt + ((b & 1) ? na : 0); }
    __device__ long kstepA() const { return 1; }
    __device__ long kstepB(int) const { return 1; }
    __device__ bool rowok(int b, int row) const { return row < ((b & 1) ? nb : na); }
    __device__ bool colok(int, int) const { return true; }
    __device__ void store(int b, int row, int col, double re, double im) const {
        const int s = b & 1, ns = s ? nb : na;
        if (row < ns && !skip_store) ghalf[((long)(b >> 1) * nt + (s ? na : 0) + row) * M + col] = cmake(re, im);
    }
};

typedef GhalfProbT<false> GhalfProb;

// Hubbard, continuous fields, REAL trial, a step whose Green's function only feeds the next force bias
// (propagation/hubbard.py:404-407 reads diag G alone): diag G_s[q] = sum_j W[q,j] phi[q,j] with W = conj(psi_s) O_s^-1 --
// a real-by-complex M x N x N product (two multiplications per element pair where O^-1 phi^T above takes three), its
// output never stored: every element meets phi[q,j] in the epilogue and the sums over 16 columns go to gdiag
// (rows = sites, cols = electrons, contraction = electrons; gdiag [2 nw, ceil(N / 16), M])
struct GdiagProbT {
    static constexpr bool A_CPLX = true, B_CPLX = true, A_REAL = true, ROWDOT = true;
    int batch, rows, cols, kdim;     // 2 nw, M, nmax, nmax
    int nt, na, nb, ld, M;
    const cplx *psic;                // conj(psi) [M, nt], imaginary parts exactly zero
    const cplx *Oinv;                // [2 nw, ld * ld]
    const cplx *phi;                 // [nw, M, nt]
    cplx *gdiag;                     // [2 nw, nparts, M]
    int nparts;
    const cplx *zero;
    __device__ bool active(int) const { return true; }
    __device__ cplx loadA(int, int, int) const { return cmake(0, 0); }
    __device__ cplx loadB(int, int, int) const { return cmake(0, 0); }
    __device__ const cplx *ptrA(int b, int row, int k) const {
        const int s = b & 1, ns = s ? nb : na;
        return k < ns ? psic + (long)row * nt + (s ? na : 0) + k : zero;
    }
    __device__ const cplx *ptrB(int b, int k, int col) const {
        const int ns = (b & 1) ? nb : na;
        return (k < ns && col < ns) ? Oinv + (long)b * ld * ld + (long)k * ld + col : zero;
    }
    static constexpr bool INCR = true;       // incremental refill of the ring engine
    __device__ int klimit(int b) const { return (b & 1) ? nb : na; }
    __device__ const cplx *baseA(int b, int row) const { return psic + (long)row * nt + ((b & 1) ? na : 0); }
    __device__ const cplx *baseB(int b, int col) const { return Oinv + (long)b * ld * ld + col; }
    __device__ long kstepA() const { return 1; }
    __device__ long kstepB(int) const { return ld; }
    __device__ bool rowok(int, int) const { return true; }
    __device__ bool colok(int b, int col) const { return col < ((b & 1) ? nb : na); }
    __device__ cplx dot_operand(int b, int row, int col) const {
        const int s = b & 1, ns = s ? nb : na;
        return col < ns ? phi[((long)(b >> 1) * M + row) * nt + (s ? na : 0) + col] : cmake(0.0, 0.0);
    }
    __device__ void store_dot(int b, int row, int tile, double re, double im) const {
        gdiag[((long)b * nparts + tile) * M + row] = cmake(re, im);
    }
    __device__ void store(int, int, int, double, double) const {}
};

// ------------------------------------------------------------------ 2. register-resident Gauss-Jordan
struct GjArgs {
    int na, nb, ld, write_inverse;
    cplx *O;                         // [2 nw, ld * ld], inverse written in place
    cplx *detm;                      // [2 nw] mantissa of det O_s
    int *dete;                       // [2 nw] binary exponent
    const int *only = nullptr;       // when set: only the matrices with a non-zero entry are processed
    unsigned long long *nflagged = nullptr;   // ... and counted here (afq_counters [2])
    unsigned long long *ts = nullptr;   // tuning builds (AFQ_GJ_TS): s_memtime stamps of work-group 0, [wave][block step][point]
    int dbg = 0;                     // tuning builds: timing ablations of gj_mfma_kernel (WRONG results): 1 no inversion of the
                                     // pivot tile, 2 no rank-16 update of the register tiles, 4 none of the diagonal tiles,
                                     // 8 no block steps at all (loads only); 64 flags every matrix (right results: everything
                                     // through the step-by-step kernel)
};

// wave-wide maximum of a 32-bit key by DPP row shifts / row broadcasts (no LDS traffic)
__device__ inline unsigned wave_max_u32(unsigned v) {
#define AFQ_DPP_MAX(ctrl, rmask)                                                                         \
    { const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false);      \
      v = v > t ? v : t; }
    AFQ_DPP_MAX(0x111, 0xf)      // row_shr:1
    AFQ_DPP_MAX(0x112, 0xf)      // row_shr:2
    AFQ_DPP_MAX(0x114, 0xf)      // row_shr:4
    AFQ_DPP_MAX(0x118, 0xf)      // row_shr:8   -> lane 15 of every row holds the row maximum
    AFQ_DPP_MAX(0x142, 0xa)      // row_bcast:15 into rows 1 and 3
    AFQ_DPP_MAX(0x143, 0xc)      // row_bcast:31 into rows 2 and 3
#undef AFQ_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// One 512-thread work-group per matrix.  Thread (tr = tid >> 5, tc = tid & 31) owns rows tr + 16 x
// (x < 8) and columns tc + 32 y (y < 4) in registers.  Pivot step k:
//   top      owners of column k (tc == k & 31) publish it                              -> barrier A
//   search   every wave: 32-bit key (fp32 |re|+|im|, row) of two column entries, DPP maximum -> p
//   owner    the half-wave that owns row p publishes row_p / d (k-th entry 1/d)         -> barrier B
//   update   v[i][j] <- (j == k ? 0 : v[i][j]) - col[i] * row[j]; row p itself <- row
// Rows are never swapped: step k marks row p as used, and the inverse is un-permuted when it is
// stored (A^-1[r][c] = W[prow[r]][invp[c]]).  The pivot is the largest fp32-rounded |re|+|im| among
// unused rows (ties -> lowest row), which differs from LAPACK's choice only in exact-tie/rounding
// cases; determinant and inverse do not depend on the pivot order beyond rounding.
__global__ __launch_bounds__(512) void gj_big_kernel(GjArgs a) {
    if (a.only) {                                        // (second pass behind the blocked kernel: flagged matrices only)
        if (!a.only[blockIdx.x]) return;
        if (threadIdx.x == 0 && a.nflagged) atomicAdd(a.nflagged, 1ull);
    }
    __shared__ cplx colk[2][GJ_N], rowk[2][GJ_N], piv[GJ_N];
    __shared__ int prow[GJ_N], invp[GJ_N], s_par;
    __shared__ cplx stage[16][GJ_N];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = (b & 1) ? a.nb : a.na;
    cplx *O = a.O + (long)b * a.ld * a.ld;
    const int tr = tid >> 5, tc = tid & 31;
    double vr[8][4], vi[8][4];
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int i = tr + 16 * x, j = tc + 32 * y;
            const cplx t = (i < n && j < n) ? O[(long)i * a.ld + j] : cmake(0.0, 0.0);
            vr[x][y] = t.x; vi[x][y] = t.y;
        }
    if (tid < GJ_N) { prow[tid] = tid; invp[tid] = tid; piv[tid] = cmake(1.0, 0.0); }
    if (tid == 0) s_par = 0;
    bool used0 = lane >= n, used1 = lane + 64 >= n;      // rows lane / lane + 64 no longer pivot candidates
#pragma unroll
    for (int y0 = 0; y0 < 4; ++y0) {
        for (int kk = 0; kk < 32; ++kk) {
            const int k = 32 * y0 + kk;
            if (k >= n) break;
            const int buf = k & 1;
            if (tc == kk) {
#pragma unroll
                for (int x = 0; x < 8; ++x) colk[buf][tr + 16 * x] = cmake(vr[x][y0], vi[x][y0]);
            }
            __syncthreads();                                                         // barrier A
            int p;
            {
                const cplx c0 = colk[buf][lane], c1 = colk[buf][lane + 64];
                const unsigned m0 = __float_as_uint((float)(fabs(c0.x) + fabs(c0.y)));
                const unsigned m1 = __float_as_uint((float)(fabs(c1.x) + fabs(c1.y)));
                const unsigned k0 = used0 ? 0u : ((((m0 >> 8) + 1u) << 7) | (unsigned)(127 - lane));
                const unsigned k1 = used1 ? 0u : ((((m1 >> 8) + 1u) << 7) | (unsigned)(63 - lane));
                const unsigned mx = wave_max_u32(k0 > k1 ? k0 : k1);
                p = 127 - (int)(mx & 127u);
                used0 = used0 || p == lane;
                used1 = used1 || p == lane + 64;
            }
            const bool own = wave == ((p & 15) >> 1);                               // wave-uniform
            if (own) {
                const cplx d = colk[buf][p];
                // reciprocal by v_rcp_f64 + two Newton steps (< 1 ulp off; the IEEE division sequence is a 12-deep dependent
                // chain on the critical path of every pivot step: see gj_block8 in gj_wave.h)
                const double nn = d.x * d.x + d.y * d.y;
                double dn = __builtin_amdgcn_rcp(nn);
                dn = fma(fma(-nn, dn, 1.0), dn, dn);
                dn = fma(fma(-nn, dn, 1.0), dn, dn);
                const double dix = d.x * dn, diy = -d.y * dn;
                const int x0 = p >> 4;
                const bool mine = tr == (p & 15);
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    if (x0 == x) {                                                   // scalar branch
                        if (mine) {
#pragma unroll
                            for (int y = 0; y < 4; ++y) {
                                const int j = tc + 32 * y;
                                const double rx = vr[x][y], ry = vi[x][y];
                                rowk[buf][j] = (j == k) ? cmake(dix, diy)
                                                        : cmake(rx * dix - ry * diy, rx * diy + ry * dix);
                            }
                        }
                    }
                }
                if (lane == 0) { piv[k] = d; prow[k] = p; invp[p] = k; }
            }
            __syncthreads();                                                         // barrier B
            double rkx[4], rky[4];
#pragma unroll
            for (int y = 0; y < 4; ++y) { const cplx t = rowk[buf][tc + 32 * y]; rkx[y] = t.x; rky[y] = t.y; }
            if (tc == kk) {
#pragma unroll
                for (int x = 0; x < 8; ++x) { vr[x][y0] = 0.0; vi[x][y0] = 0.0; }
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const cplx f = colk[buf][tr + 16 * x];
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    double ox = vr[x][y], oy = vi[x][y];
                    ox = fma(-f.x, rkx[y], ox); ox = fma(f.y, rky[y], ox);
                    oy = fma(-f.x, rky[y], oy); oy = fma(-f.y, rkx[y], oy);
                    vr[x][y] = ox; vi[x][y] = oy;
                }
            }
            if (own) {                                                               // row p <- scaled row
                const int x0 = p >> 4;
                const bool mine = tr == (p & 15);
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    if (x0 == x) {
                        if (mine) {
#pragma unroll
                            for (int y = 0; y < 4; ++y) { vr[x][y] = rkx[y]; vi[x][y] = rky[y]; }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    // inverse, un-permuted through LDS so that the global stores are row-contiguous:
    // W[i][j] belongs at (invp[i], prow[j])
    if (a.write_inverse) {
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int i = tr + 16 * x;
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const int j = tc + 32 * y;
                if (j < n) stage[tr][prow[j]] = cmake(vr[x][y], vi[x][y]);
            }
            __syncthreads();
            if (i < n) {
                cplx *dst = O + (long)invp[i] * a.ld;
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    const int c = tc + 32 * y;
                    if (c < n) dst[c] = stage[tr][c];
                }
            }
            __syncthreads();
        }
    }
    // det = sign(permutation k -> prow[k]) * prod of pivots; sign from the inversion count
    {
        const int i = tid & 127, q = tid >> 7;
        int inv = 0;
        if (i < n) {
            const int pi = prow[i];
            for (int j = 32 * q; j < 32 * q + 32; ++j)
                if (j > i && j < n && prow[j] < pi) ++inv;
        }
        if (inv & 1) atomicAdd(&s_par, 1);
    }
    __syncthreads();
    if (wave == 0) {
        const cplx d0 = piv[lane], d1 = piv[lane + 64];                // entries >= n are 1
        double px = d0.x * d1.x - d0.y * d1.y, py = d0.x * d1.y + d0.y * d1.x;
        int e;
        (void)frexp(fmax(fabs(px), fabs(py)), &e);
        px = ldexp(px, -e); py = ldexp(py, -e);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double qx = __shfl_xor(px, off), qy = __shfl_xor(py, off);
            const int qe = __shfl_xor(e, off);
            const double tx = px * qx - py * qy, ty = px * qy + py * qx;
            int e2;
            (void)frexp(fmax(fabs(tx), fabs(ty)), &e2);
            px = ldexp(tx, -e2); py = ldexp(ty, -e2);
            e += qe + e2;
        }
        if (lane == 0) {
            const double sg = (s_par & 1) ? -1.0 : 1.0;
            a.detm[b] = cmake(sg * px, sg * py);
            a.dete[b] = e;
        }
    }
}

// ------------------------------------------------------------------ 2b. blocked Gauss-Jordan on the matrix pipe (round 4)
// The step-by-step kernel above spends 4.4 k cycles per pivot on two barriers and a dependent LDS hand-over for 1 k cycles
// of FMAs (0.23 of the fp64 vector peak).  Here the pivots are 16 x 16 BLOCKS (one MFMA tile): block k is inverted by ONE
// wave with the register Gauss-Jordan of gj_wave.h (partial pivoting inside the block, no barrier; 1.2 k cycles per pivot),
// and the rest of a block step is rank-16 matrix arithmetic on MFMA:
//     P = O[k,k];  O[k,:] <- P^-1 O[k,:] (O[k,k] <- P^-1);  O[i,:] <- O[i,:] - O[i,k] O[k,:] (O[i,k] <- -O[i,k] P^-1), i != k
// which leaves the inverse in place after the last block; det O = prod det P_k (the P_k are the successive Schur
// complements).  No pivoting ACROSS blocks: a leading block whose condition is poor loses digits that the full partial
// pivoting of the kernel above keeps -- a matrix in which the smallest pivot of a block is below 1e-10 of its largest is
// flagged, left untouched, and redone by that kernel.
// One 512-thread work-group per matrix; the matrix lives in the registers of the 8 waves as 8 x 8 tiles of 16 x 16 in MFMA
// accumulator layout (element (4 r + lk, lr) of a tile in register r of lane (lk, lr) -- which is also the B-fragment
// layout of its four k-steps), wave w owning the tiles (I, (I + w) mod 8): one tile in every tile row and every tile
// column, so that every block step gives every wave the same work (seven tiles to update, one to scale) and the pivot
// tile always belongs to wave 0.  Per block step the block column (negated, A-fragment order: 32 KB), the block row
// (B-fragment order: 32 KB) and the pivot tile (row-major for the inverting wave, then A-fragment order: 4 KB) pass
// through LDS.
struct GjMfmaLds {
    // byte offsets: block column panels (two: the tiles of the next block column are published as soon as they are up to
    // date, while the current one is still being read), block row panel, the diagonal tiles (their home), the pivot tile
    // row-major for the inverting wave, its inverse in A-fragment order (double-buffered: written for step k + 1 while
    // step k reads its own), pivots and pivot rows of all eight pivot tiles (the determinant is evaluated once, at the end)
    static constexpr int FP = 0, RP = 65536, DG = 98304, PL = 131072, PA = 135168, PIV = 143360, PROW = 145408, SMALL = 145920;
    // + rowk, the flag, the growth guard: largest |entry| of block row / block column k by wave [2][8][8], lane maxima of
    // |P_k^-1| [8][64]
    static constexpr int BMAX = SMALL + 512 + 64, PMAX = BMAX + 512;
    static constexpr int BYTES = PMAX + 2048;
};

// rank-16 product of one tile: acc += A-fragments (4 k-steps of `ap`) x B-fragments (`bp`); complex by THREE real MFMAs per
// k-step (the phase is bound by the matrix pipe: two waves per SIMD): p1 = sum ar br, p2 = sum ai bi,
// p3 = acc_i + sum (ar + ai)(br + bi); then acc_r += p1 - p2, acc_i = p3 - p1 - p2.  The fragments of two k-steps at a time
// (four LDS reads in flight, then six MFMAs).
__device__ __attribute__((always_inline)) inline void gj_tile_mac(const d2_t *ap, const d2_t *bp, int lane, int nks, d4_t &accr, d4_t &acci) {
    d4_t p1 = (d4_t){0, 0, 0, 0}, p2 = (d4_t){0, 0, 0, 0};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        d2_t av[2], bv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { av[q] = ap[(2 * half + q) * 64 + lane]; bv[q] = bp[(2 * half + q) * 64 + lane]; }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (2 * half + q < nks) {
                p1 = mfma16(av[q][0], bv[q][0], p1);
                p2 = mfma16(av[q][1], bv[q][1], p2);
                acci = mfma16(av[q][0] + av[q][1], bv[q][0] + bv[q][1], acci);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { accr[r] += p1[r] - p2[r]; acci[r] -= p1[r] + p2[r]; }
}

// Work split (round 4, second version): wave 0 is the PIVOT wave -- it owns the eight diagonal tiles, which live in LDS,
// and does nothing but bring the next pivot tile up to date, invert it (register Gauss-Jordan, one wave) and publish the
// inverse, one block step AHEAD of the other seven waves, whose rank-16 update of step k runs meanwhile (look-ahead: the
// inversion is the longest dependent chain of a step and nothing else waits for it any more).  Waves 1..7 own the tiles
// (I, (I + w) mod 8) in registers -- one per tile row and tile column -- and share the updates of the diagonal tiles that
// are not the next pivot.  Two barriers per block step: the row scaling of a wave's tile of the block row needs nothing
// but P^-1 (published a step ahead) and the wave's own registers.  Global traffic overlaps the first and the last block
// step: step 0 starts when its block row and column have arrived and takes the other tiles in the order they were
// requested; every tile is stored right after its last update.
typedef unsigned int gj_u4_t __attribute__((ext_vector_type(4)));
typedef unsigned int gj_u2_t __attribute__((ext_vector_type(2)));

__device__ inline void gj_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

__global__ __launch_bounds__(512) void gj_mfma_kernel(GjArgs a, int *flag) {
    extern __shared__ __align__(16) unsigned char gsm[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, lr = lane & 15;
    const int n = (b & 1) ? a.nb : a.na;
    const int nt16 = (n + 15) >> 4;                                    // tile rows / columns that exist
    const int nsteps = (a.dbg & 8) ? 0 : nt16;
    cplx *O = a.O + (long)b * a.ld * a.ld;
    d2_t *Fp = (d2_t *)(gsm + GjMfmaLds::FP), *Rp = (d2_t *)(gsm + GjMfmaLds::RP), *Dg = (d2_t *)(gsm + GjMfmaLds::DG);
    cplx *Pl = (cplx *)(gsm + GjMfmaLds::PL);
    d2_t *Pa = (d2_t *)(gsm + GjMfmaLds::PA);                          // [2][4 k-steps][64]
    cplx *piv_all = (cplx *)(gsm + GjMfmaLds::PIV);                    // [8 pivot tiles][16]
    int *prow_all = (int *)(gsm + GjMfmaLds::PROW);
    cplx *rowk = (cplx *)(gsm + GjMfmaLds::SMALL);
    int *s_bad = (int *)(rowk + 32);
    unsigned *bmax = (unsigned *)(gsm + GjMfmaLds::BMAX), *pmaxp = (unsigned *)(gsm + GjMfmaLds::PMAX);
    // largest |re|, |im| of a tile as the high word of the double (positive doubles order like their bit patterns: integer
    // maxima of the sign-stripped high words, no fp64 operation), wave-wide.  A NaN counts as larger than everything.
    auto tile_max = [&](const d4_t &tr, const d4_t &ti) {
        unsigned key = 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned hr = (unsigned)__double2hiint(tr[r]) & 0x7fffffffu, hi = (unsigned)__double2hiint(ti[r]) & 0x7fffffffu;
            key = key > hr ? key : hr;
            key = key > hi ? key : hi;
        }
        return gj_wave_max_u32(key);
    };
#ifdef AFQ_TUNING
    auto stamp = [&](int kb, int pt) {
        if (a.ts && b == 0) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
            if (lane == 0) a.ts[(wave * 8 + kb) * 8 + pt] = t;
        }
    };
#define GJ_STAMP(kb, pt) stamp(kb, pt)
#else
#define GJ_STAMP(kb, pt)
#endif
    // B-fragments of block column J in step kb: the scaled block row, or P^-1 itself (its home) for the pivot's own column
    // (the update then turns O[i, k] into -O[i, k] P^-1)
    auto bpanel = [&](int J, int kb) { return J == kb ? Dg + kb * 256 : Rp + J * 256; };
    auto apanel = [&](int I, int kb) { return Fp + (kb & 1) * 2048 + I * 256; };       // A-fragments of tile row I in step kb

    if (wave == 0) {
        // ================================================================= pivot wave
        __builtin_amdgcn_s_setprio(3);                                 // its chain is the block step's critical path
        // inverts the pivot tile `t` (row-major in Pl, compact nb x nb), leaves P^-1 in A-fragment order in Pa[t & 1] and
        // in accumulator (= B-fragment) layout in Dg[t]; pivots and pivot rows stay in piv_all / prow_all
        auto invert = [&](const int t, const int nb) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(a.dbg & 1)) gj_wave16q_inv(Pl, nb, lane, rowk, piv_all + 16 * t, prow_all + 16 * t);
            GJ_STAMP(t ? t - 1 : 0, t ? 4 : 7);
            d2_t *pa = Pa + (t & 1) * 256;
            double m = 0.0;
            bool isnan = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = lane + 64 * q, row = e >> 4, col = e & 15;
                const cplx v = (row < nb && col < nb) ? Pl[row * nb + col] : cmake(0.0, 0.0);
                pa[(col >> 2) * 64 + (col & 3) * 16 + row] = (d2_t){v.x, v.y};
                Dg[(t * 4 + (row >> 2)) * 64 + (row & 3) * 16 + col] = (d2_t){v.x, v.y};
                m = fmax(m, fmax(fabs(v.x), fabs(v.y)));
                isnan = isnan || v.x != v.x || v.y != v.y;
            }
            pmaxp[t * 64 + lane] = isnan ? 0x7ff80000u : (unsigned)__double2hiint(m);   // (reduced once, in finish())
        };
        // after the last pivot tile, before the stores of the last block step -- is the blocked result to be trusted?
        //   pivot spread: a tile whose pivots span more than ten decades;
        //   growth guard: pivoting is confined to the pivot tile, so the multipliers O[i, k] P_k^-1 are not bounded by one;
        //     with g = max_k max|P_k^-1| max|block row and column k| the rounding error of a step is about g times that of
        //     partial pivoting -- above 1e5 the matrix is flagged (a leading tile of tiny or vanishing entries in a well-
        //     conditioned matrix, e.g. a walker whose first orbitals are orthogonal to the trial's first orbitals).
        // A flagged matrix is redone by the step-by-step kernel; nothing of it is stored here.  Integer DPP maxima of the
        // high words of the doubles: this sits between two block steps, on everybody's critical path.
        auto row16_max = [&](unsigned v) {       // lane 15 of every row of 16 lanes: maximum of the row
#define AFQ_DPP_MAX(ctrl) { const unsigned t_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, false); v = v > t_ ? v : t_; }
            AFQ_DPP_MAX(0x111) AFQ_DPP_MAX(0x112) AFQ_DPP_MAX(0x114) AFQ_DPP_MAX(0x118)
#undef AFQ_DPP_MAX
            return v;
        };
        auto judge = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bool bad = false;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = (lane >> 4) + 4 * h, k = lane & 15;
                const int nb = n - 16 * t < 16 ? n - 16 * t : 16;          // (<= 0 for tiles that do not exist)
                const bool ok = k < nb && !(a.dbg & 1);
                const cplx d = ok ? piv_all[16 * t + k] : cmake(1.0, 0.0);
                const double pm = fabs(d.x) + fabs(d.y);
                const unsigned key = pm != pm ? 0x7ff80000u : (unsigned)__double2hiint(pm);
                const unsigned kmax = row16_max(ok ? key : 0u), kmin = ~row16_max(ok ? ~key : 0u);
                // (high words: the comparison is good to 2^-20, the threshold is a decade count)
                const double pmax = __hiloint2double((int)kmax, 0), pmin = __hiloint2double((int)kmin, 0);
                bad = bad || (k == 15 && nb > 0 && !(pmin >= 1e-10 * pmax && pmax > 0.0));
            }
            unsigned mypk = 0u;
            for (int t = 0; t < nt16; ++t) {
                const unsigned pk = gj_wave_max_u32(pmaxp[t * 64 + lane]);
                mypk = (lane >> 3) == t ? pk : mypk;
            }
            {
                const int t = lane >> 3, w = lane & 7;
                unsigned bk = bmax[t * 8 + w], bc = bmax[64 + t * 8 + w];
                bk = bk > bc ? bk : bc;
                bk = row16_max(bk);             // (lane 8 t + 7: the shifts that reach it stay inside its group of eight... 
                                                //  except row_shr:8, which can only raise the maximum of an odd group by
                                                //  its even neighbour's: a conservative guard, not a wrong one)
                const double g = __hiloint2double((int)mypk, 0) * __hiloint2double((int)bk, 0);
                bad = bad || (w == 7 && t < nt16 && !(g <= 1e5));
            }
            const bool any = __ballot(bad) != 0ull || (a.dbg & 64);
            if (lane == 0) { *s_bad = any ? 1 : 0; flag[b] = any ? 1 : 0; }
        };
        // determinant = product of all pivots x parity of every tile's pivot order (the pivot wave is idle by then)
        auto determinant = [&]() {
            double px = 1.0, py = 0.0;
            int ex = 0, odd = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = (lane >> 4) + 4 * h, k = lane & 15;
                const int nb = n - 16 * t < 16 ? n - 16 * t : 16;
                const bool ok = k < nb && !(a.dbg & 1);
                const cplx d = ok ? piv_all[16 * t + k] : cmake(1.0, 0.0);
                int inv = 0;
                if (ok) {
                    const int pr = prow_all[16 * t + k];
                    for (int j = k + 1; j < nb; ++j) inv += prow_all[16 * t + j] < pr ? 1 : 0;
                }
                odd ^= inv & 1;
                const double tx = px * d.x - py * d.y, ty = px * d.y + py * d.x;
                int e2;
                (void)frexp(fmax(fabs(tx), fabs(ty)), &e2);
                px = ldexp(tx, -e2); py = ldexp(ty, -e2); ex += e2;
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double qx = __shfl_xor(px, off), qy = __shfl_xor(py, off);
                const int qe = __shfl_xor(ex, off);
                const double tx = px * qx - py * qy, ty = px * qy + py * qx;
                int e2;
                (void)frexp(fmax(fabs(tx), fabs(ty)), &e2);
                px = ldexp(tx, -e2); py = ldexp(ty, -e2);
                ex += qe + e2;
            }
            const double sg = (__popcll(__ballot(odd != 0)) & 1) ? -1.0 : 1.0;
            if (lane == 0) { a.detm[b] = cmake(sg * px, sg * py); a.dete[b] = ex; }
        };
        // prologue: the diagonal tiles to their home, the first pivot tile inverted while the others arrive
        {
            cplx dt[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * i + 4 * r + lk, col = 16 * i + lr;
                    dt[i][r] = (row < n && col < n) ? O[(long)row * a.ld + col] : cmake(0.0, 0.0);
                }
            const int nb0 = n < 16 ? n : 16;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * r + lk < nb0 && lr < nb0) Pl[(4 * r + lk) * nb0 + lr] = dt[0][r];
            invert(0, nb0);
#pragma unroll
            for (int i = 1; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) Dg[(i * 4 + r) * 64 + lane] = (d2_t){dt[i][r].x, dt[i][r].y};
            // guard slots of steps 1..7 (and wave 0's own column): zero until a tile wave writes its maximum (after P);
            // step 0's slots are written by the tile waves themselves, before P
            for (int e = lane; e < 128; e += 64)
                if (((e & 63) >> 3) != 0 || (e & 7) == 0) bmax[e] = 0u;
        }
        gj_lds_barrier();                                              // P
        for (int kb = 0; kb < nsteps; ++kb) {
            const int nblk = n - 16 * kb < 16 ? n - 16 * kb : 16;
            const int nks = (nblk + 3) >> 2;
            GJ_STAMP(kb, 0);
            if (kb == nt16 - 1) judge();                               // (every pivot tile is inverted, every guard slot written)
            GJ_STAMP(kb, 1);
            gj_lds_barrier();                                          // B1
            GJ_STAMP(kb, 2);
            // the next pivot tile up to date, inverted, published -- while the other waves update everything else
            const int nx = kb + 1;
            if (nx < nt16) {
                const int nbx = n - 16 * nx < 16 ? n - 16 * nx : 16;
                d4_t accr, acci;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const d2_t v = Dg[(nx * 4 + r) * 64 + lane]; accr[r] = v[0]; acci[r] = v[1]; }
                gj_tile_mac(apanel(nx, kb), Rp + nx * 256, lane, nks, accr, acci);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * r + lk < nbx && lr < nbx) Pl[(4 * r + lk) * nbx + lr] = cmake(accr[r], acci[r]);
                GJ_STAMP(kb, 3);
                invert(nx, nbx);
            } else {                                                   // last step: this pivot's inverse is final
                if (a.write_inverse && !*s_bad) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * kb + 4 * r + lk, col = 16 * kb + lr;
                        const d2_t v = Dg[(kb * 4 + r) * 64 + lane];
                        if (row < n && col < n) O[(long)row * a.ld + col] = cmake(v[0], v[1]);
                    }
                }
                determinant();
            }
            GJ_STAMP(kb, 5);
            gj_lds_barrier();                                          // B4
            GJ_STAMP(kb, 6);
        }
        return;
    }
    // ===================================================================== tile waves 1..7: tile i is (I = i, J = (i + wave) & 7)
    d4_t Cr[8], Ci[8];
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(O, 0, a.ld * a.ld * (int)sizeof(cplx), 0x00020000);
    const int lane_off = (lk * a.ld + lr) * (int)sizeof(cplx);
    // (buffer loads: lanes outside the matrix ask for an offset beyond the descriptor's range and get zeros -- no branch
    //  around a load, so the loads retire in the order they were requested and step 0 waits only for what it uses)
    auto load_tile = [&](int I, int J, d4_t &tr, d4_t &ti) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + 4 * r + lk, col = 16 * J + lr;
            const int uo = ((16 * I + 4 * r) * a.ld + 16 * J) * (int)sizeof(cplx);
            // (real and imaginary part by separate 8-byte loads, straight into the accumulator registers: a 16-byte load
            //  needs two moves per row that the compiler places in front of the block-step loop, behind a wait for all loads)
            const int vo = (row < n && col < n) ? lane_off : 0x7fffffff;
            tr[r] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(orsrc, vo, uo, 0));
            ti[r] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(orsrc, vo, uo + 8, 0));   // (+ 8 in the scalar
                                                                   // offset: as an immediate the two loads are merged again)
        }
    };
    // (buffer stores: descriptor + one 32-bit lane offset + a scalar offset per row of a tile; as 64-bit addresses the 32
    //  rows of a wave's tiles are hoisted out of the block-step loop into 64 VGPRs that the loop does not have)
    auto store_tile = [&](int I, int J, const d4_t &tr, const d4_t &ti) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + 4 * r + lk, col = 16 * J + lr;
            const int uo = ((16 * I + 4 * r) * a.ld + 16 * J) * (int)sizeof(cplx);
            if (row < n && col < n)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gj_u4_t, (d2_t){tr[r], ti[r]}), orsrc, lane_off, uo, 0);
        }
    };
    // block column tile (negated) -> Fp in A-fragment order
    auto publish_col = [&](int I, int kb, const d4_t &tr, const d4_t &ti) {
        d2_t *fp = apanel(I, kb);
#pragma unroll
        for (int r = 0; r < 4; ++r) fp[(lr >> 2) * 64 + (lr & 3) * 16 + 4 * r + lk] = (d2_t){-tr[r], -ti[r]};
    };
    // requests in the order of first use: step 0's tile of the block row, its tile of the block column (a second copy, so
    // that nothing else has to have arrived when it is published), then the tiles in the order step 0 updates them
    {
        const int i0 = (8 - wave) & 7;                                 // tile (i0, 0) is this wave's
        load_tile(0, wave, Cr[0], Ci[0]);
        d4_t f0r, f0i;
        load_tile(i0, 0, f0r, f0i);
        __builtin_amdgcn_sched_barrier(0);                             // (the scheduler would request these two last)
#pragma unroll
        for (int i = 1; i < 8; ++i) { load_tile(i, (i + wave) & 7, Cr[i], Ci[i]); __builtin_amdgcn_sched_barrier(0); }
        if (i0 < nt16) publish_col(i0, 0, f0r, f0i);
        const unsigned mc = tile_max(f0r, f0i), mr = tile_max(Cr[0], Ci[0]);       // (zeros for tiles outside the matrix)
        if (lane == 0) { bmax[wave] = mr; bmax[64 + wave] = mc; }
    }
    gj_lds_barrier();                                                  // P
    for (int kb = 0; kb < nsteps; ++kb) {
        const int nblk = n - 16 * kb < 16 ? n - 16 * kb : 16;
        const int nks = (nblk + 3) >> 2;                               // k-steps of 4 that hold anything
        const int Jk = (kb + wave) & 7;                                // this wave's tile of the block row is (kb, Jk)
        GJ_STAMP(kb, 0);
        // ---- row scaling R' = P^-1 R of this wave's tile of the block row, straight from its registers (the accumulator
        //      layout IS the B-fragment order: register ks = k-step ks) -> Rp and back into the registers
        if (Jk < nt16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i == kb) {                                         // wave-uniform; static register index
                    const d2_t *pa = Pa + (kb & 1) * 256;
                    d2_t av[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) av[ks] = pa[ks * 64 + lane];
                    d4_t Nr = (d4_t){0, 0, 0, 0}, Ni = (d4_t){0, 0, 0, 0}, p2 = (d4_t){0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        if (ks < nks) {                                // three products per k-step, as in gj_tile_mac
                            Nr = mfma16(av[ks][0], Cr[i][ks], Nr);
                            p2 = mfma16(av[ks][1], Ci[i][ks], p2);
                            Ni = mfma16(av[ks][0] + av[ks][1], Cr[i][ks] + Ci[i][ks], Ni);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { Ni[r] -= Nr[r] + p2[r]; Nr[r] -= p2[r]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) Rp[(Jk * 4 + r) * 64 + lane] = (d2_t){Nr[r], Ni[r]};
                    Cr[i] = Nr; Ci[i] = Ni;
                    if (Jk == kb + 1) {                                // (wave 1: row kb of the next step's block column)
                        publish_col(i, kb + 1, Nr, Ni);
                        const unsigned mc = tile_max(Nr, Ni);
                        if (lane == 0) bmax[64 + (kb + 1) * 8 + wave] = mc;
                    }
                }
            }
        }
        GJ_STAMP(kb, 1);
        gj_lds_barrier();                                              // B1
        GJ_STAMP(kb, 2);
        const bool st = kb == nt16 - 1 && a.write_inverse && !*s_bad; // last step: every tile is final after its update
        // ---- rank-16 update of every tile outside the block row: C <- (J == kb ? 0 : C) + (-F) R'
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int J = (i + wave) & 7;
            if (i < nt16 && J < nt16 && i != kb && !(a.dbg & 2)) {
                d4_t accr = Cr[i], acci = Ci[i];
                if (J == kb) { accr = (d4_t){0, 0, 0, 0}; acci = (d4_t){0, 0, 0, 0}; }
                gj_tile_mac(apanel(i, kb), bpanel(J, kb), lane, nks, accr, acci);
                Cr[i] = accr; Ci[i] = acci;
                if (st) store_tile(i, J, accr, acci);
                if (J == kb + 1 || i == kb + 1) {                      // wave-uniform: the next step's block column / row
                    if (J == kb + 1) publish_col(i, kb + 1, accr, acci);
                    const unsigned m = tile_max(accr, acci);
                    if (lane == 0) bmax[(J == kb + 1 ? 64 : 0) + (kb + 1) * 8 + wave] = m;
                }
            } else if (i == kb && J < nt16 && st) store_tile(i, J, Cr[i], Ci[i]);   // (the scaled block row)
        }
        GJ_STAMP(kb, 3);
        // ... and the diagonal tiles that are neither this pivot nor the next one (that is wave 0's): one each for waves
        // 1, 2, 3, 5, 6, 7 -- not wave 4, the pivot wave's neighbour on its SIMD: the register Gauss-Jordan runs 17 k cycles
        // beside 12 k cycles of fp64 MFMAs and 21.5 k beside 23 k (all six tiles on wave 4), 12.5 k alone
        // (a load inside the loop whose value the loop uses -- an out-of-range buffer load: zero, no memory access: in front
        //  of a loop that stores, loads nothing and uses registers with loads in flight the compiler waits for ALL
        //  outstanding loads, which is exactly the wait step 0 is arranged to avoid.  Here, behind the last use of a tile,
        //  because loads retire in order.)
        const int zero = __builtin_amdgcn_raw_buffer_load_b32(orsrc, 0x7fffffff, 0, 0);
        for (int i = 0; i < nt16; ++i) {
            const int rank = i - (i > kb ? 1 : 0) - (i > kb + 1 ? 1 : 0);      // 0..5 among the tiles that are updated here
            const int owner = rank < 3 ? rank + 1 : rank < 6 ? rank + 2 : 4;   // (0..6 in the last step: there is no next pivot)
            if (i == kb || i == kb + 1 || wave != owner + zero || (a.dbg & 4)) continue;
            d4_t accr, acci;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const d2_t v = Dg[(i * 4 + r) * 64 + lane]; accr[r] = v[0]; acci[r] = v[1]; }
            gj_tile_mac(apanel(i, kb), Rp + i * 256, lane, nks, accr, acci);
            if (st) store_tile(i, i, accr, acci);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) Dg[(i * 4 + r) * 64 + lane] = (d2_t){accr[r], acci[r]};
            }
        }
        GJ_STAMP(kb, 5);
        gj_lds_barrier();                                              // B4: panels free for the next block step
        GJ_STAMP(kb, 6);
    }
}

// (wa.weight != null: det IS wa.ovlp_new and the walker's weight update, weight cap and estimator terms run right behind
//  its determinant, as in greens_small_kernel: no separate weight_kernel launch)
__global__ void det_combine_kernel(const cplx *detm, const int *dete, cplx *det, cplx *det_a, int nw, WeightArgs wa) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    const cplx p = cmul(detm[2 * w], detm[2 * w + 1]);
    const int e = dete[2 * w] + dete[2 * w + 1];
    det[w] = cmake(ldexp(p.x, e), ldexp(p.y, e));
    if (det_a) det_a[w] = cmake(ldexp(detm[2 * w].x, dete[2 * w]), ldexp(detm[2 * w].y, dete[2 * w]));
    if (wa.weight) weight_update_and_cap(wa, w);
}

// N > 45, or a smaller determinant whose walker does not fit the one-work-group kernel of k_small.hip (M > 128, or walker +
// overlap matrices above 160 KB of LDS: e.g. 45 + 45 electrons on 100 sites)
int k_greens_big_supported(afq_handle *h) {
    const int nmax = h->na > h->nb ? h->na : h->nb;
    const size_t lds_small = sizeof(cplx) * (2 * ((size_t)nmax * nmax + 2 * nmax) + ((2 * nmax + 3) / 4 + 1) + (size_t)h->M * h->nt);
    const bool small_fits = nmax <= 45 && h->M <= 128 && lds_small <= 160 * 1024;
    return !small_fits && nmax > 16 && nmax <= GJ_N && h->nb > 0 && !h->no_ring;
}

int k_greens_big(afq_handle *h, cplx *ghalf, cplx *det, cplx *oinv, const WeightArgs *wa_in) {
    const int nmax = h->na > h->nb ? h->na : h->nb;
    const int nb2 = 2 * h->nw;
    const size_t wsn = (size_t)nb2 * nmax * nmax;
    if (!h->big_ws) AFQ_HIP(h, hipMalloc(&h->big_ws, sizeof(cplx) * wsn));
    if (!h->detm) {
        AFQ_HIP(h, hipMalloc(&h->detm, sizeof(cplx) * nb2));
        AFQ_HIP(h, hipMalloc(&h->dete, sizeof(int) * nb2));
    }
    auto overlap = [&](auto p) -> int {
        p.batch = nb2; p.rows = nmax; p.cols = nmax; p.kdim = h->M;
        p.nt = h->nt; p.na = h->na; p.nb = h->nb; p.ld = nmax;
        p.phi = h->phi; p.psic = h->psic; p.psi_stride = h->psi_stride; p.O = h->big_ws; p.zero = (const cplx *)h->zero_page;
#ifdef AFQ_TUNING
        if (AFQ_KNOB_SET("AFQ_OVLP_CFG")) AFQ_GEMM_AS(h, "k_greens_big: OvlpProb GEMM", (launch_mfma_gemm_wg<4, 2, 1, 4, 4, decltype(p), MAP_COLS_FAST, true, 1, 2>(p, h->stream, h->zero_page)));
        else if (AFQ_KNOB_SET("AFQ_BIG_LEAN")) AFQ_GEMM_AS(h, "k_greens_big: OvlpProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, decltype(p), MAP_COLS_FAST, true, 1, 5, 4>(p, h->stream, h->zero_page)));
        else if (AFQ_KNOB_SET("AFQ_BIG_WPE")) AFQ_GEMM_AS(h, "k_greens_big: OvlpProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, decltype(p), MAP_COLS_FAST, true, 1, 3, 4>(p, h->stream, h->zero_page)));
        else if (AFQ_KNOB_SET("AFQ_BIG_NOLOADER")) AFQ_GEMM_AS(h, "k_greens_big: OvlpProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, decltype(p), MAP_COLS_FAST, true>(p, h->stream, h->zero_page)));
        else
#endif
        AFQ_GEMM_AS(h, "k_greens_big: OvlpProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, decltype(p), MAP_COLS_FAST, true, 1, 3>(p, h->stream, h->zero_page)));
        return AFQ_OK;
    };
    {
        const int rc = (h->psi_real && h->psi_stride == 0 && h->ndet == 1) ? overlap(OvlpProbT<true>()) : overlap(OvlpProbT<false>());
        if (rc) return rc;
    }
    {
        GjArgs a;
        a.na = h->na; a.nb = h->nb; a.ld = nmax; a.write_inverse = ghalf != nullptr || oinv != nullptr;
        a.O = h->big_ws; a.detm = h->detm; a.dete = h->dete;
        // blocked Gauss-Jordan on the matrix pipe for more than one block of 32; the step-by-step kernel for matrices it flags
        // as poorly conditioned block-wise (none in any test or benchmark so far) and for tuning builds that ask for it
        const bool blocked = nmax > 32 && !AFQ_KNOB_SET("AFQ_GJ_STEPWISE");     // (up to 32: one leaf would do all the work)
        if (blocked) {
            static size_t lds_set[AFQ_MAX_DEVICES] = {0};
            AFQ_HIP(h, afq_raise_lds((const void *)gj_mfma_kernel, GjMfmaLds::BYTES, lds_set));
            if (!h->gj_flag) AFQ_HIP(h, hipMalloc(&h->gj_flag, sizeof(int) * nb2));
            a.dbg = AFQ_KNOB_INT("AFQ_GJ_DBG", 0);
#ifdef AFQ_TUNING
            static unsigned long long *ts_dev = nullptr;
            static int ts_launch = 0;
            if (AFQ_KNOB_SET("AFQ_GJ_TS")) {
                if (!ts_dev) { hipMalloc(&ts_dev, 512 * 8); hipMemset(ts_dev, 0, 512 * 8); }
                a.ts = ts_dev;
            }
#endif
            AFQ_LAUNCH(h, gj_mfma_kernel, dim3(nb2), dim3(512), GjMfmaLds::BYTES, h->stream, a, h->gj_flag);
            AFQ_POST(h);
#ifdef AFQ_TUNING
            if (a.ts && ++ts_launch == 30) {
                unsigned long long t[512];
                hipStreamSynchronize(h->stream);
                hipMemcpy(t, a.ts, sizeof(t), hipMemcpyDeviceToHost);
                const unsigned long long t0 = t[0];
                for (int kb = 0; kb < 8; ++kb)
                    for (int wv = 0; wv < 8; ++wv) {
                        const unsigned long long *o = t + (wv * 8 + kb) * 8;
                        fprintf(stderr, "GJ_TS step %d wave %d: start %+7lld | to B1 %5lld | in B1 %5lld | %s %5lld | %s %5lld | %s %5lld | in B4 %5lld\n", kb, wv,
                                (long long)(o[0] - t0), (long long)(o[1] - o[0]), (long long)(o[2] - o[1]),
                                wv ? "tiles" : "upd next", (long long)(o[3] - o[2]),
                                wv ? "-" : "leaf", wv ? 0ll : (long long)(o[4] - o[3]),
                                wv ? "diag share" : "conv", (long long)(o[5] - (wv ? o[3] : o[4])), (long long)(o[6] - o[5]));
                    }
            }
#endif
            a.only = h->gj_flag;                                      // (work-groups of unflagged matrices return at once)
            a.nflagged = h->counters ? h->counters + 2 : nullptr;
            afq_note_launch(h, "gj_big_kernel (fallback pass)");      // (its own name in launch traces: priced by the matrices it processes)
            hipLaunchKernelGGL(gj_big_kernel, dim3(nb2), dim3(512), 0, h->stream, a);
        } else {
            AFQ_LAUNCH(h, gj_big_kernel, dim3(nb2), dim3(512), 0, h->stream, a);
        }
        AFQ_POST(h);
        WeightArgs wa;
        if (wa_in) wa = *wa_in;
        else std::memset(&wa, 0, sizeof(wa));
        AFQ_LAUNCH(h, det_combine_kernel, dim3((h->nw + 127) / 128), dim3(128), 0, h->stream, h->detm,
                           h->dete, det, h->det_a_out, h->nw, wa);
        AFQ_POST(h);
    }
    if (oinv)    // [nw, 2, nmax, nmax]: the layout of the workspace (batch = 2 w + spin)
        AFQ_HIP(h, hipMemcpyAsync(oinv, h->big_ws, sizeof(cplx) * wsn, hipMemcpyDeviceToDevice, h->stream));
    if (ghalf) {
        auto run = [&](auto p) -> int {
            p.batch = nb2; p.rows = nmax; p.cols = h->M; p.kdim = nmax;
            p.nt = h->nt; p.na = h->na; p.nb = h->nb; p.ld = nmax; p.M = h->M;
            p.Oinv = h->big_ws; p.phi = h->phi; p.ghalf = ghalf; p.zero = (const cplx *)h->zero_page;
            p.psicT = h->psicT; p.gdiag = h->gdiag; p.nparts = h->gdiag_parts;
            p.skip_store = 0;
            if (decltype(p)::COLDOT && h->ghalf_skip_store) { p.skip_store = 1; h->ghalf_skipped = true; }
#ifdef AFQ_TUNING
            if (AFQ_KNOB_SET("AFQ_GHALF_CFG")) AFQ_GEMM_AS(h, "k_greens_big: GhalfProb GEMM", (launch_mfma_gemm_wg<4, 2, 1, 4, 4, decltype(p), MAP_COLS_FAST, true, 1, 2>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_BIG_LEAN")) AFQ_GEMM_AS(h, "k_greens_big: GhalfProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, decltype(p), MAP_COLS_FAST, true, 1, 5, 4>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_BIG_WPE")) AFQ_GEMM_AS(h, "k_greens_big: GhalfProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, decltype(p), MAP_COLS_FAST, true, 1, 3, 4>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_BIG_NOLOADER")) AFQ_GEMM_AS(h, "k_greens_big: GhalfProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, decltype(p), MAP_COLS_FAST, true>(p, h->stream, h->zero_page)));
            else
#endif
            AFQ_GEMM_AS(h, "k_greens_big: GhalfProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, decltype(p), MAP_COLS_FAST, true, 1, 3>(p, h->stream, h->zero_page)));
            return AFQ_OK;
        };
        // Hubbard, the walkers' own Ghalf, shared single-determinant trial: the diagonal of G comes along
        const bool want_diag = h->kind == AFQ_SYS_HUBBARD && ghalf == h->ghalf && h->ndet == 1 && h->psi_stride == 0 && h->psicT;
        if (want_diag) {
            // partial sums per row block of Ghalf (a wave block covers 16 TM rows: 32 with the 2 x 2 waves of 2 x 2 tiles) or,
            // from the real-trial product below, per 16 columns of W: the buffer holds the larger count, gdiag_parts is what
            // the last writer used
            const int parts16 = (nmax + 15) / 16;
            if (!h->gdiag) AFQ_HIP(h, hipMalloc(&h->gdiag, sizeof(cplx) * (size_t)nb2 * parts16 * h->M));
            if (h->ghalf_skip_store && h->psi_real && !AFQ_KNOB_SET("AFQ_NO_GDIAG_REAL")) {
                // only diag G is wanted and the trial is real: W = conj(psi) O^-1 (real by complex), rowdot with phi
                GdiagProbT p;
                p.batch = nb2; p.rows = h->M; p.cols = nmax; p.kdim = nmax;
                p.nt = h->nt; p.na = h->na; p.nb = h->nb; p.ld = nmax; p.M = h->M;
                p.psic = h->psic; p.Oinv = h->big_ws; p.phi = h->phi; p.gdiag = h->gdiag; p.nparts = parts16;
                p.zero = (const cplx *)h->zero_page;
                h->gdiag_parts = parts16;
                h->ghalf_skipped = true;
#ifdef AFQ_TUNING
                const int gc = AFQ_KNOB_INT("AFQ_GDIAG_CFG", 0);
                if (gc == 1) AFQ_GEMM_AS(h, "k_greens_big: GdiagProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 4, 4, GdiagProbT, MAP_COLS_FAST, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (gc == 2) AFQ_GEMM_AS(h, "k_greens_big: GdiagProb GEMM", (launch_mfma_gemm_wg<4, 2, 2, 4, 4, GdiagProbT, MAP_COLS_FAST, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (gc == 3) AFQ_GEMM_AS(h, "k_greens_big: GdiagProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GdiagProbT, MAP_COLS_FAST, false, 1, 2>(p, h->stream, h->zero_page)));
                else if (gc == 4) AFQ_GEMM_AS(h, "k_greens_big: GdiagProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 4, 4, GdiagProbT, MAP_COLS_FAST, false, 1, 2>(p, h->stream, h->zero_page)));
                else if (gc == 5) AFQ_GEMM_AS(h, "k_greens_big: GdiagProb GEMM", (launch_mfma_gemm_wg<4, 2, 2, 4, 2, GdiagProbT, MAP_COLS_FAST, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (gc == 6) AFQ_GEMM_AS(h, "k_greens_big: GdiagProb GEMM", (launch_mfma_gemm_wg<4, 2, 1, 4, 4, GdiagProbT, MAP_COLS_FAST, false, 1, 2>(p, h->stream, h->zero_page)));
                else if (gc == 8) AFQ_GEMM_AS(h, "k_greens_big: GdiagProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GdiagProbT, MAP_BATCH_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (gc == 9) AFQ_GEMM_AS(h, "k_greens_big: GdiagProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GdiagProbT, MAP_BATCH_XCD_ROWS, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (gc == 10) AFQ_GEMM_AS(h, "k_greens_big: GdiagProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 2, GdiagProbT, MAP_COLS_FAST, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (gc == 11) AFQ_GEMM_AS(h, "k_greens_big: GdiagProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 8, GdiagProbT, MAP_COLS_FAST, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (gc == 7) AFQ_GEMM_AS(h, "k_greens_big: GdiagProb GEMM", (launch_mfma_gemm_wg<4, 2, 2, 4, 4, GdiagProbT, MAP_COLS_FAST, false, 1, 2>(p, h->stream, h->zero_page)));
                else
#endif
                AFQ_GEMM_AS(h, "k_greens_big: GdiagProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GdiagProbT, MAP_COLS_FAST, false, 1, 3>(p, h->stream, h->zero_page)));
                h->gdiag_version = h->ghalf_version;
                return AFQ_OK;
            }
            h->gdiag_parts = AFQ_KNOB_SET("AFQ_GHALF_CFG") ? parts16 : (nmax + 31) / 32;
            const int rc = run(GhalfProbT<true>());
            if (rc) return rc;
            h->gdiag_version = h->ghalf_version;
        } else {
            const int rc = run(GhalfProbT<false>());
            if (rc) return rc;
        }
    }
    return AFQ_OK;
}

// ==========================================================================================
// Re-orthogonalisation for large N by Cholesky-QR2 (walkers/single_det.py:225-254: economic QR
// with the sign convention diag(R) > 0, which is exactly the R of the Cholesky factorisation of
// phi^H phi).  Two passes of
//     S = X^H X  (GEMM)      S = R^H R,  T = R^-1  (register-resident kernel)      X <- X T  (GEMM)
// with X = phi_s then Q_1; det R = prod sqrt(D) over both passes.  One pass loses orthogonality
// like eps cond(phi)^2, the second pass restores it to eps as long as cond(phi) < ~1e7; a
// non-positive pivot flags the walker and the Gram-Schmidt kernel (k_small.hip) redoes it.
struct GramProb {
    static constexpr bool A_CPLX = true, B_CPLX = true, A_CONJ = true;
    int batch, rows, cols, kdim;     // 2 nw, nmax, nmax, M
    int nt, na, nb, ld;
    const cplx *x;                   // [nw, M, nt]
    cplx *S;                         // [2 nw, ld * ld]
    const cplx *zero;
    __device__ bool active(int) const { return true; }
    __device__ cplx loadA(int, int, int) const { return cmake(0, 0); }
    __device__ cplx loadB(int, int, int) const { return cmake(0, 0); }
    __device__ const cplx *ptrA(int b, int row, int k) const {
        const int s = b & 1, ns = s ? nb : na;
        return row < ns ? x + ((long)(b >> 1) * kdim + k) * nt + (s ? na : 0) + row : zero;
    }
    __device__ const cplx *ptrB(int b, int k, int col) const {
        const int s = b & 1, ns = s ? nb : na;
        return col < ns ? x + ((long)(b >> 1) * kdim + k) * nt + (s ? na : 0) + col : zero;
    }
    static constexpr bool INCR = true;       // incremental refill of the ring engine
    __device__ int klimit(int) const { return kdim; }
    __device__ const cplx *baseA(int b, int row) const { return x + (long)(b >> 1) * kdim * nt + ((b & 1) ? na : 0) + row; }
    __device__ const cplx *baseB(int b, int col) const { return x + (long)(b >> 1) * kdim * nt + ((b & 1) ? na : 0) + col; }
    __device__ long kstepA() const { return nt; }
    __device__ long kstepB(int) const { return nt; }
    __device__ bool rowok(int b, int row) const { return row < ((b & 1) ? nb : na); }
    __device__ bool colok(int b, int col) const { return col < ((b & 1) ? nb : na); }
    __device__ void store(int b, int row, int col, double re, double im) const {
        S[(long)b * ld * ld + (long)row * ld + col] = cmake(re, im);
    }
};

struct QProb {
    static constexpr bool A_CPLX = true, B_CPLX = true;
    int batch, rows, cols, kdim;     // 2 nw, M, nmax, nmax
    int nt, na, nb, ld;
    const cplx *x;                   // [nw, M, nt]
    const cplx *Tt;                  // [2 nw, ld * ld]: Tt[c][j] = T[j][c]
    cplx *out;                       // [nw, M, nt]
    const int *fail;                 // walkers flagged by the Cholesky kernel are left alone
    const cplx *zero;
    __device__ bool active(int b) const { return fail[b >> 1] == 0; }
    __device__ cplx loadA(int, int, int) const { return cmake(0, 0); }
    __device__ cplx loadB(int, int, int) const { return cmake(0, 0); }
    __device__ const cplx *ptrA(int b, int row, int k) const {
        const int s = b & 1, ns = s ? nb : na;
        return k < ns ? x + ((long)(b >> 1) * rows + row) * nt + (s ? na : 0) + k : zero;
    }
    __device__ const cplx *ptrB(int b, int k, int col) const {
        const int ns = (b & 1) ? nb : na;
        return (k < ns && col < ns) ? Tt + (long)b * ld * ld + (long)col * ld + k : zero;
    }
    static constexpr bool INCR = true;       // incremental refill of the ring engine
    __device__ int klimit(int b) const { return (b & 1) ? nb : na; }
    __device__ const cplx *baseA(int b, int row) const { return x + ((long)(b >> 1) * rows + row) * nt + ((b & 1) ? na : 0); }
    __device__ const cplx *baseB(int b, int col) const { return Tt + (long)b * ld * ld + (long)col * ld; }
    __device__ long kstepA() const { return 1; }
    __device__ long kstepB(int) const { return 1; }
    __device__ bool rowok(int, int) const { return true; }
    __device__ bool colok(int b, int col) const { return col < ((b & 1) ? nb : na); }
    __device__ void store(int b, int row, int col, double re, double im) const {
        const int s = b & 1, ns = s ? nb : na;
        if (col < ns) out[((long)(b >> 1) * rows + row) * nt + (s ? na : 0) + col] = cmake(re, im);
    }
};

struct CholArgs {
    int na, nb, ld;
    const cplx *S;                   // [2 nw, ld * ld] Hermitian positive definite
    cplx *Tt;                        // [2 nw, ld * ld] transposed inverse Cholesky factor
    double *logd;                    // [2 nw] log det R of this pass
    int *fail;                       // [nw]
};

// Forward elimination of [S | I] in place without pivoting (same thread <-> element map and
// register residency as gj_big_kernel; one barrier per step because the pivot is known):
// after step k the slots (i > k, k) hold -m_i = column k of the unit-lower inverse factor, so at the
// end v[i][j] (j < i) = Ltilde^-1[i][j] with S = Ltilde D Ltilde^H, and
//     T = R^-1 = Ltilde^-H D^-1/2,   Tt[i][j] = T[j][i] = conj(v[i][j]) / sqrt(D_i).
#ifdef AFQ_TUNING      // (tuning builds only: AFQ_CHOL_STEPWISE; the product runs chol_mfma_kernel)
__global__ __launch_bounds__(512) void chol_linv_kernel(CholArgs a) {
    __shared__ cplx colk[2][GJ_N], rowk[2][GJ_N];
    __shared__ double piv[GJ_N];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = (b & 1) ? a.nb : a.na;
    const cplx *S = a.S + (long)b * a.ld * a.ld;
    cplx *Tt = a.Tt + (long)b * a.ld * a.ld;
    const int tr = tid >> 5, tc = tid & 31;
    double vr[8][4], vi[8][4];
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int i = tr + 16 * x, j = tc + 32 * y;
            const cplx t = (i < n && j < n) ? S[(long)i * a.ld + j] : cmake(0.0, 0.0);
            vr[x][y] = t.x; vi[x][y] = t.y;
        }
    if (tid < GJ_N) piv[tid] = 1.0;
    bool bad = false;
#pragma unroll
    for (int y0 = 0; y0 < 4; ++y0) {
        for (int kk = 0; kk < 32; ++kk) {
            const int k = 32 * y0 + kk;
            if (k >= n) break;
            const int buf = k & 1;
            if (tc == kk) {
#pragma unroll
                for (int x = 0; x < 8; ++x) colk[buf][tr + 16 * x] = cmake(vr[x][y0], vi[x][y0]);
            }
            if (wave == ((k & 15) >> 1)) {
                const int x0 = k >> 4;
                const bool mine = tr == (k & 15);
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    if (x0 == x) {
                        if (mine) {
#pragma unroll
                            for (int y = 0; y < 4; ++y) {
                                const int j = tc + 32 * y;
                                rowk[buf][j] = (j == k) ? cmake(1.0, 0.0) : cmake(vr[x][y], vi[x][y]);
                            }
                        }
                    }
                }
            }
            __syncthreads();
            const double d = colk[buf][k].x;
            bad = bad || !(d > 0.0);
            const double dinv = 1.0 / d;
            if (tid == 0) piv[k] = d;
            double rkx[4], rky[4];
#pragma unroll
            for (int y = 0; y < 4; ++y) { const cplx t = rowk[buf][tc + 32 * y]; rkx[y] = t.x; rky[y] = t.y; }
            if (tc == kk) {
#pragma unroll
                for (int x = 0; x < 8; ++x) { vr[x][y0] = 0.0; vi[x][y0] = 0.0; }
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int i = tr + 16 * x;
                const cplx c = colk[buf][i];
                const double fx = i > k ? c.x * dinv : 0.0, fy = i > k ? c.y * dinv : 0.0;
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    double ox = vr[x][y], oy = vi[x][y];
                    ox = fma(-fx, rkx[y], ox); ox = fma(fy, rky[y], ox);
                    oy = fma(-fx, rky[y], oy); oy = fma(-fy, rkx[y], oy);
                    vr[x][y] = ox; vi[x][y] = oy;
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        const int i = tr + 16 * x;
        if (i >= n) continue;
        const double rs = 1.0 / sqrt(piv[i]);
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int j = tc + 32 * y;
            if (j >= n) continue;
            cplx t = cmake(0.0, 0.0);
            if (j < i) t = cmake(vr[x][y] * rs, -vi[x][y] * rs);
            else if (j == i) t = cmake(rs, 0.0);
            Tt[(long)i * a.ld + j] = t;
        }
    }
    if (wave == 0) {
        double l = log(piv[lane]) + log(piv[lane + 64]);
        for (int o = 32; o > 0; o >>= 1) l += __shfl_down(l, o);
        if (lane == 0) {
            a.logd[b] = 0.5 * l;
            if (bad) a.fail[b >> 1] = 1;
        }
    }
}
#endif

// Blocked version of chol_linv_kernel on the matrix pipe (round 4; the structure of gj_mfma_kernel: eight waves per matrix,
// 16 x 16 tiles in MFMA accumulator layout, wave w > 0 owns the tiles (I, (I + w) mod 8), the pivot wave 0 keeps the
// diagonal tiles in LDS and works one block step ahead).  Block forward elimination of [S | I] with S = L L^H:
//   block step k:  C = L_kk^-1 (inverse Cholesky factor of the pivot tile, one wave in registers);
//                  Y_j = C x (block row k, tile j)    -- the remaining S for j > k, the accumulated right-hand side W for
//                                                        j < k, the identity for j = k (Y_k = C);  X_kj = Y_j (j <= k) is
//                                                        row block k of L^-1, final;
//                  tile (i, j) -= Y_i^H Y_j  for i > k and (j <= k: right-hand side | j >= i: upper triangle of S)
// S is Hermitian, so the block column is the conjugate transpose of the block row and never read: the accumulator layout
// of Y_i IS the A-fragment order of Y_i^H (conjugated on the way), both operands of the update come from the one published
// panel.  About half the tile products of the Gauss-Jordan.  No pivoting (positive definite); a non-positive pivot sets the
// walker's breakdown flag as in chol_linv_kernel.  Output: Tt = conj(L^-1) (zeros above the diagonal), logd = log det L.
struct CholMfmaLds {
    static constexpr int RP = 0, DG = 32768, PL = 65536, PT = 69632, CA = 73728, DALL = 81920, SMALL = 82944;
    static constexpr int BYTES = SMALL + 512 + 256 + 64;                        // + rowk, piv of one tile, the flag
};

// acc += A B with A = -Y_i^H, B = Y_j: fragments of both from panels in accumulator layout (register ks = k-step ks);
// complex by three real MFMAs per k-step
__device__ __attribute__((always_inline)) inline void chol_tile_mac(const d2_t *yi, const d2_t *yj, int lane, int nks, d4_t &accr, d4_t &acci) {
    d4_t p1 = (d4_t){0, 0, 0, 0}, p2 = (d4_t){0, 0, 0, 0};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        d2_t av[2], bv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { av[q] = yi[(2 * half + q) * 64 + lane]; bv[q] = yj[(2 * half + q) * 64 + lane]; }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (2 * half + q < nks) {
                const double ar = -av[q][0], ai = av[q][1];            // -conj(y)
                p1 = mfma16(ar, bv[q][0], p1);
                p2 = mfma16(ai, bv[q][1], p2);
                acci = mfma16(ar + ai, bv[q][0] + bv[q][1], acci);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { accr[r] += p1[r] - p2[r]; acci[r] -= p1[r] + p2[r]; }
}

__global__ __launch_bounds__(512) void chol_mfma_kernel(CholArgs a) {
    extern __shared__ __align__(16) unsigned char gsm[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, lr = lane & 15;
    const int n = (b & 1) ? a.nb : a.na;
    const int nt16 = (n + 15) >> 4;
    const cplx *S = a.S + (long)b * a.ld * a.ld;
    cplx *Tt = a.Tt + (long)b * a.ld * a.ld;
    d2_t *Rp = (d2_t *)(gsm + CholMfmaLds::RP), *Dg = (d2_t *)(gsm + CholMfmaLds::DG);
    cplx *Pl = (cplx *)(gsm + CholMfmaLds::PL), *Pt = (cplx *)(gsm + CholMfmaLds::PT);
    d2_t *Ca = (d2_t *)(gsm + CholMfmaLds::CA);                        // [2][4 k-steps][64]: C in A-fragment order
    double *dall = (double *)(gsm + CholMfmaLds::DALL);                // [128] pivots D_k (1 beyond n)
    cplx *rowk = (cplx *)(gsm + CholMfmaLds::SMALL);
    double *piv = (double *)(rowk + 32);
    // Tt rows / columns of tile (I, J), conjugated (buffer stores: one 32-bit lane offset, see gj_mfma_kernel)
    const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc(Tt, 0, a.ld * a.ld * (int)sizeof(cplx), 0x00020000);
    const int lane_off = (lk * a.ld + lr) * (int)sizeof(cplx);
    auto store_conj = [&](int I, int J, const d4_t &tr, const d4_t &ti) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + 4 * r + lk, col = 16 * J + lr;
            const int uo = ((16 * I + 4 * r) * a.ld + 16 * J) * (int)sizeof(cplx);
            if (row < n && col < n)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gj_u4_t, (d2_t){tr[r], -ti[r]}), trsrc, lane_off, uo, 0);
        }
    };
    if (wave == 0) {
        // ================================================================= pivot wave
        __builtin_amdgcn_s_setprio(3);
        bool bad = false;
        // pivot tile t (row-major in Pl, compact nb x nb) -> C = L_tt^-1: A-fragment order in Ca[t & 1], accumulator layout
        // in Dg[t] (its final value: X_tt), conjugated to Tt; pivots to dall
        auto factor = [&](const int t, const int nb) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            chol_wave16q(Pl, nb, Pt, nb, nb, lane, rowk, piv, bad);          // Pt = conj(C), zeros above the diagonal
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < 16) dall[16 * t + lane] = lane < nb ? piv[lane] : 1.0;
            d2_t *ca = Ca + (t & 1) * 256;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = lane + 64 * q, row = e >> 4, col = e & 15;
                const cplx v = (row < nb && col <= row) ? Pt[row * nb + col] : cmake(0.0, 0.0);
                ca[(col >> 2) * 64 + (col & 3) * 16 + row] = (d2_t){v.x, -v.y};
                Dg[(t * 4 + (row >> 2)) * 64 + (row & 3) * 16 + col] = (d2_t){v.x, -v.y};
                const int gr = 16 * t + row, gc = 16 * t + col;
                if (gr < n && gc < n) Tt[(long)gr * a.ld + gc] = v;
            }
        };
        for (int e = lane; e < 128; e += 64) dall[e] = 1.0;
        {
            cplx dt[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * i + 4 * r + lk, col = 16 * i + lr;
                    dt[i][r] = (row < n && col < n) ? S[(long)row * a.ld + col] : cmake(0.0, 0.0);
                }
            const int nb0 = n < 16 ? n : 16;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * r + lk < nb0 && lr < nb0) Pl[(4 * r + lk) * nb0 + lr] = dt[0][r];
            factor(0, nb0);
#pragma unroll
            for (int i = 1; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) Dg[(i * 4 + r) * 64 + lane] = (d2_t){dt[i][r].x, dt[i][r].y};
        }
        gj_lds_barrier();                                              // P
        for (int kb = 0; kb < nt16; ++kb) {
            const int nblk = n - 16 * kb < 16 ? n - 16 * kb : 16;
            const int nks = (nblk + 3) >> 2;
            gj_lds_barrier();                                          // B1
            const int nx = kb + 1;
            if (nx < nt16) {
                const int nbx = n - 16 * nx < 16 ? n - 16 * nx : 16;
                d4_t accr, acci;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const d2_t v = Dg[(nx * 4 + r) * 64 + lane]; accr[r] = v[0]; acci[r] = v[1]; }
                chol_tile_mac(Rp + nx * 256, Rp + nx * 256, lane, nks, accr, acci);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * r + lk < nbx && lr < nbx) Pl[(4 * r + lk) * nbx + lr] = cmake(accr[r], acci[r]);
                factor(nx, nbx);
            }
            gj_lds_barrier();                                          // B4
        }
        // log det L = 1/2 sum log D_k; breakdown flag
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        double l = log(dall[lane]) + log(dall[lane + 64]);
        bad = bad || !(dall[lane] > 0.0) || !(dall[lane + 64] > 0.0);
        for (int o = 32; o > 0; o >>= 1) l += __shfl_down(l, o);
        const bool anybad = __ballot(bad) != 0ull;
        if (lane == 0) {
            a.logd[b] = 0.5 * l;
            if (anybad) a.fail[b >> 1] = 1;
        }
        return;
    }
    // ===================================================================== tile waves 1..7: tile i is (I = i, J = (i + wave) & 7)
    d4_t Cr[8], Ci[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int J = (i + wave) & 7;
        Cr[i] = (d4_t){0, 0, 0, 0}; Ci[i] = (d4_t){0, 0, 0, 0};
        if (J > i) {                                                   // upper triangle of S; Tt is zero there
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * i + 4 * r + lk, col = 16 * J + lr;
                if (row < n && col < n) {
                    const cplx t = S[(long)row * a.ld + col];
                    Cr[i][r] = t.x; Ci[i][r] = t.y;
                    Tt[(long)row * a.ld + col] = cmake(0.0, 0.0);
                }
            }
        }
    }
    gj_lds_barrier();                                                  // P
    for (int kb = 0; kb < nt16; ++kb) {
        const int nblk = n - 16 * kb < 16 ? n - 16 * kb : 16;
        const int nks = (nblk + 3) >> 2;
        const int Jk = (kb + wave) & 7;                                // this wave's tile of block row kb is (kb, Jk)
        // ---- Y = C x tile of the block row, from the registers -> Rp; a right-hand-side tile (Jk < kb) is final: row block
        //      kb of L^-1
        if (Jk < nt16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i == kb) {                                         // wave-uniform; static register index
                    const d2_t *ca = Ca + (kb & 1) * 256;
                    d2_t av[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) av[ks] = ca[ks * 64 + lane];
                    d4_t Nr = (d4_t){0, 0, 0, 0}, Ni = (d4_t){0, 0, 0, 0}, p2 = (d4_t){0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        if (ks < nks) {
                            Nr = mfma16(av[ks][0], Cr[i][ks], Nr);
                            p2 = mfma16(av[ks][1], Ci[i][ks], p2);
                            Ni = mfma16(av[ks][0] + av[ks][1], Cr[i][ks] + Ci[i][ks], Ni);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { Ni[r] -= Nr[r] + p2[r]; Nr[r] -= p2[r]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) Rp[(Jk * 4 + r) * 64 + lane] = (d2_t){Nr[r], Ni[r]};
                    if (Jk < kb) store_conj(kb, Jk, Nr, Ni);
                }
            }
        }
        gj_lds_barrier();                                              // B1
        // ---- tile (i, J) -= Y_i^H Y_J for i > kb, J <= kb (right-hand side) or J > i (upper triangle of S)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int J = (i + wave) & 7;
            if (i > kb && i < nt16 && J < nt16 && (J <= kb || J > i)) {
                const d2_t *yj = J == kb ? Dg + kb * 256 : Rp + J * 256;    // (Y_kb = C itself: the identity block of the right-hand side)
                chol_tile_mac(Rp + i * 256, yj, lane, nks, Cr[i], Ci[i]);
            }
        }
        // ... and the diagonal tiles below the next pivot (that one is wave 0's), one each for waves 1, 2, 3, 5, 6, 7
        for (int i = kb + 2; i < nt16; ++i) {
            const int rank = i - kb - 2;
            const int owner = rank < 3 ? rank + 1 : rank + 2;
            if (wave != owner) continue;
            d4_t accr, acci;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const d2_t v = Dg[(i * 4 + r) * 64 + lane]; accr[r] = v[0]; acci[r] = v[1]; }
            chol_tile_mac(Rp + i * 256, Rp + i * 256, lane, nks, accr, acci);
#pragma unroll
            for (int r = 0; r < 4; ++r) Dg[(i * 4 + r) * 64 + lane] = (d2_t){accr[r], acci[r]};
        }
        gj_lds_barrier();                                              // B4
    }
}

// n <= 32: one wave per matrix, four matrices per work-group, everything in registers (gj_wave.h)
__global__ __launch_bounds__(256) void chol_small_kernel(CholArgs a, int nmat) {
    __shared__ cplx rowk_s[4][32];
    __shared__ double piv_s[4][32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= nmat) return;
    const int n = __builtin_amdgcn_readfirstlane((b & 1) ? a.nb : a.na);
    const cplx *S = a.S + (long)b * a.ld * a.ld;
    cplx *Tt = a.Tt + (long)b * a.ld * a.ld;
    cplx *rowk = rowk_s[wave];
    double *piv = piv_s[wave];
    bool bad = false;
    chol_wave32(S, a.ld, Tt, a.ld, n, lane, rowk, piv, bad);
    double l = lane < 32 ? log(piv[lane]) : 0.0;
    for (int o = 16; o > 0; o >>= 1) l += __shfl_down(l, o);
    if (lane == 0) {
        a.logd[b] = 0.5 * l;
        if (bad) a.fail[b >> 1] = 1;
    }
}

__global__ void qr_finish_kernel(const double *logd, const int *fail, double *detR, cplx *ot, double *weight,
                                 int nw, int free_projection) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw || fail[w]) return;
    const double d = exp(logd[2 * w] + logd[2 * w + 1] + logd[2 * nw + 2 * w] + logd[2 * nw + 2 * w + 1]);
    detR[w] = d;
    ot[w] = cmake(ot[w].x / d, ot[w].y / d);             // single_det.py:253
    if (free_projection) weight[w] *= d;                   // walkers/handler.py:178-181
}

int k_reortho_big(afq_handle *h) {
    const int nmax = h->na > h->nb ? h->na : h->nb;
    const int nb2 = 2 * h->nw;
    const size_t wsn = (size_t)nb2 * nmax * nmax;
    if (!h->big_ws) AFQ_HIP(h, hipMalloc(&h->big_ws, sizeof(cplx) * wsn));
    if (!h->big_ws2) AFQ_HIP(h, hipMalloc(&h->big_ws2, sizeof(cplx) * wsn));
    if (!h->qr_logd) {
        AFQ_HIP(h, hipMalloc(&h->qr_logd, sizeof(double) * 2 * nb2));
        AFQ_HIP(h, hipMalloc(&h->qr_fail, sizeof(int) * h->nw));
    }
    AFQ_HIP(h, hipMemsetAsync(h->qr_fail, 0, sizeof(int) * h->nw, h->stream));
    for (int pass = 0; pass < 2; ++pass) {
        const cplx *src = pass == 0 ? h->phi : h->phi_t;
        cplx *dst = pass == 0 ? h->phi_t : h->phi;
        {
            GramProb p;
            p.batch = nb2; p.rows = nmax; p.cols = nmax; p.kdim = h->M;
            p.nt = h->nt; p.na = h->na; p.nb = h->nb; p.ld = nmax;
            p.x = src; p.S = h->big_ws; p.zero = (const cplx *)h->zero_page;
#ifdef AFQ_TUNING
            if (AFQ_KNOB_SET("AFQ_BIG_LEAN")) AFQ_GEMM_AS(h, "k_reortho_big: GramProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GramProb, MAP_COLS_FAST, true, 1, 5, 4>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_BIG_WPE")) AFQ_GEMM_AS(h, "k_reortho_big: GramProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GramProb, MAP_COLS_FAST, true, 1, 3, 4>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_BIG_NOLOADER")) AFQ_GEMM_AS(h, "k_reortho_big: GramProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GramProb, MAP_COLS_FAST, true>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_BIG_NOLEAN")) AFQ_GEMM_AS(h, "k_reortho_big: GramProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GramProb, MAP_COLS_FAST, true, 1, 3>(p, h->stream, h->zero_page)));
            else
#endif
            // round 5: the lean loop at two work-groups per CU (mfma_gemm_wg.h, STAG = 5): 238 -> 218 us at C4.  (The overlap
            // and Ghalf GEMMs of the Green's function measured 2-3 % SLOWER that way and keep the pipelined loop.)
            AFQ_GEMM_AS(h, "k_reortho_big: GramProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GramProb, MAP_COLS_FAST, true, 1, 5, 4>(p, h->stream, h->zero_page)));
        }
        {
            CholArgs a;
            a.na = h->na; a.nb = h->nb; a.ld = nmax;
            a.S = h->big_ws; a.Tt = h->big_ws2; a.logd = h->qr_logd + (size_t)pass * nb2; a.fail = h->qr_fail;
            if (nmax <= 32) AFQ_LAUNCH(h, chol_small_kernel, dim3((nb2 + 3) / 4), dim3(256), 0, h->stream, a, nb2);
#ifdef AFQ_TUNING
            else if (AFQ_KNOB_SET("AFQ_CHOL_STEPWISE")) AFQ_LAUNCH(h, chol_linv_kernel, dim3(nb2), dim3(512), 0, h->stream, a);
#endif
            else {
                static size_t lds_set[AFQ_MAX_DEVICES] = {0};
                AFQ_HIP(h, afq_raise_lds((const void *)chol_mfma_kernel, CholMfmaLds::BYTES, lds_set));
                AFQ_LAUNCH(h, chol_mfma_kernel, dim3(nb2), dim3(512), CholMfmaLds::BYTES, h->stream, a);
            }
            AFQ_POST(h);
        }
        {
            QProb p;
            p.batch = nb2; p.rows = h->M; p.cols = nmax; p.kdim = nmax;
            p.nt = h->nt; p.na = h->na; p.nb = h->nb; p.ld = nmax;
            p.x = src; p.Tt = h->big_ws2; p.out = dst; p.fail = h->qr_fail; p.zero = (const cplx *)h->zero_page;
#ifdef AFQ_TUNING
            if (AFQ_KNOB_SET("AFQ_BIG_LEAN")) AFQ_GEMM_AS(h, "k_reortho_big: QProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, QProb, MAP_COLS_FAST, true, 1, 5, 4>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_BIG_WPE")) AFQ_GEMM_AS(h, "k_reortho_big: QProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, QProb, MAP_COLS_FAST, true, 1, 3, 4>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_BIG_NOLOADER")) AFQ_GEMM_AS(h, "k_reortho_big: QProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, QProb, MAP_COLS_FAST, true>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_BIG_NOLEAN")) AFQ_GEMM_AS(h, "k_reortho_big: QProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, QProb, MAP_COLS_FAST, true, 1, 3>(p, h->stream, h->zero_page)));
            else
#endif
            AFQ_GEMM_AS(h, "k_reortho_big: QProb GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, QProb, MAP_COLS_FAST, true, 1, 5, 4>(p, h->stream, h->zero_page)));   // 253 -> 230 us
        }
    }
    AFQ_LAUNCH(h, qr_finish_kernel, dim3((h->nw + 127) / 128), dim3(128), 0, h->stream, h->qr_logd,
                       h->qr_fail, h->detR, h->ot, h->weight, h->nw, (h->flags & AFQ_PROP_FREE_PROJECTION) ? 1 : 0);
    AFQ_POST(h);
    return AFQ_OK;
}
