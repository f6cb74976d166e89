// Half-rotated Cholesky local energy (estimators/generic.py:156-221), all
// walkers at once.
//
//   e1b   = sum_s sum_{pq} H1_s[p,q] G_s[p,q] = sum_{s,i,q} rH1[i,q] Ghalf_s[i,q]
//   X_s   = rchol_s^T vec(Ghalf_s)                (the force-bias contraction)
//   ecoul = (X_a + X_b).(X_a + X_b)
//   T_s[x,i,j] = sum_p rchol_s[(i,p),x] Ghalf_s[j,p]
//   exx   = sum_s sum_{x,i,j} T_s[x,i,j] T_s[x,j,i]
//
// The exchange term is the dense contraction (2 K N^2 M real x complex MACs per
// walker) and runs on fp64 MFMA with the Cholesky index x on the tile rows and
// the WALKER index on the tile columns: a 16x16 tile is (16 x) x (16 walkers)
// for one fixed orbital pair (i,j).  With that choice
//   * every tile is full (no padding of N=25 to 32),
//   * T[x,i,j] and T[x,j,i] of the same (x, walker) sit in the same lane and
//     register of two accumulators, so the trace T_ij T_ji is a lane-local
//     multiply: the K x N x N intermediate never leaves registers.
// Operands are stored in MFMA fragment order so every fragment load is one
// fully coalesced 512-byte wavefront load:
//   afrag[s][i][xt][ks][lane] = rchol_s[(i, p = 4 ks + lane/16), x = 16 xt + lane%16]   (built once)
//   gfrag[orb][c][wt][ks][lane] = (re|im) Ghalf[w = 16 wt + lane%16][orb][p = 4 ks + lane/16]
#include <cstdlib>
#include "mfma_gemm_wg.h"

#define EXX_CHUNKS 2      // wave-tasks per (spin, x-tile, walker-tile) cell

struct ExxArgs {
    int M, K, nw, nt, nks, nxt, nwt;
    int ns[2], goff[2];
    const double *afrag[2];
    const double *afrag_im[2];
    const double *gfrag;
    cplx *part;               // [2, nxt, nwt, EXX_CHUNKS, 16]
};

__device__ inline void cmul_acc(double &sr, double &si, const d4_t &ar, const d4_t &ai, const d4_t &br,
                                const d4_t &bi, double f) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sr += f * (ar[r] * br[r] - ai[r] * bi[r]);
        si += f * (ar[r] * bi[r] + ai[r] * br[r]);
    }
}

// One k-step worth of operand fragments of a 2x2 orbital block pair.
template <bool RC>
struct ExxFrags {
    double Air[2], Ajr[2], Aii[2], Aji[2], Bir[2], Bii[2], Bjr[2], Bji[2];
};

template <bool RC>
__global__ __launch_bounds__(256, 2) void exx_kernel(ExxArgs a) {
    const int lane = threadIdx.x & 63;
    // XCD affinity: workgroup g runs on XCD g % 8 (round-robin dispatch); give each
    // XCD the walker tiles wt = xcd, xcd + 8, ... so that the Ghalf fragments it
    // reads (640 KB per walker tile and spin at C3) stay resident in its 4 MiB L2
    // while the rchol fragments stream through.
    const int xcd = blockIdx.x & 7;
    const int nj = (a.nwt + 7) >> 3;
    const long t = (long)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    const int chunk = (int)(t % EXX_CHUNKS);
    const int j = (int)((t / EXX_CHUNKS) % nj);
    const int xt = (int)((t / ((long)EXX_CHUNKS * nj)) % a.nxt);
    const int s = (int)(t / ((long)EXX_CHUNKS * nj * a.nxt));
    const int wt = xcd + 8 * j;
    if (s >= 2 || wt >= a.nwt) return;
    const long task = (((long)s * a.nxt + xt) * a.nwt + wt) * EXX_CHUNKS + chunk;
    const int ns = a.ns[s];
    double sr = 0.0, si = 0.0;
    if (ns > 0) {
        const int nblk = (ns + 1) / 2;
        const int npairs = nblk * (nblk + 1) / 2;
        const int p_lo = (int)((long)npairs * chunk / EXX_CHUNKS);
        const int p_hi = (int)((long)npairs * (chunk + 1) / EXX_CHUNKS);
        const long astride_i = (long)a.nxt * a.nks * 64;      // doubles between orbitals in afrag
        const long gstride_o = 2L * a.nwt * a.nks * 64;       // doubles between orbitals in gfrag
        const long gstride_c = (long)a.nwt * a.nks * 64;
        const double *A0 = a.afrag[s] + (long)xt * a.nks * 64 + lane;
        const double *A0i = RC ? a.afrag_im[s] + (long)xt * a.nks * 64 + lane : nullptr;
        const double *G0 = a.gfrag + (long)a.goff[s] * gstride_o + (long)wt * a.nks * 64 + lane;
        int pidx = 0;
        for (int ib = 0; ib < nblk; ++ib) {
            for (int jb = ib; jb < nblk; ++jb, ++pidx) {
                if (pidx < p_lo || pidx >= p_hi) continue;
                const int i0 = 2 * ib, j0 = 2 * jb;
                const int ci = (i0 + 1 < ns) ? 2 : 1, cj = (j0 + 1 < ns) ? 2 : 1;
                // wave-uniform strip offsets (scalar registers); an orbital past the end is clamped onto a
                // valid one and its tiles get weight 0 in the trace (only the last block of an odd
                // electron count pays for that)
                long oAi[2], oAj[2], oBi[2], oBj[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const int oi = x < ci ? i0 + x : i0, oj = x < cj ? j0 + x : j0;
                    oAi[x] = (long)oi * astride_i; oAj[x] = (long)oj * astride_i;
                    oBi[x] = (long)oi * gstride_o; oBj[x] = (long)oj * gstride_o;
                }
                if (ib == jb) {
                    // tiles T[i_x, i_y]; fragments of k-step ks+1 in flight while the MFMAs of ks issue
                    d4_t Tr[2][2], Ti[2][2];
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int y = 0; y < 2; ++y) { Tr[x][y] = (d4_t){0, 0, 0, 0}; Ti[x][y] = (d4_t){0, 0, 0, 0}; }
                    // Two fragment sets, loop unrolled by two with scheduling fences: written as cur/nxt with a copy
                    // at the end of the iteration the compiler folds the copy away and ends up loading, waiting for and
                    // consuming the fragments of one k-step inside the same iteration (no prefetch at all).
                    struct DFr { double A[2], Ai[2], Br[2], Bi[2]; };
                    DFr fA, fB;
                    auto dload = [&](DFr &f, int o) __attribute__((always_inline)) {
#pragma unroll
                        for (int x = 0; x < 2; ++x) {
                            f.A[x] = A0[oAi[x] + o]; f.Br[x] = G0[oBi[x] + o]; f.Bi[x] = G0[oBi[x] + gstride_c + o];
                            f.Ai[x] = RC ? A0i[oAi[x] + o] : 0.0;
                        }
                    };
                    auto dmfma = [&](const DFr &f) __attribute__((always_inline)) {
#pragma unroll
                        for (int x = 0; x < 2; ++x)
#pragma unroll
                            for (int y = 0; y < 2; ++y) {
                                Tr[x][y] = mfma16(f.A[x], f.Br[y], Tr[x][y]);
                                Ti[x][y] = mfma16(f.A[x], f.Bi[y], Ti[x][y]);
                                if (RC) {
                                    Tr[x][y] = mfma16(-f.Ai[x], f.Bi[y], Tr[x][y]);
                                    Ti[x][y] = mfma16(f.Ai[x], f.Br[y], Ti[x][y]);
                                }
                            }
                    };
                    dload(fA, 0);
                    for (int ks = 0; ks < a.nks; ks += 2) {
                        dload(fB, (ks + 1 < a.nks ? ks + 1 : ks) * 64);
                        __builtin_amdgcn_sched_barrier(0);
                        dmfma(fA);
                        __builtin_amdgcn_sched_barrier(0);
                        if (ks + 1 < a.nks) {
                            dload(fA, (ks + 2 < a.nks ? ks + 2 : ks + 1) * 64);
                            __builtin_amdgcn_sched_barrier(0);
                            dmfma(fB);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    cmul_acc(sr, si, Tr[0][0], Ti[0][0], Tr[0][0], Ti[0][0], 1.0);
                    if (ci == 2) {
                        cmul_acc(sr, si, Tr[1][1], Ti[1][1], Tr[1][1], Ti[1][1], 1.0);
                        cmul_acc(sr, si, Tr[0][1], Ti[0][1], Tr[1][0], Ti[1][0], 2.0);
                    }
                } else {
                    // T[x][y] = tile(i_x, j_y), S[y][x] = tile(j_y, i_x)
                    d4_t Tr[2][2], Ti[2][2], Sr[2][2], Si[2][2];
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int y = 0; y < 2; ++y) {
                            Tr[x][y] = (d4_t){0, 0, 0, 0}; Ti[x][y] = (d4_t){0, 0, 0, 0};
                            Sr[x][y] = (d4_t){0, 0, 0, 0}; Si[x][y] = (d4_t){0, 0, 0, 0};
                        }
                    ExxFrags<RC> fA, fB;                   // two sets, unrolled by two: see the diagonal case
                    auto load = [&](ExxFrags<RC> &f, int o) __attribute__((always_inline)) {
#pragma unroll
                        for (int x = 0; x < 2; ++x) {
                            f.Air[x] = A0[oAi[x] + o]; f.Ajr[x] = A0[oAj[x] + o];
                            f.Bir[x] = G0[oBi[x] + o]; f.Bii[x] = G0[oBi[x] + gstride_c + o];
                            f.Bjr[x] = G0[oBj[x] + o]; f.Bji[x] = G0[oBj[x] + gstride_c + o];
                            if (RC) { f.Aii[x] = A0i[oAi[x] + o]; f.Aji[x] = A0i[oAj[x] + o]; }
                        }
                    };
                    auto mfmas = [&](const ExxFrags<RC> &cur) __attribute__((always_inline)) {
#pragma unroll
                        for (int x = 0; x < 2; ++x)
#pragma unroll
                            for (int y = 0; y < 2; ++y) {
                                Tr[x][y] = mfma16(cur.Air[x], cur.Bjr[y], Tr[x][y]);
                                Ti[x][y] = mfma16(cur.Air[x], cur.Bji[y], Ti[x][y]);
                                Sr[y][x] = mfma16(cur.Ajr[y], cur.Bir[x], Sr[y][x]);
                                Si[y][x] = mfma16(cur.Ajr[y], cur.Bii[x], Si[y][x]);
                                if (RC) {
                                    Tr[x][y] = mfma16(-cur.Aii[x], cur.Bji[y], Tr[x][y]);
                                    Ti[x][y] = mfma16(cur.Aii[x], cur.Bjr[y], Ti[x][y]);
                                    Sr[y][x] = mfma16(-cur.Aji[y], cur.Bii[x], Sr[y][x]);
                                    Si[y][x] = mfma16(cur.Aji[y], cur.Bir[x], Si[y][x]);
                                }
                            }
                    };
                    load(fA, 0);
                    for (int ks = 0; ks < a.nks; ks += 2) {
                        load(fB, (ks + 1 < a.nks ? ks + 1 : ks) * 64);
                        __builtin_amdgcn_sched_barrier(0);
                        mfmas(fA);
                        __builtin_amdgcn_sched_barrier(0);
                        if (ks + 1 < a.nks) {
                            load(fA, (ks + 2 < a.nks ? ks + 2 : ks + 1) * 64);
                            __builtin_amdgcn_sched_barrier(0);
                            mfmas(fB);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int y = 0; y < 2; ++y)
                            if (x < ci && y < cj) cmul_acc(sr, si, Tr[x][y], Ti[x][y], Sr[y][x], Si[y][x], 2.0);
                }
            }
        }
    }
    // sum the 4 row groups of each walker column
    sr += __shfl_xor(sr, 16); si += __shfl_xor(si, 16);
    sr += __shfl_xor(sr, 32); si += __shfl_xor(si, 32);
    if (lane < 16) a.part[task * 16 + lane] = cmake(sr, si);
}

// --------------------------------------------------------------------------
// Exchange energy as a quadratic form (estimators/generic.py:198-216, re-associated):
//   exx_s = sum_x sum_ij T[x,i,j] T[x,j,i],   T[x,i,j] = sum_p R[(i,p),x] G[j,p]            (R = rchol_s, G = Ghalf_s)
//         = sum_{(j,p),(i,q)} G[j,p] Atil[(j,p),(i,q)] G[i,q],   Atil[(j,p),(i,q)] = sum_x R[(i,p),x] R[(j,q),x]
// Atil_s is a property of the trial: (N_s M)^2 entries (50 MB per spin at M=100, N=25; 3.2 GB at M=400, N=50 -- this
// is what 288 GB of HBM are for), built once on the device.  Per evaluation the exchange energy of every walker is
// then ONE real-by-complex GEMM  Y[w, :] = g_w^T Atil  ([nw x NM] x [NM x NM], 4 nw (NM)^2 flops per spin) followed
// by the dot products Y[w] . g_w in energy_finish_kernel: K / M times fewer flops than forming T (5x at K = 5 M),
// and no Cholesky-index loop at all.  Same number to rounding (different summation order, ~1e-15 relative).
// Used when K >= M and the matrices fit the memory budget; exx_kernel above otherwise.
#define EXQ_MAX_BATCH 32
template <bool RC>
struct ExxQProb {
    static constexpr bool A_CPLX = true, B_CPLX = RC;
    int batch, rows, cols, kdim;      // 2 spins x slices, nw, max N_s M, longest slice
    long astride;                     // elements between walkers in ghalf
    long goff[EXQ_MAX_BATCH];         // first contraction element of the batch inside a walker's ghalf
    int len[EXQ_MAX_BATCH];           // slice length
    int ncol[EXQ_MAX_BATCH];          // N_s M of the batch's spin
    const void *B[EXQ_MAX_BATCH];     // Atil_s + (slice start) * ldq
    long ldq[EXQ_MAX_BATCH];
    const cplx *ghalf;
    // the product g^T Atil never goes to memory: every 16-column tile of it is contracted with g again in the epilogue
    // (ROWDOT) and only those partial sums are stored, E[batch, nw, ncb] with ncb = ceil(cols / 16)
    static constexpr bool ROWDOT = true;
    long soff[EXQ_MAX_BATCH];         // first element of the batch's spin inside a walker's ghalf
    cplx *E;
    int ncb;
    const cplx *zero;
    // closed-shell populations (afq_internal.h: closed_bad): the launch that holds the beta spin's slices returns at once when
    // every walker's Ghalf_b equals its Ghalf_a -- the finish kernel then takes the alpha sums twice
    // (skip_from: the first batch that belongs to the beta spin -- 0 for a launch of its own, the alpha batch count when both
    //  spins' slices share one launch)
    const unsigned long long *skip_flag;
    unsigned long long skip_epoch;
    int skip_from;
    __device__ bool active(int b) const { return !(skip_flag && b >= skip_from && *skip_flag < skip_epoch); }
    __device__ const cplx *ptrA(int b, int row, int k) const {
        return k < len[b] ? ghalf + row * astride + goff[b] + k : zero;
    }
    __device__ const void *ptrB(int b, int k, int col) const {
        if (k >= len[b] || col >= ncol[b]) return (const void *)zero;
        return RC ? (const void *)((const cplx *)B[b] + k * ldq[b] + col) : (const void *)((const double *)B[b] + k * ldq[b] + col);
    }
    // Atil is stored as its upper triangle times two (diagonal once): rows beyond the last column of a tile are zero
    static constexpr bool KCUT = true;
    long k0[EXQ_MAX_BATCH];           // first row of Atil_s the slice contracts
    __device__ int kcut(int b, int col0, int ncols) const {
        const long need = (long)col0 + ncols - k0[b];
        return need <= 0 ? 0 : (need < len[b] ? (int)need : len[b]);
    }
    // incremental refill of the ring engine
    static constexpr bool INCR = true;
    using elemB = typename std::conditional<RC, cplx, double>::type;
    __device__ int klimit(int b) const { return len[b]; }
    __device__ const cplx *baseA(int b, int row) const { return ghalf + row * astride + goff[b]; }
    __device__ const elemB *baseB(int b, int col) const { return (const elemB *)B[b] + col; }
    __device__ long kstepA() const { return 1; }
    __device__ long kstepB(int b) const { return ldq[b]; }
    __device__ bool rowok(int, int) const { return true; }
    __device__ bool colok(int b, int col) const { return col < ncol[b]; }
    __device__ void store(int, int, int, double, double) const {}
    __device__ cplx dot_operand(int b, int row, int col) const {
        return col < ncol[b] ? ghalf[row * astride + soff[b] + col] : cmake(0.0, 0.0);
    }
    __device__ void store_dot(int b, int row, int tile, double re, double im) const {
        E[((long)b * rows + row) * ncb + tile] = cmake(re, im);
    }
};

// The beta-spin launch of a closed-shell try (launch_exx_quadratic): the same problem under its own name, so that a kernel
// trace separates the launches that return at once from the ones that multiply.
template <bool RC> struct ExxQBetaProb : ExxQProb<RC> {};

// A = R R^T (unconjugated) for one spin, stored permuted: out[(j,p),(i,q)] = A[(i,p),(j,q)].  One-time set-up
// kernel: 64 x 64 tiles, 16 contraction indices per LDS stage, 4 x 4 outputs per thread on the fp64 vector ALU.
template <bool RC>
__global__ __launch_bounds__(256) void atilde_build_kernel(const double *Rre, const double *Rim, long ld, int K, int ns,
                                                           int M, long row0, void *out_v, long ldo) {
    __shared__ double As[RC ? 2 : 1][16][68], Bs[RC ? 2 : 1][16][68];
    const int NM = ns * M;
    const int tr = blockIdx.y * 64, tc = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double ar[4][4], ai[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) { ar[u][v] = 0.0; ai[u][v] = 0.0; }
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int t = threadIdx.x; t < 64 * 16; t += 256) {
            const int r = t >> 4, kk = t & 15;
            const bool kok = k0 + kk < K;
            const bool aok = kok && tr + r < NM, bok = kok && tc + r < NM;
            const long ia = (row0 + tr + r) * ld + k0 + kk, ib = (row0 + tc + r) * ld + k0 + kk;
            As[0][kk][r] = aok ? Rre[ia] : 0.0;
            Bs[0][kk][r] = bok ? Rre[ib] : 0.0;
            if (RC) { As[1][kk][r] = aok ? Rim[ia] : 0.0; Bs[1][kk][r] = bok ? Rim[ib] : 0.0; }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            double a[4], b[4], a2[4], b2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = As[0][kk][ty * 4 + u]; b[u] = Bs[0][kk][tx * 4 + u];
                if (RC) { a2[u] = As[1][kk][ty * 4 + u]; b2[u] = Bs[1][kk][tx * 4 + u]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    ar[u][v] = fma(a[u], b[v], ar[u][v]);
                    if (RC) {
                        ar[u][v] = fma(-a2[u], b2[v], ar[u][v]);
                        ai[u][v] = fma(a[u], b2[v], ai[u][v]);
                        ai[u][v] = fma(a2[u], b[v], ai[u][v]);
                    }
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = tr + ty * 4 + u, c = tc + tx * 4 + v;
            if (r >= NM || c >= NM) continue;
            const int i = r / M, p = r % M, j = c / M, q = c % M;
            // the permuted matrix Q[(j,p),(i,q)] = A[(i,p),(j,q)] is symmetric (A = R R^T is), so g^T Q g needs the upper
            // triangle only: stored as 2 Q above the diagonal, Q on it, zero (the memset) below -- the GEMM then skips
            // the rows below the diagonal block of a column tile (KCUT): half the flops of the evaluation
            const long qa = (long)j * M + p, qb = (long)i * M + q;
            if (qa > qb) continue;
            const double f = qa < qb ? 2.0 : 1.0;
            const long dst = qa * ldo + qb;
            if (RC) ((cplx *)out_v)[dst] = cmake(f * ar[u][v], f * ai[u][v]);
            else ((double *)out_v)[dst] = f * ar[u][v];
        }
}

// Ghalf [nw, nt, M] -> fragment order (see header comment)
__global__ void gfrag_kernel(const cplx *ghalf, double *gfrag, int nw, int nt, int M, int nwt, int nks) {
    const int lane = threadIdx.x & 63;
    const long f = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // (orb, wt, ks)
    const long nf = (long)nt * nwt * nks;
    if (f >= nf) return;
    const int ks = (int)(f % nks);
    const int wt = (int)((f / nks) % nwt);
    const int orb = (int)(f / ((long)nks * nwt));
    const int w = wt * 16 + (lane & 15), p = ks * 4 + (lane >> 4);
    cplx v = cmake(0.0, 0.0);
    if (w < nw && p < M) v = ghalf[((long)w * nt + orb) * M + p];
    gfrag[(((long)orb * 2 + 0) * nwt + wt) * nks * 64 + (long)ks * 64 + lane] = v.x;
    gfrag[(((long)orb * 2 + 1) * nwt + wt) * nks * 64 + (long)ks * 64 + lane] = v.y;
}

struct EFinArgs {
    int M, K, nw, nt, nsplit, nxt, nwt;
    double ecore;
    const cplx *rH1, *ghalf, *vbias, *part;
    cplx *energy;
    // quadratic-form exchange: E[2 * qsplit, nw, ncb] partial sums of (g^T Atil)[q] g[q] (null: exx_kernel partials in `part`)
    const cplx *Eq;
    int qsplit, na, nb, ncb;
    // closed_try: Eq holds two passes of 2 qsplit batches each -- alpha's slices, then beta's; the beta pass was skipped on the
    // device, and the alpha sums count twice, when *closed_bad < closed_epoch (every walker's Ghalf_b == Ghalf_a)
    int closed_try;
    const unsigned long long *closed_bad;
    unsigned long long closed_epoch;
    unsigned long long *counters;       // afq_counters_ext [4]
};

// EF_THR threads per walker: the kernel streams Ghalf, rH1, the Coulomb partials and (quadratic-form exchange) the
// Y partials of its walker once; 16 waves per CU keep enough loads in flight
#define EF_THR 1024
__global__ __launch_bounds__(EF_THR) void energy_finish_kernel(EFinArgs a) {
    __shared__ double red[EF_THR / 64][6];
    const int w = blockIdx.x, tid = threadIdx.x;
    // one-body
    double e1r = 0, e1i = 0;
    const long nq = (long)a.nt * a.M;
    const cplx *gh = a.ghalf + (long)w * nq;
    double exr = 0, exi = 0;
    for (long q = tid; q < nq; q += EF_THR) {
        const cplx h = a.rH1[q], g = gh[q];
        e1r += h.x * g.x - h.y * g.y;
        e1i += h.x * g.y + h.y * g.x;
    }
    if (a.Eq) {          // exchange: the per-tile partial sums the GEMM epilogue left, in a fixed order
        // (the batches of a spin without electrons are never written; two-pass scheme: qsplit IS the batches of a pass)
        const int nb_ = a.closed_try ? a.qsplit : a.qsplit * (a.nb > 0 ? 2 : 1);
        const bool closed = a.closed_try && *a.closed_bad < a.closed_epoch;
        const int npass = a.closed_try && !closed ? 2 : 1;
        for (int t = tid; t < npass * nb_ * a.ncb; t += EF_THR) {
            const int b = t / a.ncb, ct = t % a.ncb;             // (the second pass's batches lie right behind the first's)
            const cplx v = a.Eq[((long)b * a.nw + w) * a.ncb + ct];
            exr += v.x; exi += v.y;
        }
        // (the verdict is the population's: one work-group counts every walker of the launch -- an atomic per walker on one
        //  address cost the launch 2 us)
        if (closed) { exr *= 2.0; exi *= 2.0; if (tid == 0 && w == 0 && a.counters) atomicAdd(&a.counters[4], (unsigned long long)a.nw); }
    }
    // Coulomb
    double ecr = 0, eci = 0;
    for (int n = tid; n < a.K; n += EF_THR) {
        // (the split-K partial sums of a field in flights of eight, summed in the order of the plain loop)
        cplx x = cmake(0.0, 0.0);
        const cplx *vb = a.vbias + (long)w * a.K + n;
        const long bs = (long)a.nw * a.K;
        const int nb2 = 2 * a.nsplit;
        int b = 0;
        for (; b + 7 < nb2; b += 8) {
            cplx t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = vb[(b + j) * bs];
#pragma unroll
            for (int j = 0; j < 8; ++j) x = cadd(x, t[j]);
        }
        for (; b < nb2; ++b) x = cadd(x, vb[b * bs]);
        ecr += x.x * x.x - x.y * x.y;
        eci += 2.0 * x.x * x.y;
    }
    // exchange partials of this walker (exx_kernel path)
    const int wt = w >> 4, wl = w & 15;
    const int np = a.Eq ? 0 : 2 * a.nxt * EXX_CHUNKS;
    for (int t = tid; t < np; t += EF_THR) {
        const int chunk = t % EXX_CHUNKS;
        const int xt = (t / EXX_CHUNKS) % a.nxt;
        const int s = t / (EXX_CHUNKS * a.nxt);
        const long task = (((long)s * a.nxt + xt) * a.nwt + wt) * EXX_CHUNKS + chunk;
        const cplx v = a.part[task * 16 + wl];
        exr += v.x; exi += v.y;
    }
    // block sums: one reduction for all six (wave shuffles, one barrier; the order of the sums is that of six separate ones)
    double v[6] = {e1r, e1i, ecr, eci, exr, exi};
#pragma unroll
    for (int k = 0; k < 6; ++k)
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) red[tid >> 6][k] = v[k];
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            double tot = 0.0;
            for (int i = 0; i < EF_THR / 64; ++i) tot += red[i][k];
            v[k] = tot;
        }
        const double e2r = 0.5 * (v[2] - v[4]), e2i = 0.5 * (v[3] - v[5]);
        a.energy[3 * w + 0] = cmake(v[0] + e2r + a.ecore, v[1] + e2i);
        a.energy[3 * w + 1] = cmake(v[0] + a.ecore, v[1]);
        a.energy[3 * w + 2] = cmake(e2r, e2i);
    }
}

// Builds afrag from the host rchol (c128 [(na+nb) M, K]); called once from
// afq_set_system_generic.  Also decides real / complex rchol.
int k_prepare_energy_operands(afq_handle *h, const double *rchol_host) {
    const int M = h->M, K = h->K;
    const int nks = (M + 3) / 4, nxt = (K + 15) / 16;
    for (int s = 0; s < 2; ++s) {
        const int ns = s == 0 ? h->na : h->nb, off = s == 0 ? 0 : h->na;
        if (ns == 0) continue;
        const size_t n = (size_t)ns * nxt * nks * 64;
        std::vector<double> fr(n, 0.0), fi;
        if (!h->rchol_real) fi.assign(n, 0.0);
        for (int i = 0; i < ns; ++i)
            for (int xt = 0; xt < nxt; ++xt)
                for (int ks = 0; ks < nks; ++ks)
                    for (int l = 0; l < 64; ++l) {
                        const int x = xt * 16 + (l & 15), p = ks * 4 + (l >> 4);
                        if (x >= K || p >= M) continue;
                        const size_t src = (((size_t)(off + i) * M + p) * K + x) * 2;
                        const size_t dst = (((size_t)i * nxt + xt) * nks + ks) * 64 + l;
                        fr[dst] = rchol_host[src];
                        if (!h->rchol_real) fi[dst] = rchol_host[src + 1];
                    }
        AFQ_HIP(h, hipMalloc(&h->rchol_frag[s], n * sizeof(double)));
        AFQ_HIP(h, hipMemcpy(h->rchol_frag[s], fr.data(), n * sizeof(double), hipMemcpyHostToDevice));
        if (!h->rchol_real) {
            AFQ_HIP(h, hipMalloc(&h->rchol_frag_im[s], n * sizeof(double)));
            AFQ_HIP(h, hipMemcpy(h->rchol_frag_im[s], fi.data(), n * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    return AFQ_OK;
}

void k_free_atil(void *(&atil)[2]) {
    if (atil[0]) hipFree(atil[0]);
    if (atil[1] && atil[1] != atil[0]) hipFree(atil[1]);
    atil[0] = atil[1] = nullptr;
}

// bytes of the quadratic-form operands of ONE determinant
static double atil_bytes(afq_handle *h) {
    const double e = h->rchol_real ? 8.0 : 16.0;
    const double a = (double)h->na * h->M, b = (double)h->nb * h->M;
    return e * (a * a + (h->rchol_same ? 0.0 : b * b));
}

int k_exchange_uses_quadratic(afq_handle *h) {
    if (h->exx_mode == 1) return 0;
    if (h->exx_mode == 2) return 1;
    if (h->atil_unavailable) return 0;           // an earlier allocation failed: T-intermediate kernel from then on
    if (h->K < h->M) return 0;                    // K / M times fewer flops only when K >= M
    if (h->atil[0]) return 1;                     // already built
    // the operands of every determinant must fit what is free NOW on this GPU (other handles, other processes), and at
    // most a quarter of the memory: the walkers, the HS potential and the work buffers want the rest
    size_t free_b = 0, total_b = 0;
    hipSetDevice(h->device);
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return atil_bytes(h) * h->ndet <= 72e9; }
    const double need = atil_bytes(h) * h->ndet;
    return need <= 0.25 * (double)total_b && need <= 0.8 * (double)free_b;
}

static int ensure_atil(afq_handle *h) {
    if (h->atil[0]) return AFQ_OK;
    const int M = h->M;
    for (int s = 0; s < 2; ++s) {
        const int ns = s == 0 ? h->na : h->nb;
        if (ns == 0) continue;
        if (s == 1 && h->rchol_same) { h->atil[1] = h->atil[0]; break; }
        const long NM = (long)ns * M, ldq = (NM + 1) & ~1L;
        const size_t bytes = (size_t)NM * ldq * (h->rchol_real ? sizeof(double) : sizeof(cplx));
        if (hipMalloc(&h->atil[s], bytes) != hipSuccess) {
            (void)hipGetLastError();
            h->atil[s] = nullptr;
            k_free_atil(h->atil);                 // the other spin's operand, if it was made
            AFQ_FAIL(h, AFQ_ENOMEM, "quadratic-form exchange operand does not fit");
        }
        AFQ_HIP(h, hipMemsetAsync(h->atil[s], 0, bytes, h->stream));
        const dim3 grid((unsigned)((NM + 63) / 64), (unsigned)((NM + 63) / 64));
        const long row0 = s == 0 ? 0 : (long)h->na * M;
        if (h->rchol_real)
            AFQ_LAUNCH(h, atilde_build_kernel<false>, grid, dim3(256), 0, h->stream, h->rchol_re, h->rchol_im, h->ld_rc, h->K, ns, M, row0, h->atil[s], ldq);
        else
            AFQ_LAUNCH(h, atilde_build_kernel<true>, grid, dim3(256), 0, h->stream, h->rchol_re, h->rchol_im, h->ld_rc, h->K, ns, M, row0, h->atil[s], ldq);
        AFQ_POST(h);
    }
    return AFQ_OK;
}

// Closed-shell verdict for a Ghalf no Green's function launch has checked (the large-system path, k_bigdet.hip: three GEMM-shaped
// launches, none of which holds a whole walker): one work-group per walker compares the two spin blocks of the stored Ghalf
// bit for bit -- EVERY walker, the energy is evaluated for every one -- and raises closed_bad to this launch's epoch when they
// differ (afq_internal.h: closed_bad).  nw * nt * M * 16 bytes read once (C5 sizes: 164 MB, ~45 us) ahead of an exchange
// evaluation whose beta half (3 ms there) it can spare.
__global__ __launch_bounds__(256) void ghalf_closed_check_kernel(const cplx *ghalf, long half, unsigned long long *closed_bad,
                                                                 unsigned long long epoch) {
    const double2 *a = (const double2 *)(ghalf + (long)blockIdx.x * 2 * half), *b = a + half;
    int same = 1;
    for (long e = threadIdx.x; e < half; e += 256) {
        const double2 x = a[e], y = b[e];
        same &= (int)((__double_as_longlong(x.x) == __double_as_longlong(y.x)) & (__double_as_longlong(x.y) == __double_as_longlong(y.y)));
    }
    if (!__syncthreads_and(same) && threadIdx.x == 0) atomicMax(closed_bad, epoch);
}

// (*S_out: contraction slices per spin of the un-split scheme; *two_pass: alpha and beta in two launches of 2 S slices each)
template <bool RC>
static int launch_exx_quadratic(afq_handle *h, int *S_out, bool *two_pass) {
    const int M = h->M;
    const long nma = (long)h->na * M, nmb = (long)h->nb * M, nmax = nma > nmb ? nma : nmb;
    // contraction slices: enough 64 x 64 work-group tiles to fill the chip about twice
    const long tiles = (long)((h->nw + 63) / 64) * ((nmax + 63) / 64) * (h->nb > 0 ? 2 : 1);
    int S = (int)((1000 + tiles - 1) / tiles);
    if (S < 1) S = 1;
    if (S > EXQ_MAX_BATCH / 2) S = EXQ_MAX_BATCH / 2;
    while (S > 1 && nmax / S < 64) --S;
    S = AFQ_KNOB_INT("AFQ_EXQ_SPLIT", S);
    // Closed-shell population (every walker's Ghalf_b == Ghalf_a, verified on the device by the Green's function launch this
    // Ghalf comes from: closed_checked_version) and one Atil for both spins: the 2 S slices of the FIRST launch all belong to
    // spin alpha (every XCD busy), the second launch holds spin beta's and returns at once on the device when the flag says
    // closed; energy_finish_kernel then counts the alpha sums twice.  Nothing is decided on the host.
    if (h->closed_bad && h->closed_checked_version != h->ghalf_version && k_greens_big_supported(h) && h->ndet == 1 &&
        h->na == h->nb && h->atil[0] == h->atil[1] && !h->exx_open_hint && !AFQ_KNOB_SET("AFQ_NO_CLOSED_EXX")) {
        // (an open-shell population is found out by the first evaluation's published verdict, exx_open_hint: no check, and
        //  the two-spin launch, from then on)
        AFQ_LAUNCH(h, ghalf_closed_check_kernel, dim3(h->nw), dim3(256), 0, h->stream, h->ghalf, nma, h->closed_bad, ++h->closed_epoch);
        AFQ_POST(h);
        h->closed_checked_version = h->ghalf_version;
    }
    const bool closed_try = h->closed_bad && h->closed_checked_version == h->ghalf_version && h->closed_checked_version != 0 &&
                            h->ndet == 1 && h->na == h->nb && h->atil[0] == h->atil[1] && !h->exx_open_hint && !AFQ_KNOB_SET("AFQ_NO_CLOSED_EXX");
    // (slices of the one-spin launch: 2 S, as many work-groups as the two-spin launch has.  C3, us per evaluation: S = 4
    //  slices 106.6, 5 100.9, 6 96.9, 7 91.4, 8 = 2 S 96.9, 10 139.8, 16 107.3; the two-spin launch 138.9)
    int SL = closed_try ? 2 * S : S;
    if (closed_try) SL = AFQ_KNOB_INT("AFQ_EXQ_CLOSED_SL", SL);
    if (SL > EXQ_MAX_BATCH) SL = EXQ_MAX_BATCH;
    const int NB = closed_try ? SL : 2 * S;                      // batches of one launch
    ExxQProb<RC> p;
    p.batch = NB; p.rows = h->nw; p.cols = (int)nmax; p.astride = (long)h->nt * M;
    p.ghalf = h->ghalf; p.zero = (const cplx *)h->zero_page;
    p.skip_flag = nullptr; p.skip_epoch = 0; p.skip_from = 0;
    // a whole number of rounds over the eight XCDs per spin (C3: 8 slices): the beta slices ride in the alpha launch, behind
    // the alpha slices in every XCD's queue -- work-groups that return at once instead of a launch that returns at once
    // (5.0 us per evaluation in the kernel trace); otherwise (C5 sizes: 2 slices) the beta launch keeps the alpha launch's
    // spread over the XCDs
    const bool merged = closed_try && NB % 8 == 0 && 2 * NB <= EXQ_MAX_BATCH && !AFQ_KNOB_SET("AFQ_EXQ_TWO_LAUNCHES");
    const int npass = closed_try && !merged ? 2 : 1;
    const int nbat = merged ? 2 * NB : NB;
    p.batch = nbat;
    // (ONE event pair around both launches: with a closed-shell population the second is a few microseconds of work-groups
    //  that return at once, and the pair times what the evaluation costs)
    KernelTrace kt(h, AFQ_K_EXCHANGE);
    for (int pass = 0; pass < npass; ++pass) {
    int kmax = 0;
    for (int b = 0; b < nbat; ++b) {
        const int s = closed_try ? (merged ? b / NB : pass) : b / S, sl = closed_try ? b % NB : b % S;
        const long tot = s == 0 ? nma : nmb;
        if (tot == 0) { p.goff[b] = 0; p.soff[b] = 0; p.len[b] = 0; p.ncol[b] = 0; p.B[b] = h->zero_page; p.ldq[b] = 0; p.k0[b] = 0; continue; }
        const long ldq = (tot + 1) & ~1L;
        // slices of EQUAL WORK on the triangular operand: row a meets tot - a columns, so the boundaries sit at
        // tot (1 - sqrt(1 - s / S)) (rounded to whole k-chunks); every slice is one batch = one XCD's share
        auto bound = [&](int x) -> long {
            if (x <= 0) return 0;
            if (x >= SL) return tot;
            long r = (long)((double)tot * (1.0 - std::sqrt(1.0 - (double)x / SL)));
            r = (r + 7) & ~7L;
            return r > tot ? tot : r;
        };
        long k0 = bound(sl), l = bound(sl + 1) - k0;
        if (l < 0) { l = 0; k0 = 0; }
        p.goff[b] = (s ? nma : 0) + k0; p.soff[b] = s ? nma : 0;
        p.len[b] = (int)l; p.ncol[b] = (int)tot;
        p.B[b] = h->rchol_real ? (const void *)((const double *)h->atil[s] + k0 * ldq) : (const void *)((const cplx *)h->atil[s] + k0 * ldq);
        p.ldq[b] = ldq; p.k0[b] = k0;
        if (l > kmax) kmax = (int)l;
    }
    p.kdim = kmax;
    p.ncb = (int)((nmax + 15) / 16);
    const size_t per_pass = (size_t)NB * h->nw * p.ncb, need = per_pass * (closed_try ? 2 : 1);
    if (h->exq_y_len < need) {
        if (h->exq_y) hipFree(h->exq_y);
        AFQ_HIP(h, hipMalloc(&h->exq_y, sizeof(cplx) * need));
        h->exq_y_len = need;
    }
    p.E = h->exq_y + per_pass * pass;
    if (pass == 1 || merged) { p.skip_flag = h->closed_bad; p.skip_epoch = h->closed_epoch; p.skip_from = merged ? NB : 0; }
    // short contractions (several slices: C3 sizes) run better on eight waves with a 1 x 2 tile block each (146 vs 165 us),
    // long ones (one slice: C5 sizes) on four waves with 2 x 2 (11.53 vs 11.61 ms per step)
    // (round 4: the loader-wave configuration for long contractions too: C5 sizes 6.63 -> 6.36 ms per evaluation)
    {
        if (pass == 0) {   // every configuration below multiplies 64 x 64 work-group tiles; the contraction of a tile is its KCUT length
            auto klen = [&](int b, int col0, int ncols) -> long {
                const long need = (long)col0 + ncols - p.k0[b];
                return need <= 0 ? 0 : (need < p.len[b] ? need : p.len[b]);
            };
            ExxQProb<RC> pa = p;
            pa.batch = NB;             // (the alpha slices, or both spins' of the plain launch: what a closed-shell population executes)
            h->issued_flops[AFQ_K_EXCHANGE] = mfma_gemm_wg_issued_flops<2, 2, 2, 2, ExxQProb<RC>, RC>(pa, klen);
        }
        // round 4: four compute waves with 2 x 2 tiles + four loader waves (STAG = 3: the ring refill kept out of the waves
        // that issue MFMAs; see the HS-potential GEMM in k_gemm.hip): 155 -> 139 us per evaluation at C3 against the eight
        // compute waves with 1 x 2 tiles that refill the ring themselves (the round-3 choice)
        // round 5: a complex Atil (3-multiplication products, 144 VGPRs) on the lean loop at two work-groups per CU
        // (k_apply_exponential in k_gemm.hip): 9.95 -> 9.74 ms per determinant at C5; the real one (99 VGPRs) is two per CU anyway
        // (short contractions -- several slices: C3 sizes -- on eight waves with a 1 x 2 tile block each, the ring depths, the
        //  pipelined loops and the two-work-groups-per-CU variants measured against these are in tuning builds: AFQ_EXQ_CFG)
#ifdef AFQ_TUNING
        const int cfg = AFQ_KNOB_INT("AFQ_EXQ_CFG", 1);
        if (cfg == 16) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 3>(p, h->stream, h->zero_page)));
        else if (cfg == 9) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC>(p, h->stream, h->zero_page)));
        else if (cfg == 4) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 2>(p, h->stream, h->zero_page)));
        else if (cfg == 5) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 4, 1, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC>(p, h->stream, h->zero_page)));
        else if (cfg == 6) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 4, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC>(p, h->stream, h->zero_page)));
        else if (cfg == 7) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 3>(p, h->stream, h->zero_page)));
        else if (cfg == 8) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 3>(p, h->stream, h->zero_page)));
        else if (cfg == 12) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 2, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 3>(p, h->stream, h->zero_page)));
        else if (cfg == 13) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 2, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 2>(p, h->stream, h->zero_page)));
        else if (cfg == 14) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 2>(p, h->stream, h->zero_page)));
        else if (cfg == 15) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 5, 4>(p, h->stream, h->zero_page)));
        else if (cfg == 10) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 3, 4>(p, h->stream, h->zero_page)));
        else if (cfg == 11) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 2, 4>(p, h->stream, h->zero_page)));
        else if (cfg == 2) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 4, 2, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC>(p, h->stream, h->zero_page)));
        else if (cfg == 3) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD, RC>(p, h->stream, h->zero_page)));
        else if (AFQ_KNOB_SET("AFQ_EXQ_PIPE")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 2>(p, h->stream, h->zero_page)));
        else if (cfg != 1) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC>(p, h->stream, h->zero_page)));
        else
#endif
        if (pass == 1) {
            ExxQBetaProb<RC> pb;
            static_cast<ExxQProb<RC> &>(pb) = p;
            if constexpr (RC) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ExxQBetaProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 5, 4>(pb, h->stream, h->zero_page)));
            else AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ExxQBetaProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 3>(pb, h->stream, h->zero_page)));
        }
        else if constexpr (RC) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 5, 4>(p, h->stream, h->zero_page)));
        else AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ExxQProb<RC>, MAP_BATCH_XCD_ROWS, RC, 1, 3>(p, h->stream, h->zero_page)));
    }
    }   // pass
    *S_out = closed_try ? NB : S; *two_pass = closed_try;        // (two-pass: batches per pass)
    return AFQ_OK;
}

int k_energy_generic(afq_handle *h) {
    const int M = h->M, K = h->K;
    const int nks = (M + 3) / 4, nxt = (K + 15) / 16, nwt = (h->nw + 15) / 16;
    // Coulomb vectors: the force-bias contraction on the current Ghalf
    int rc = k_force_bias_generic(h);
    if (rc) return rc;
    bool quadratic = k_exchange_uses_quadratic(h);
    if (quadratic && (rc = ensure_atil(h))) {
        // automatic choice: an operand that does not fit is not an error, the T-intermediate kernel needs none
        if (rc != AFQ_ENOMEM || h->exx_mode == 2) return rc;
        h->atil_unavailable = true;
        h->err.clear();
        quadratic = false;
    }
    if (quadratic) {
        int S = 0;
        bool closed_try = false;
        rc = h->rchol_real ? launch_exx_quadratic<false>(h, &S, &closed_try) : launch_exx_quadratic<true>(h, &S, &closed_try);
        if (rc) return rc;
        EFinArgs f;
        f.M = M; f.K = K; f.nw = h->nw; f.nt = h->nt; f.nsplit = h->fb_split; f.nxt = nxt; f.nwt = nwt;
        f.ecore = h->ecore; f.rH1 = h->rH1; f.ghalf = h->ghalf; f.vbias = h->vbias; f.part = nullptr;
        f.energy = h->energy; f.Eq = h->exq_y; f.qsplit = S; f.na = h->na; f.nb = h->nb;
        f.ncb = (int)(((long)(h->na > h->nb ? h->na : h->nb) * M + 15) / 16);
        f.closed_try = closed_try ? 1 : 0; f.closed_bad = h->closed_bad; f.closed_epoch = h->closed_epoch;
        f.counters = h->counters;
        AFQ_LAUNCH(h, energy_finish_kernel, dim3(h->nw), dim3(EF_THR), 0, h->stream, f);
        AFQ_POST(h);
        return AFQ_OK;
    }
    const size_t gbytes = sizeof(double) * (size_t)h->nt * 2 * nwt * nks * 64;
    if (!h->gfrag || h->gfrag_bytes < gbytes) {
        if (h->gfrag) hipFree(h->gfrag);
        AFQ_HIP(h, hipMalloc(&h->gfrag, gbytes));
        h->gfrag_bytes = gbytes;
    }
    {
        const long nf = (long)h->nt * nwt * nks;
        AFQ_LAUNCH(h, gfrag_kernel, dim3((unsigned)((nf + 3) / 4)), dim3(256), 0, h->stream, h->ghalf,
                           h->gfrag, h->nw, h->nt, M, nwt, nks);
        AFQ_POST(h);
    }
    const long ntask = 2L * nxt * nwt * EXX_CHUNKS;
    if (h->exx_part_len < ntask * 16) {
        if (h->exx_part) hipFree(h->exx_part);
        AFQ_HIP(h, hipMalloc(&h->exx_part, sizeof(cplx) * ntask * 16));
        h->exx_part_len = ntask * 16;
    }
    ExxArgs a;
    a.M = M; a.K = K; a.nw = h->nw; a.nt = h->nt; a.nks = nks; a.nxt = nxt; a.nwt = nwt;
    a.ns[0] = h->na; a.ns[1] = h->nb; a.goff[0] = 0; a.goff[1] = h->na;
    a.afrag[0] = h->rchol_frag[0]; a.afrag[1] = h->rchol_frag[1];
    a.afrag_im[0] = h->rchol_frag_im[0]; a.afrag_im[1] = h->rchol_frag_im[1];
    a.gfrag = h->gfrag; a.part = h->exx_part;
    {
        KernelTrace kt(h, AFQ_K_EXCHANGE);
        const long per_xcd = 2L * nxt * ((nwt + 7) / 8) * EXX_CHUNKS;
        const unsigned nblk = (unsigned)(8 * ((per_xcd + 3) / 4));
        if (h->rchol_real)
            AFQ_LAUNCH(h, exx_kernel<false>, dim3(nblk), dim3(256), 0, h->stream, a);
        else
            AFQ_LAUNCH(h, exx_kernel<true>, dim3(nblk), dim3(256), 0, h->stream, a);
    }
    AFQ_POST(h);
    EFinArgs f;
    f.M = M; f.K = K; f.nw = h->nw; f.nt = h->nt; f.nsplit = h->fb_split; f.nxt = nxt; f.nwt = nwt;
    f.ecore = h->ecore; f.rH1 = h->rH1; f.ghalf = h->ghalf; f.vbias = h->vbias; f.part = h->exx_part;
    f.energy = h->energy; f.Eq = nullptr; f.qsplit = 0; f.na = h->na; f.nb = h->nb; f.ncb = 0;
    f.closed_try = 0; f.closed_bad = nullptr; f.closed_epoch = 0; f.counters = nullptr;
    AFQ_LAUNCH(h, energy_finish_kernel, dim3(h->nw), dim3(EF_THR), 0, h->stream, f);
    AFQ_POST(h);
    return AFQ_OK;
}
