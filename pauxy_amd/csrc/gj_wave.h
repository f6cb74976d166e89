// Wave-level (64 lanes, no barriers) register-resident dense kernels for small matrices (n <= 32):
// Gauss-Jordan inverse + determinant (Green's function, k_small.hip) and the inverse Cholesky factor
// (Cholesky-QR re-orthogonalisation, k_bigdet.hip).
#pragma once
#include "afq_internal.h"

// --------------------------------------------------------------------------
// Register-resident Gauss-Jordan inverse of an n x n complex matrix, n <= 32, by ONE wave:
// lane (h = lane >> 5, r = lane & 31) keeps row r, columns 16 h .. 16 h + 15, in VGPRs.  Per pivot
// step (fully unrolled, so every register index is static): the half that owns column k hands it
// to the other half with v_permlane32_swap (the owning half is a template argument of the block), a DPP wave maximum of a packed (|pivot|, row) key picks
// the pivot row among the rows not used yet (implicit pivoting: no row swaps), the pivot row crosses
// the wave through 32 x 16 bytes of LDS, and every lane updates its 16 entries.  The division by the
// pivot is deferred: row p is only scaled once, at the end.  No barrier, no atomics; the matrix never
// returns to LDS until the inverse is stored un-permuted (A^-1[step(i)][prow[j]] = W[i][j]).
// det A = sign(prow) prod_k d_k is returned as mantissa / binary exponent.
#ifdef AFQ_TUNING
// tuning builds, timing ablation only (WRONG results unless no pivoting is needed): the pivot of step k is row k, no search
static __device__ int afq_gj_nopiv = 0;
#endif
__device__ inline unsigned gj_wave_max_u32(unsigned v) {
#define AFQ_DPP_MAX(ctrl, rmask)                                                                         \
    { const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false);      \
      v = v > t ? v : t; }
    AFQ_DPP_MAX(0x111, 0xf) AFQ_DPP_MAX(0x112, 0xf) AFQ_DPP_MAX(0x114, 0xf) AFQ_DPP_MAX(0x118, 0xf)
    AFQ_DPP_MAX(0x142, 0xa) AFQ_DPP_MAX(0x143, 0xc)
#undef AFQ_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// (re, im) of one matrix entry to LDS as ONE ds_write2_b64 straight from the two register arrays: a 16-byte store costs
// four register moves per column to build its operand (64 moves per pivot row, executed by the whole wave for two lanes);
// the compiler re-fuses two plain 8-byte stores into exactly that, hence the instruction by hand.
// base = LDS byte address, slot = offset of the real part in units of 8 bytes (compile-time, < 255)
__device__ __attribute__((always_inline)) inline void gj_store_pair(unsigned base, const int slot, double re, double im) {
    switch (slot) {
#define AFQ_GJ_SP(S) case S: asm volatile("ds_write2_b64 %0, %1, %2 offset0:" #S " offset1:%3" ::"v"(base), "v"(re), "v"(im), "n"(S + 1) : "memory"); break;
        AFQ_GJ_SP(0) AFQ_GJ_SP(2) AFQ_GJ_SP(4) AFQ_GJ_SP(6) AFQ_GJ_SP(8) AFQ_GJ_SP(10) AFQ_GJ_SP(12) AFQ_GJ_SP(14)
        AFQ_GJ_SP(16) AFQ_GJ_SP(18) AFQ_GJ_SP(20) AFQ_GJ_SP(22) AFQ_GJ_SP(24) AFQ_GJ_SP(26) AFQ_GJ_SP(28) AFQ_GJ_SP(30)
#undef AFQ_GJ_SP
    default: break;
    }
}

// value of half HK (lanes 32 HK ..) handed to both halves, HK known at compile time: one swap per dword, no selects
template <int HK>
__device__ __attribute__((always_inline)) inline double gj_bcast_half_s(double x) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[HK], (int)a[HK]);
}

// One block of <= 8 pivot steps with static register indices.  Lane (h, r) keeps RJ columns of row r: columns
// RJ h .. RJ h + RJ - 1 (RJ = ceil(n / 2) rounded up to 4, 8, 12, 13 or 16: every instruction of a step -- 4 fp64 FMAs, one
// LDS store and one LDS load per register -- is issued for BOTH halves, so the columns are split evenly and no more
// registers are carried than the matrix needs; the step is bound by the instruction issue of its one wave).
//   RJ <= 8: block `it` = half `it`, registers 0 .. RJ - 1.
//   RJ > 8:  blocks 2 hk and 2 hk + 1 of half hk cover its registers 0..7 and 8..RJ - 1; the caller's register array is
//            rotated by 8 after every block (swap j <-> j + 8), so the active column of step u is always physical register u.
// The body is a loop over the blocks so that the code stays in the instruction cache instead of streaming n unrolled steps.
template <int RJ, int HK>
__device__ inline void gj_block8(double (&vr)[RJ], double (&vi)[RJ], int it, int n, int lane, bool &used,
                                 double &sx, double &sy, double &pdx, double &pdy, int &mystep, cplx *rowk) {
    const int h = lane >> 5, r = lane & 31;
    constexpr int hk = HK;                                   // (it: block of this half, 0 or 1 when RJ > 8)
    const int kbase = hk * RJ + 8 * it;
    const int nu = RJ <= 8 ? RJ : (it ? RJ - 8 : 8);
    const unsigned rowk_l = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)rowk + 16u * RJ * h;
    constexpr int NU = RJ < 8 ? RJ : 8;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int k = kbase + u;
        if (u < nu && k < n) {
            // 1. column k to every lane of its row
            const double fx = gj_bcast_half_s<HK>(vr[u]), fy = gj_bcast_half_s<HK>(vi[u]);
            // 2. pivot row: largest |re| + |im| among unused rows, compared on the top 26 bits of the double (exponent + 15
            //    mantissa bits: positive doubles order like their bit patterns), ties -> lowest row
            const unsigned mb = (unsigned)__double2hiint(fabs(fx) + fabs(fy));
            const unsigned key = used ? 0u : ((((mb >> 5) + 1u) << 5) | (unsigned)(31 - r));
            // every lane's own reciprocal, speculatively: independent of the search, it issues under the DPP chain (the
            // reciprocal of the row that wins is then two v_readlane pairs away instead of a 12-deep chain behind them)
            const double nnl = fx * fx + fy * fy;
            double dnl = __builtin_amdgcn_rcp(nnl);
            dnl = fma(fma(-nnl, dnl, 1.0), dnl, dnl);
            dnl = fma(fma(-nnl, dnl, 1.0), dnl, dnl);
            const double ixl = fx * dnl, iyl = -fy * dnl;
#ifdef AFQ_TUNING
            const int p = afq_gj_nopiv ? k : 31 - (int)(gj_wave_max_u32(key) & 31u);
#else
            const int p = 31 - (int)(gj_wave_max_u32(key) & 31u);
#endif
            const bool isp = r == p;
            used = used || isp;
            const double fxz = isp ? 0.0 : fx, fyz = isp ? 0.0 : fy;     // (the pivot row's own multiplier is zero)
            // 3. slot (., k) becomes the identity column of the pivot row; the (unscaled) pivot row goes
            //    to LDS first so that its round trip overlaps the reciprocal below
            if (h == hk) { vr[u] = isp ? 1.0 : 0.0; vi[u] = 0.0; }
            if (isp) {
#pragma unroll
                for (int j = 0; j < RJ; ++j) gj_store_pair(rowk_l, 2 * j, vr[j], vi[j]);
            }
            __builtin_amdgcn_wave_barrier();
            cplx rk[RJ];
#pragma unroll
            for (int j = 0; j < RJ; ++j) rk[j] = rowk[RJ * h + j];
            __builtin_amdgcn_sched_barrier(0);
            // 4. reciprocal of the pivot (uniform; v_rcp_f64 + two Newton steps, < 1 ulp off: the IEEE division sequence is a
            //    12-deep dependent chain), multipliers (zero for the pivot row itself).  The pivot row's lanes keep their
            //    pivot and its step for the determinant and the un-permutation at the end.
            const double ix = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ixl), p),
                                               __builtin_amdgcn_readlane(__double2loint(ixl), p));
            const double iy = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(iyl), p),
                                               __builtin_amdgcn_readlane(__double2loint(iyl), p));
            if (isp) { sx = ix; sy = iy; pdx = fx; pdy = fy; mystep = k; }
            const double mx = fxz * ix - fyz * iy, my = fxz * iy + fyz * ix;
            __builtin_amdgcn_sched_barrier(0);
            // 5. eliminate
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                vr[j] = fma(-mx, rk[j].x, vr[j]); vr[j] = fma(my, rk[j].y, vr[j]);
                vi[j] = fma(-mx, rk[j].y, vi[j]); vi[j] = fma(-my, rk[j].x, vi[j]);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (RJ > 8) {
#pragma unroll
        for (int j = 0; j + 8 < RJ; ++j) {
            double t = vr[j]; vr[j] = vr[j + 8]; vr[j + 8] = t;
            t = vi[j]; vi[j] = vi[j + 8]; vi[j + 8] = t;
        }
    }
}

// the inversion proper for one register count; `mystep`, (sx, sy) = step and reciprocal pivot of this lane's row
template <int RJ>
__device__ __attribute__((always_inline)) inline void gj_wave_rj(cplx *O, int n, int lane, bool write_inverse, cplx *rowk, cplx *piv, int *prow) {
    const int h = lane >> 5, r = lane & 31;
    double vr[RJ], vi[RJ];
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
        const int c = RJ * h + j;
        const cplx t = (r < n && c < n) ? O[r * n + c] : cmake(0.0, 0.0);
        vr[j] = t.x; vi[j] = t.y;
    }
    bool used = r >= n;                                      // rows that are no pivot candidates (any more)
    double sx = 1.0, sy = 0.0, pdx = 1.0, pdy = 0.0;
    int mystep = r;
    __builtin_amdgcn_wave_barrier();
    // (an even number of blocks per half: the register rotation of RJ > 8 is back where it started)
    constexpr int NSUB = RJ <= 8 ? 1 : 2;
    for (int it = 0; it < NSUB; ++it) gj_block8<RJ, 0>(vr, vi, it, n, lane, used, sx, sy, pdx, pdy, mystep, rowk);
    for (int it = 0; it < NSUB; ++it) gj_block8<RJ, 1>(vr, vi, it, n, lane, used, sx, sy, pdx, pdy, mystep, rowk);
    // pivots in step order and the pivot row of every step, for the determinant and the un-permutation
    if (h == 0 && r < n) { piv[mystep] = cmake(pdx, pdy); prow[mystep] = r; }
    __builtin_amdgcn_wave_barrier();
    if (write_inverse && r < n) {
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int c = RJ * h + j;
            if (c < n) O[mystep * n + prow[c]] = cmake(vr[j] * sx - vi[j] * sy, vr[j] * sy + vi[j] * sx);
        }
    }
}

// determinant from the pivots: their product (lanes 0..31, normalised after every multiply) and the parity of the pivot
// permutation from its inversion count
__device__ __attribute__((always_inline)) inline void gj_wave_det(int n, int lane, const cplx *piv, const int *prow, cplx &ph, int &la) {
    const cplx d = piv[lane & 31];
    double px = d.x, py = d.y;
    int e;
    (void)frexp(fmax(fabs(px), fabs(py)), &e);
    px = ldexp(px, -e); py = ldexp(py, -e);
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const double qx = __shfl_xor(px, off), qy = __shfl_xor(py, off);
        const int qe = __shfl_xor(e, off);
        const double tx = px * qx - py * qy, ty = px * qy + py * qx;
        int e2;
        (void)frexp(fmax(fabs(tx), fabs(ty)), &e2);
        px = ldexp(tx, -e2); py = ldexp(ty, -e2);
        e += qe + e2;
    }
    int inv = 0;
    if (lane < n) {
        const int pr = prow[lane];
        for (int j = lane + 1; j < n; ++j) inv += prow[j] < pr ? 1 : 0;
    }
    const unsigned long long odd = __ballot((inv & 1) != 0);
    const double sg = (__popcll(odd) & 1) ? -1.0 : 1.0;
    ph = cmake(sg * px, sg * py);
    la = e;
}

// O: n x n row-major in LDS (overwritten with the inverse when write_inverse); returns det via ph / la
__device__ __attribute__((always_inline)) inline void gj_wave32(cplx *O, int n, int lane, bool write_inverse, cplx *rowk, cplx *piv, int *prow,
                                 cplx &ph, int &la) {
    if (lane < 32) { piv[lane] = cmake(1.0, 0.0); prow[lane] = lane; }
    n = __builtin_amdgcn_readfirstlane(n);
    if (n <= 8) gj_wave_rj<4>(O, n, lane, write_inverse, rowk, piv, prow);
    else if (n <= 16) gj_wave_rj<8>(O, n, lane, write_inverse, rowk, piv, prow);
    else if (n <= 24) gj_wave_rj<12>(O, n, lane, write_inverse, rowk, piv, prow);
    else if (n <= 26) gj_wave_rj<13>(O, n, lane, write_inverse, rowk, piv, prow);
    else gj_wave_rj<16>(O, n, lane, write_inverse, rowk, piv, prow);
    __builtin_amdgcn_wave_barrier();
    gj_wave_det(n, lane, piv, prow, ph, la);
}

// the same for n <= 16 only: carries the register arrays of one, not five, instantiations (a caller that keeps a lot of
// its own state in registers: the blocked Gauss-Jordan of k_bigdet.hip)
__device__ __attribute__((always_inline)) inline void gj_wave16(cplx *O, int n, int lane, bool write_inverse, cplx *rowk, cplx *piv, int *prow,
                                 cplx &ph, int &la) {
    if (lane < 32) { piv[lane] = cmake(1.0, 0.0); prow[lane] = lane; }
    n = __builtin_amdgcn_readfirstlane(n);
    gj_wave_rj<8>(O, n, lane, write_inverse, rowk, piv, prow);       // (columns beyond n are zero, steps beyond n skipped)
    __builtin_amdgcn_wave_barrier();
    gj_wave_det(n, lane, piv, prow, ph, la);
}

// --------------------------------------------------------------------------
// The same inversion for n <= 16 with every lane busy: lane (q = lane >> 4, r = lane & 15) keeps columns 4 q .. 4 q + 3 of
// row r (8 VGPRs).  Per pivot step: column k goes from its quad of lanes to the other three by one v_permlane16_swap and
// one v_permlane32_swap per dword, the pivot search is a DPP maximum over ONE row of 16 lanes (the four rows hold the same
// column), the pivot row crosses through 16 x 16 bytes of LDS (four stores, four loads) and every lane updates four
// entries: about half the instructions of gj_wave_rj<8>, whose step is bound by the instruction issue of its one wave.
// Used by the blocked Gauss-Jordan of k_bigdet.hip, where the pivot tile's inversion is the longest chain of a block step.
template <int QK>
__device__ __attribute__((always_inline)) inline double gj_bcast_quad(double x) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a1 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const unsigned lo2 = a1[QK & 1], hi2 = b1[QK & 1];
    const auto a2 = __builtin_amdgcn_permlane32_swap(lo2, lo2, false, false);
    const auto b2 = __builtin_amdgcn_permlane32_swap(hi2, hi2, false, false);
    return __hiloint2double((int)b2[QK >> 1], (int)a2[QK >> 1]);
}

__device__ inline unsigned gj_row16_max_u32(unsigned v) {
#define AFQ_DPP_MAX(ctrl, rmask)                                                                         \
    { const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false);      \
      v = v > t ? v : t; }
    AFQ_DPP_MAX(0x111, 0xf) AFQ_DPP_MAX(0x112, 0xf) AFQ_DPP_MAX(0x114, 0xf) AFQ_DPP_MAX(0x118, 0xf)
#undef AFQ_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
}

template <int K>
__device__ __attribute__((always_inline)) inline void gj_quad_step(double (&vr)[4], double (&vi)[4], int n, int lane, bool &used,
                                 double &sx, double &sy, double &pdx, double &pdy, int &mystep, cplx *rowk) {
    constexpr int QK = K >> 2, U = K & 3;
    const int q = lane >> 4, r = lane & 15;
    const unsigned rowk_l = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)rowk + 64u * q;
    const double fx = gj_bcast_quad<QK>(vr[U]), fy = gj_bcast_quad<QK>(vi[U]);
    const unsigned mb = (unsigned)__double2hiint(fabs(fx) + fabs(fy));
    const unsigned key = used ? 0u : ((((mb >> 5) + 1u) << 5) | (unsigned)(31 - r));
    const double nnl = fx * fx + fy * fy;
    double dnl = __builtin_amdgcn_rcp(nnl);
    dnl = fma(fma(-nnl, dnl, 1.0), dnl, dnl);
    dnl = fma(fma(-nnl, dnl, 1.0), dnl, dnl);
    const double ixl = fx * dnl, iyl = -fy * dnl;
    const int p = 31 - (int)(gj_row16_max_u32(key) & 31u);
    const bool isp = r == p;
    used = used || isp;
    const double fxz = isp ? 0.0 : fx, fyz = isp ? 0.0 : fy;
    if (q == QK) { vr[U] = isp ? 1.0 : 0.0; vi[U] = 0.0; }
    if (isp) {
#pragma unroll
        for (int j = 0; j < 4; ++j) gj_store_pair(rowk_l, 2 * j, vr[j], vi[j]);
    }
    __builtin_amdgcn_wave_barrier();
    cplx rk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rk[j] = rowk[4 * q + j];
    __builtin_amdgcn_sched_barrier(0);
    const double ix = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ixl), p),
                                       __builtin_amdgcn_readlane(__double2loint(ixl), p));
    const double iy = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(iyl), p),
                                       __builtin_amdgcn_readlane(__double2loint(iyl), p));
    if (isp) { sx = ix; sy = iy; pdx = fx; pdy = fy; mystep = K; }
    const double mx = fxz * ix - fyz * iy, my = fxz * iy + fyz * ix;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        vr[j] = fma(-mx, rk[j].x, vr[j]); vr[j] = fma(my, rk[j].y, vr[j]);
        vi[j] = fma(-mx, rk[j].y, vi[j]); vi[j] = fma(-my, rk[j].x, vi[j]);
    }
    __builtin_amdgcn_wave_barrier();
}

// O: n x n row-major in LDS, n <= 16, overwritten with the inverse; piv16[k], prow16[k] = pivot and pivot row of step k
// afterwards (1 and k for k >= n): the determinant is the caller's business (gj_wave_det below, or all the pivot tiles of a
// blocked inversion at once)
__device__ __attribute__((always_inline)) inline void gj_wave16q_inv(cplx *O, int n, int lane, cplx *rowk, cplx *piv16, int *prow16) {
    if (lane < 16) { piv16[lane] = cmake(1.0, 0.0); prow16[lane] = lane; }
    n = __builtin_amdgcn_readfirstlane(n);
    const int q = lane >> 4, r = lane & 15;
    double vr[4], vi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * q + j;
        const cplx t = (r < n && c < n) ? O[r * n + c] : cmake(0.0, 0.0);
        vr[j] = t.x; vi[j] = t.y;
    }
    bool used = r >= n;
    double sx = 1.0, sy = 0.0, pdx = 1.0, pdy = 0.0;
    int mystep = r;
    __builtin_amdgcn_wave_barrier();
#define AFQ_GJ_QS(K) if (K < n) gj_quad_step<K>(vr, vi, n, lane, used, sx, sy, pdx, pdy, mystep, rowk);
    AFQ_GJ_QS(0) AFQ_GJ_QS(1) AFQ_GJ_QS(2) AFQ_GJ_QS(3) AFQ_GJ_QS(4) AFQ_GJ_QS(5) AFQ_GJ_QS(6) AFQ_GJ_QS(7)
    AFQ_GJ_QS(8) AFQ_GJ_QS(9) AFQ_GJ_QS(10) AFQ_GJ_QS(11) AFQ_GJ_QS(12) AFQ_GJ_QS(13) AFQ_GJ_QS(14) AFQ_GJ_QS(15)
#undef AFQ_GJ_QS
    if (q == 0 && r < n) { piv16[mystep] = cmake(pdx, pdy); prow16[mystep] = r; }
    __builtin_amdgcn_wave_barrier();
    if (r < n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * q + j;
            if (c < n) O[mystep * n + prow16[c]] = cmake(vr[j] * sx - vi[j] * sy, vr[j] * sy + vi[j] * sx);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// the same with the determinant (piv, prow: 32 entries each, as gj_wave32)
__device__ __attribute__((always_inline)) inline void gj_wave16q(cplx *O, int n, int lane, bool write_inverse, cplx *rowk, cplx *piv, int *prow,
                                  cplx &ph, int &la) {
    if (lane < 32) { piv[lane] = cmake(1.0, 0.0); prow[lane] = lane; }
    gj_wave16q_inv(O, n, lane, rowk, piv, prow);
    gj_wave_det(__builtin_amdgcn_readfirstlane(n), lane, piv, prow, ph, la);
}

// --------------------------------------------------------------------------
// Inverse Cholesky factor of a Hermitian positive definite n x n matrix, n <= 32, by ONE wave; same
// data layout and block structure as gj_block8, but the pivot of step k is row k (no search) and only
// the rows below it are eliminated: forward elimination of [S | I] in place, so that at the end
// v[i][j] (j < i) = Ltilde^-1[i][j] with S = Ltilde D Ltilde^H and piv[k] = D_k.
template <int RJ, int HK>
__device__ __attribute__((always_inline)) inline void chol_block8(double (&vr)[RJ], double (&vi)[RJ], int it, int n, int lane,
                                                                  cplx *rowk, double *piv, bool &bad) {
    const int h = lane >> 5, r = lane & 31;
    constexpr int hk = HK;                                   // (it: block of this half, 0 or 1 when RJ > 8)
    const int kbase = hk * RJ + 8 * it;
    const int nu = RJ <= 8 ? RJ : (it ? RJ - 8 : 8);
    const unsigned rowk_l = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)rowk + 16u * RJ * h;
    constexpr int NU = RJ < 8 ? RJ : 8;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int k = kbase + u;
        if (u < nu && k < n) {
            const double fx = gj_bcast_half_s<HK>(vr[u]), fy = gj_bcast_half_s<HK>(vi[u]);
            const bool isp = r == k;
            if (h == hk) { vr[u] = isp ? 1.0 : 0.0; vi[u] = 0.0; }
            if (isp) {
#pragma unroll
                for (int j = 0; j < RJ; ++j) gj_store_pair(rowk_l, 2 * j, vr[j], vi[j]);
            }
            __builtin_amdgcn_wave_barrier();
            cplx rk[RJ];
#pragma unroll
            for (int j = 0; j < RJ; ++j) rk[j] = rowk[RJ * h + j];
            __builtin_amdgcn_sched_barrier(0);
            const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(fx), k),
                                              __builtin_amdgcn_readlane(__double2loint(fx), k));
            bad = bad || !(d > 0.0);
            // reciprocal by v_rcp_f64 + two Newton steps (see gj_block8: the IEEE division is a 12-deep dependent chain)
            double dinv = __builtin_amdgcn_rcp(d);
            dinv = fma(fma(-d, dinv, 1.0), dinv, dinv);
            dinv = fma(fma(-d, dinv, 1.0), dinv, dinv);
            if (lane == 0) piv[k] = d;
            const double mx = r > k ? fx * dinv : 0.0, my = r > k ? fy * dinv : 0.0;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                vr[j] = fma(-mx, rk[j].x, vr[j]); vr[j] = fma(my, rk[j].y, vr[j]);
                vi[j] = fma(-mx, rk[j].y, vi[j]); vi[j] = fma(-my, rk[j].x, vi[j]);
            }
        }
    }
    if constexpr (RJ > 8) {
#pragma unroll
        for (int j = 0; j + 8 < RJ; ++j) {
            double t = vr[j]; vr[j] = vr[j + 8]; vr[j + 8] = t;
            t = vi[j]; vi[j] = vi[j + 8]; vi[j + 8] = t;
        }
    }
}

// S [n, n] (leading dimension ld) -> Tt[r][c] = conj(Ltilde^-1[r][c]) / sqrt(D_r) (c < r), 1 / sqrt(D_r) (c == r), 0 above
// (leading dimension ldt; S and Tt may be the same array: everything is loaded before anything is stored); piv[k] = D_k
template <int RJ>
__device__ __attribute__((always_inline)) inline void chol_wave_rj(const cplx *S, long ld, cplx *Tt, long ldt, int n, int lane,
                                                                   cplx *rowk, double *piv, bool &bad) {
    const int h = lane >> 5, r = lane & 31;
    double vr[RJ], vi[RJ];
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
        const int c = RJ * h + j;
        const cplx t = (r < n && c < n) ? S[(long)r * ld + c] : cmake(0.0, 0.0);
        vr[j] = t.x; vi[j] = t.y;
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int NSUB = RJ <= 8 ? 1 : 2;     // (an even number of blocks per half: the register rotation is undone)
    for (int it = 0; it < NSUB; ++it) chol_block8<RJ, 0>(vr, vi, it, n, lane, rowk, piv, bad);
    for (int it = 0; it < NSUB; ++it) chol_block8<RJ, 1>(vr, vi, it, n, lane, rowk, piv, bad);
    __builtin_amdgcn_wave_barrier();
    if (r < n) {
        const double rs = 1.0 / sqrt(piv[r]);
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int c = RJ * h + j;
            if (c >= n) continue;
            cplx t = cmake(0.0, 0.0);
            if (c < r) t = cmake(vr[j] * rs, -vi[j] * rs);
            else if (c == r) t = cmake(rs, 0.0);
            Tt[(long)r * ldt + c] = t;
        }
    }
}

// n <= 32, one wave; piv must hold 32 doubles (set to 1 here); ceil(n / 2) columns per lane as in gj_wave32
__device__ __attribute__((always_inline)) inline void chol_wave32(const cplx *S, long ld, cplx *Tt, long ldt, int n, int lane,
                                                                  cplx *rowk, double *piv, bool &bad) {
    if (lane < 32) piv[lane] = 1.0;
    n = __builtin_amdgcn_readfirstlane(n);
    if (n <= 8) chol_wave_rj<4>(S, ld, Tt, ldt, n, lane, rowk, piv, bad);
    else if (n <= 16) chol_wave_rj<8>(S, ld, Tt, ldt, n, lane, rowk, piv, bad);
    else if (n <= 24) chol_wave_rj<12>(S, ld, Tt, ldt, n, lane, rowk, piv, bad);
    else if (n <= 26) chol_wave_rj<13>(S, ld, Tt, ldt, n, lane, rowk, piv, bad);
    else chol_wave_rj<16>(S, ld, Tt, ldt, n, lane, rowk, piv, bad);
}

// --------------------------------------------------------------------------
// Inverse Cholesky factor of an n x n tile, n <= 16, in the quad layout of gj_wave16q_inv (lane (q, r) keeps columns
// 4 q .. 4 q + 3 of row r: every lane busy, a quarter of the FMAs and LDS operations of chol_wave_rj<8> per lane).
// S row-major with leading dimension ld (LDS); Tt as chol_wave_rj: conj(Ltilde^-1[r][c]) / sqrt(D_r) below the diagonal,
// 1 / sqrt(D_r) on it, zero above; piv[k] = D_k (16 doubles, 1 for k >= n).
template <int K>
__device__ __attribute__((always_inline)) inline void chol_quad_step(double (&vr)[4], double (&vi)[4], int lane, cplx *rowk, double *piv, bool &bad) {
    constexpr int QK = K >> 2, U = K & 3;
    const int q = lane >> 4, r = lane & 15;
    const unsigned rowk_l = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)rowk + 64u * q;
    const double fx = gj_bcast_quad<QK>(vr[U]), fy = gj_bcast_quad<QK>(vi[U]);
    const bool isp = r == K;
    if (q == QK) { vr[U] = isp ? 1.0 : 0.0; vi[U] = 0.0; }
    if (isp) {
#pragma unroll
        for (int j = 0; j < 4; ++j) gj_store_pair(rowk_l, 2 * j, vr[j], vi[j]);
    }
    __builtin_amdgcn_wave_barrier();
    cplx rk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rk[j] = rowk[4 * q + j];
    __builtin_amdgcn_sched_barrier(0);
    const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(fx), K),
                                      __builtin_amdgcn_readlane(__double2loint(fx), K));
    bad = bad || !(d > 0.0);
    double dinv = __builtin_amdgcn_rcp(d);
    dinv = fma(fma(-d, dinv, 1.0), dinv, dinv);
    dinv = fma(fma(-d, dinv, 1.0), dinv, dinv);
    if (lane == 0) piv[K] = d;
    const double mx = r > K ? fx * dinv : 0.0, my = r > K ? fy * dinv : 0.0;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        vr[j] = fma(-mx, rk[j].x, vr[j]); vr[j] = fma(my, rk[j].y, vr[j]);
        vi[j] = fma(-mx, rk[j].y, vi[j]); vi[j] = fma(-my, rk[j].x, vi[j]);
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __attribute__((always_inline)) inline void chol_wave16q(const cplx *S, int ld, cplx *Tt, int ldt, int n, int lane,
                                                                   cplx *rowk, double *piv, bool &bad) {
    if (lane < 16) piv[lane] = 1.0;
    n = __builtin_amdgcn_readfirstlane(n);
    const int q = lane >> 4, r = lane & 15;
    double vr[4], vi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * q + j;
        const cplx t = (r < n && c < n) ? S[r * ld + c] : cmake(0.0, 0.0);
        vr[j] = t.x; vi[j] = t.y;
    }
    __builtin_amdgcn_wave_barrier();
#define AFQ_CH_QS(K) if (K < n) chol_quad_step<K>(vr, vi, lane, rowk, piv, bad);
    AFQ_CH_QS(0) AFQ_CH_QS(1) AFQ_CH_QS(2) AFQ_CH_QS(3) AFQ_CH_QS(4) AFQ_CH_QS(5) AFQ_CH_QS(6) AFQ_CH_QS(7)
    AFQ_CH_QS(8) AFQ_CH_QS(9) AFQ_CH_QS(10) AFQ_CH_QS(11) AFQ_CH_QS(12) AFQ_CH_QS(13) AFQ_CH_QS(14) AFQ_CH_QS(15)
#undef AFQ_CH_QS
    __builtin_amdgcn_wave_barrier();
    if (r < n) {
        const double rs = 1.0 / sqrt(piv[r]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * q + j;
            if (c >= n) continue;
            cplx t = cmake(0.0, 0.0);
            if (c < r) t = cmake(vr[j] * rs, -vi[j] * rs);
            else if (c == r) t = cmake(rs, 0.0);
            Tt[r * ldt + c] = t;
        }
    }
}

