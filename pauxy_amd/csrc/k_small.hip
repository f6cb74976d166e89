// Per-walker small dense work: overlap matrices, LU (determinant + solve for
// the half-rotated Green's function), Gram-Schmidt re-orthogonalisation, the
// field shift / clipping, the phaseless weight update, population control and
// the mixed-estimator accumulation.  One workgroup per walker; panels live in
// LDS when they fit and in a global workspace otherwise.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "mfma_gemm.h"
#include "gj_wave.h"
#include "lds_dma.h"

#include "block_scan.h"
#define NTHR 256

// --------------------------------------------------------------------------
__global__ void alive_kernel(const double *weight, int *alive, int nw) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nw) alive[w] = fabs(weight[w]) > 1e-8 ? 1 : 0;   // qmc/afqmc.py:232
}

int k_alive(afq_handle *h) {
    AFQ_LAUNCH(h, alive_kernel, dim3((h->nw + 255) / 256), dim3(256), 0, h->stream, h->weight,
                       h->alive, h->nw);
    AFQ_POST(h);
    return AFQ_OK;
}

// --------------------------------------------------------------------------
// LU with partial pivoting of an n x n complex matrix O (row-major, leading
// dimension n) held in LDS or global memory.  All NTHR threads cooperate.
// perm[i] = source row of pivoted row i.  Returns (on every thread) nothing;
// thread 0 accumulates phase and log|det| into *ph, *la.
__device__ inline void lu_factor(cplx *O, int n, int *perm, int *piv_s, cplx *ph, double *la) {
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += NTHR) perm[i] = i;
    if (tid == 0) { *ph = cmake(1.0, 0.0); *la = 0.0; }
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        // pivot search by wave 0: LAPACK izamax metric |re| + |im|
        if (tid < 64) {
            double best = -1.0; int bi = k;
            for (int i = k + tid; i < n; i += 64) {
                const cplx v = O[(long)i * n + k];
                const double m = fabs(v.x) + fabs(v.y);
                if (m > best) { best = m; bi = i; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_down(best, off);
                const int oi = __shfl_down(bi, off);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (tid == 0) *piv_s = bi;
        }
        __syncthreads();
        const int p = *piv_s;
        if (p != k) {
            for (int j = tid; j < n; j += NTHR) {
                const cplx t = O[(long)k * n + j];
                O[(long)k * n + j] = O[(long)p * n + j];
                O[(long)p * n + j] = t;
            }
            if (tid == 0) { const int t = perm[k]; perm[k] = perm[p]; perm[p] = t; }
        }
        __syncthreads();
        const cplx d = O[(long)k * n + k];
        if (tid == 0) {
            const double a = hypot(d.x, d.y);
            cplx u = cmake(d.x / a, d.y / a);
            if (p != k) u = cmake(-u.x, -u.y);
            *ph = cmul(*ph, u);
            *la += log(a);
        }
        for (int i = k + 1 + tid; i < n; i += NTHR) O[(long)i * n + k] = cdiv(O[(long)i * n + k], d);
        __syncthreads();
        const int m = n - k - 1;
        for (int e = tid; e < m * m; e += NTHR) {
            const int i = k + 1 + e / m, j = k + 1 + e % m;
            cplx v = O[(long)i * n + j];
            const cplx l = O[(long)i * n + k], u = O[(long)k * n + j];
            v.x = fma(-l.x, u.x, v.x); v.x = fma(l.y, u.y, v.x);
            v.y = fma(-l.x, u.y, v.y); v.y = fma(-l.y, u.x, v.y);
            O[(long)i * n + j] = v;
        }
        __syncthreads();
    }
}

struct GreensArgs {
    int M, na, nb, nt, nw;
    const cplx *phi, *psi;
    long psi_stride;    // 0: one trial for all walkers; M*nt: walker w uses psi + w*psi_stride
    cplx *ghalf;        // may be null (determinant only)
    cplx *gsum;         // optional [nw, na*M]: Ghalf_a + Ghalf_b (na == nb), see k_force_bias_generic
    cplx *oinv;         // optional [nw, 2, nmax, nmax]: O^-1 (O = phi^T conj(psi)), row-major, leading dim nmax
    cplx *det;          // [nw]
    cplx *det_a;        // optional [nw]: the alpha spin's determinant alone (multi-determinant trials: walkers/multi_det.py:209)
    cplx *ws;           // global workspace [nw, nmax*nmax] when O does not fit LDS
    int o_in_lds;
    int only_alive;
    const int *alive;
    int dbg;            // timing experiments only (AFQ_GREENS_DBG, tuning builds; 0 in the product): 1 skip pivot loop,
                        // 2 skip phase 3, 4 skip phase 1, 8 LDS Gauss-Jordan instead of the register one
    int psi_real;       // every imaginary part of the (single, shared) trial is exactly zero (checked at upload)
    int skip_spin;      // with gsum: do not store the per-spin Ghalf (nobody will read it: afq_propagate_finish)
    int psi_closed;     // the alpha and beta blocks of the (single, shared) trial are bitwise equal, na == nb (checked at upload)
    unsigned long long *closed_bad;     // raised to closed_epoch by a walker whose spin blocks differ (afq_internal.h), or null
    unsigned long long closed_epoch;
    unsigned long long *counters;       // afq_counters_ext [5]: walkers that took the one-spin path
};

// One workgroup per walker, spins in sequence.  O = phi_s^T conj(psi_s)
// (walkers/single_det.py:310,316); Ghalf_s = O^-1 phi_s^T solved column by
// column from the LU factors; det = prod over spins of det O.
__global__ __launch_bounds__(NTHR) void greens_kernel(GreensArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int perm[256];
    __shared__ int piv_s;
    __shared__ cplx ph_s;
    __shared__ double la_s;
    const int w = blockIdx.x, tid = threadIdx.x;
    if (a.only_alive && !a.alive[w]) return;
    const int M = a.M, nt = a.nt;
    const cplx *phi = a.phi + (long)w * M * nt;
    cplx phase = cmake(1.0, 0.0);
    double logabs = 0.0;
    for (int s = 0; s < 2; ++s) {
        const int n = s == 0 ? a.na : a.nb;
        const int off = s == 0 ? 0 : a.na;
        if (n == 0) continue;
        cplx *O = a.o_in_lds ? (cplx *)smem : a.ws + (long)w * ((a.na > a.nb ? a.na : a.nb) * (long)(a.na > a.nb ? a.na : a.nb));
        for (int e = tid; e < n * n; e += NTHR) {
            const int i = e / n, j = e % n;
            cplx acc = cmake(0.0, 0.0);
            for (int p = 0; p < M; ++p) {
                const cplx x = phi[(long)p * nt + off + i];
                const cplx y = a.psi[w * a.psi_stride + (long)p * nt + off + j];
                // x * conj(y)
                acc.x = fma(x.x, y.x, acc.x); acc.x = fma(x.y, y.y, acc.x);
                acc.y = fma(x.y, y.x, acc.y); acc.y = fma(-x.x, y.y, acc.y);
            }
            O[(long)i * n + j] = acc;
        }
        __syncthreads();
        lu_factor(O, n, perm, &piv_s, &ph_s, &la_s);
        if (tid == 0) { phase = cmul(phase, ph_s); logabs += la_s; }
        if (tid == 0 && s == 0 && a.det_a) { const double e0 = exp(logabs); a.det_a[w] = cmake(phase.x * e0, phase.y * e0); }
        if (a.ghalf) {
            // column c of Ghalf_s: solve L U x = P phi_s^T[:, c]
            cplx *gh = a.ghalf + ((long)w * nt + off) * M;
            for (int c = tid; c < M; c += NTHR) {
                for (int i = 0; i < n; ++i) {
                    cplx acc = phi[(long)c * nt + off + perm[i]];
                    for (int j = 0; j < i; ++j) {
                        const cplx l = O[(long)i * n + j], y = gh[(long)j * M + c];
                        acc.x = fma(-l.x, y.x, acc.x); acc.x = fma(l.y, y.y, acc.x);
                        acc.y = fma(-l.x, y.y, acc.y); acc.y = fma(-l.y, y.x, acc.y);
                    }
                    gh[(long)i * M + c] = acc;
                }
                for (int i = n - 1; i >= 0; --i) {
                    cplx acc = gh[(long)i * M + c];
                    for (int j = i + 1; j < n; ++j) {
                        const cplx u = O[(long)i * n + j], x = gh[(long)j * M + c];
                        acc.x = fma(-u.x, x.x, acc.x); acc.x = fma(u.y, x.y, acc.x);
                        acc.y = fma(-u.x, x.y, acc.y); acc.y = fma(-u.y, x.x, acc.y);
                    }
                    gh[(long)i * M + c] = cdiv(acc, O[(long)i * n + i]);
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double e = exp(logabs);
        a.det[w] = cmake(phase.x * e, phase.y * e);
    }
}


// --------------------------------------------------------------------------
#include "weight_update.h"


static WeightArgs weight_args(afq_handle *h, cplx eshift);
static WeightArgs no_weight_args() {
    WeightArgs a;
    memset(&a, 0, sizeof(a));
    return a;
}

__global__ void weight_kernel(WeightArgs a) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.nw) return;
    weight_update_and_cap(a, w);
}

// --------------------------------------------------------------------------
// Fast path for N <= 45 electrons per spin (the overlap matrix and its inverse
// fit LDS twice over).  One 512-thread workgroup per walker: waves 0-3 work on
// spin up, waves 4-7 on spin down, in lock step.
//   phase 1  O = phi_s^T conj(psi_s)            fp64 MFMA, one 16x16 tile per wave at a time
//   phase 2  in-place Gauss-Jordan inverse with row pivoting + determinant, one wave per spin,
//            the whole N x N update of a pivot step spread over the 64 lanes
//   phase 3  Ghalf_s = O^-1 phi_s^T             fp64 MFMA, A fragment from LDS
// (reference: scipy.linalg.inv + numpy.dot + slogdet, walkers/single_det.py:310-320)
#define GS_KSMAX 32       // k-steps of 4 supported by the fast Green's kernel (M <= 128)
// wa.weight != null: the hybrid / free-projection weight update of this walker (propagation/continuous.py:264-292,
// :194-200) and the driver's weight cap run right behind its determinant (a.det IS wa.ovlp_new then), which
// saves the separate weight_kernel launch of the step.
#ifdef AFQ_TUNING
__device__ unsigned long long *afq_gs_ts = nullptr;      // [8 waves][12 stamps] of work-group 0 (AFQ_GS_TS)
#define GS_STAMP(i) do { if (afq_gs_ts && blockIdx.x == 0 && (threadIdx.x & 63) == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); afq_gs_ts[(threadIdx.x >> 6) * 12 + (i)] = t_; } } while (0)
#else
#define GS_STAMP(i)
#endif
// WGJ: both spins have n <= 32 and invert by one wave each in registers (gj_wave.h); otherwise the LDS Gauss-Jordan of
// wave 0.  Two instantiations, so that neither carries the other's registers and scalars.
template <bool INVERSE, bool WGJ>
__global__ __launch_bounds__(512) void greens_small_kernel(GreensArgs a, WeightArgs wa) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ cplx ph_s[2];
    __shared__ double la_s[2];
    __shared__ unsigned long long pmax_s[2][48];
    const int w = blockIdx.x, tid = threadIdx.x;
    if (a.only_alive && !a.alive[w]) return;
    // wave-uniform on purpose (readfirstlane): the spin's electron count n and column offset derive from g, and every
    // test on them is then a scalar branch instead of an exec-mask region
    const int g = __builtin_amdgcn_readfirstlane(tid >> 8), wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
    const int lane = tid & 63;
    unsigned long long *pmax = pmax_s[g];
    if ((tid & 255) < 48) pmax[tid & 255] = 0ull;
    const int M = a.M, nt = a.nt;
    const int nmax = a.na > a.nb ? a.na : a.nb;
    const int n = g == 0 ? a.na : a.nb, off = g == 0 ? 0 : a.na;
    cplx *O = (cplx *)smem + (long)g * (nmax * nmax + 2 * nmax);
    cplx *colk = O + nmax * nmax, *rowk = colk + nmax;
    int *piv = (int *)((cplx *)smem + 2L * (nmax * nmax + 2 * nmax)) + g * nmax;
    cplx *phi_l = (cplx *)smem + 2L * (nmax * nmax + 2 * nmax) + ((2 * nmax + 3) / 4 + 1);   // [M, nt] copy of the walker
    const cplx *phi_g = a.phi + (long)w * M * nt;
    const int lr = lane & 15, lk = lane >> 4;
    const int nt16 = (n + 15) >> 4;
    const int nks = (M + 3) >> 2;
    GS_STAMP(0);
    // ---- phases 0 and 1: the walker's Slater matrix into LDS, O = phi_s^T conj(psi_s) by MFMA (phi fragments from LDS, trial
    // fragments from memory).  k-steps go in groups of four.  The first sixteen trial fragments of the wave's first tile are
    // requested BEFORE the copy (they do not depend on the walker: their latency runs under phase 0), the rest right behind
    // the barrier, ahead of the first MFMA.  The MFMA loop is pipelined by hand: the four phi fragments of group g + 1 are
    // read from LDS before the twelve MFMAs of group g are issued, and there is ONE uniform branch per group -- with a branch
    // per k-step every LDS read sat in its own basic block right in front of the three MFMAs that wait for it (~800 cycles
    // per k-step measured, 64 x 3 of them MFMA).  Addresses are 32-bit offsets from wave-uniform bases.
    constexpr int NG = GS_KSMAX / 4;
    const int ng = (nks + 3) >> 2;
    // real trial (RHF / UHF orbitals of a real Hamiltonian, plane waves, lattice sites): x * y by two MFMAs per k-step
    // instead of three -- the phase is bound by the matrix pipe, two waves per SIMD.  The choice is made ONCE, around the
    // whole phase (a generic lambda instantiated for both cases): a test inside the loops is a branch per k-step.
    const bool yreal = WGJ && a.psi_real != 0;
    const char *psi_w = (const char *)(a.psi + w * a.psi_stride);
    const bool has_tile = wave < nt16 * nt16 && !(a.dbg & 4);
    bool closed = false;                                     // (set in phase01, uniform over the work-group)
    auto phase01 = [&](auto yr_tag) __attribute__((always_inline)) {
        constexpr bool YR = decltype(yr_tag)::value;
        using y_t = typename std::conditional<YR, double, cplx>::type;
        y_t yf[GS_KSMAX];
        // sixteen k-steps of trial fragments: unconditional loads of clamped rows, nothing that reads them in between (a
        // select on a loaded value in this block makes the compiler wait for every group of loads before it issues the
        // next); rows past M are zeroed where they are used
        auto load_trial = [&](int tj, const int g0) __attribute__((always_inline)) {
            const int jb = tj * 16 + lr, jbc = jb < n ? jb : n - 1;
            const unsigned col = (unsigned)(off + jbc) * 16u, rowb = (unsigned)nt * 16u;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int ks = g0 * 4 + q, p = ks * 4 + lk;
                yf[ks] = *(const y_t *)(psi_w + ((unsigned)(p < M ? p : M - 1) * rowb + col));
            }
        };
        const int tj0 = has_tile ? wave % nt16 : 0;
        if (has_tile) {
            load_trial(tj0, 0);
            if (ng > 4) load_trial(tj0, 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            // the walker by LDS-DMA: 1 KB per wave and instruction straight into phi_l, all requests of a wave in flight at once
            const unsigned total = (unsigned)(M * nt) * 16u;
            const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
            for (unsigned b0 = (unsigned)wave8 * 1024u; b0 < total; b0 += 8 * 1024u) {
                const unsigned bo = b0 + (unsigned)lane * 16u;
                if (bo < total) glds16((const char *)phi_g + bo, (char *)phi_l + b0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        GS_STAMP(1);
        __syncthreads();                                         // phi_l complete
        // Closed-shell walker (round 5): the trial's spin blocks are bitwise equal (host-checked) and so are THIS walker's
        // -- an RHF run, where every operator of the step acts on both spins alike.  Checked here, on the copy in LDS, every
        // time, for this walker alone: no state, nothing to invalidate.  Then O_b = O_a, det = det_a^2, Ghalf_b = Ghalf_a
        // bit for bit, and the beta half of phases 1 - 3 is not computed (spin-sum mode only: that is the step's path).
        if (WGJ && a.psi_closed && a.gsum && !a.oinv && INVERSE) {
            int same = 1;
            const int na_ = a.na;
            for (int e = tid; e < M * na_; e += 512) {
                const int p_ = e / na_, i_ = e - p_ * na_;
                const cplx x = phi_l[p_ * nt + i_], y = phi_l[p_ * nt + na_ + i_];
                same &= (__double_as_longlong(x.x) == __double_as_longlong(y.x)) & (__double_as_longlong(x.y) == __double_as_longlong(y.y));
            }
            closed = __syncthreads_and(same) != 0;
            if (!closed && tid == 0 && a.closed_bad) atomicMax(a.closed_bad, a.closed_epoch);
            if (closed && tid == 0 && a.counters) atomicAdd(&a.counters[5], 1ull);
        }
        for (int t = wave; t < ((a.dbg & 4) || (closed && g == 1) ? 0 : nt16 * nt16); t += 4) {
            const int ti = t / nt16, tj = t % nt16;
            const int ia = ti * 16 + lr;
            const int iac = ia < n ? ia : n - 1;
            if (!WGJ && t != wave) {                             // (n <= 32: four tiles per spin, one per wave)
                load_trial(tj, 0);
                if (ng > 4) load_trial(tj, 4);
            }
            d4_t accR = {0, 0, 0, 0}, accI = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
            cplx xa[4], xb[4];
            const unsigned xcol = (unsigned)(off + iac), xrow = (unsigned)nt;
            auto readx = [&](cplx (&x)[4], const int gq) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = (gq * 4 + u) * 4 + lk;
                    x[u] = phi_l[xcol + (unsigned)(p < M ? p : M - 1) * xrow];
                }
            };
            auto mf = [&](const cplx (&x)[4], const int gq) __attribute__((always_inline)) {
                const bool last = gq == ng - 1;                  // (uniform) only the last group can run past M
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool dead = last && (gq * 4 + u) * 4 + lk >= M;
                    if constexpr (YR) {
                        const double y = dead ? 0.0 : yf[gq * 4 + u];
                        accR = mfma16(x[u].x, y, accR);
                        acc3 = mfma16(x[u].y, y, acc3);
                    } else {
                        const cplx y = dead ? cmake(0.0, 0.0) : yf[gq * 4 + u];
                        // x * conj(y) by three multiplications: P1 = xr yr, P2 = xi yi, P3 = (xr + xi)(yr - yi);
                        // re = P1 + P2, im = P3 - P1 + P2 (three independent accumulator chains)
                        accR = mfma16(x[u].x, y.x, accR);
                        accI = mfma16(x[u].y, y.y, accI);
                        acc3 = mfma16(x[u].x + x[u].y, y.x - y.y, acc3);
                    }
                }
            };
            readx(xa, 0);
#pragma unroll
            for (int gq = 0; gq < NG; gq += 2) {
                if (gq < ng) {
                    if (gq + 1 < ng) readx(xb, gq + 1);
                    mf(xa, gq);
                }
                if (gq + 1 < ng) {
                    if (gq + 2 < ng) readx(xa, gq + 2 < NG ? gq + 2 : 0);
                    mf(xb, gq + 1);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + lk + 4 * r, j = tj * 16 + lr;
                if (i < n && j < n) O[i * n + j] = YR ? cmake(accR[r], acc3[r]) : cmake(accR[r] + accI[r], acc3[r] - accR[r] + accI[r]);
            }
            if (WGJ) break;
        }
    };
    if (yreal) phase01(std::true_type{});
    else phase01(std::false_type{});
    GS_STAMP(2);
    __syncthreads();
    GS_STAMP(3);
    // ---- phase 2
    __shared__ cplx gj_row[2][32], gj_piv[2][32];
    __shared__ int gj_prow[2][32];
    // (the two single-wave inversions run on different SIMDs: spin up on wave 0, spin down on wave 1 of its group = wave 5)
    if (WGJ) {
        if (wave == g && !(a.dbg & 1) && !(closed && g == 1)) {
            cplx ph;
            int la;
            gj_wave32(O, n, lane, INVERSE, gj_row[g], gj_piv[g], gj_prow[g], ph, la);
            if (lane == 0) {
                ph_s[g] = ph; la_s[g] = (double)la;
                if (closed) { ph_s[1] = ph; la_s[1] = (double)la; }         // det O_b = det O_a
            }
        }
    } else if (wave == 0 && !(a.dbg & 1)) {
        // det = prod of pivots, kept as (mantissa, binary exponent) so that neither log, exp nor
        // hypot sits on the per-pivot critical path
        cplx ph = cmake(1.0, 0.0);
        int la = 0;
        const int cw_shift = n <= 32 ? 5 : 6;
        const int cj = lane & ((1 << cw_shift) - 1), ri = lane >> cw_shift, rstep = 64 >> cw_shift;
        for (int k = 0; k < n; ++k) {
            // column k -> colk; pivot = first row with the largest |re|+|im| (LAPACK izamax metric):
            // one LDS atomic max on the bit pattern of the (non-negative) metric, then a ballot
            unsigned long long mbits = 0;
            if (lane < n) {
                const cplx v = O[lane * n + k];
                colk[lane] = v;
                if (lane >= k) {
                    mbits = (unsigned long long)__double_as_longlong(fabs(v.x) + fabs(v.y));
                    atomicMax(&pmax[k], mbits);
                }
            }
            __builtin_amdgcn_wave_barrier();
            const unsigned long long mx = pmax[k];
            const unsigned long long hit = __ballot(lane >= k && lane < n && mbits == mx);
            const int p = hit ? __ffsll((long long)hit) - 1 : k;
            if (p != k) {
                for (int j = lane; j < n; j += 64) {
                    const cplx t = O[k * n + j];
                    O[k * n + j] = O[p * n + j];
                    O[p * n + j] = t;
                }
                if (lane == 0) { const cplx t = colk[k]; colk[k] = colk[p]; colk[p] = t; }
            }
            if (lane == 0) piv[k] = p;
            __builtin_amdgcn_wave_barrier();
            const cplx d = colk[k];
            ph = cmul(ph, p != k ? cmake(-d.x, -d.y) : d);
            {
                int e;
                (void)frexp(fmax(fabs(ph.x), fabs(ph.y)), &e);
                ph = cmake(ldexp(ph.x, -e), ldexp(ph.y, -e));
                la += e;
            }
            const double dn = 1.0 / (d.x * d.x + d.y * d.y);
            const cplx dinv = cmake(d.x * dn, -d.y * dn);
            for (int j = lane; j < n; j += 64) rowk[j] = cmul(O[k * n + j], dinv);
            __builtin_amdgcn_wave_barrier();
            if (n <= 32) {
                // n x n update with every LDS read of the step in flight before the arithmetic:
                // lane (ri, cj) owns rows ri, ri+2, ... of column cj; 16 rows per lane, unrolled
                if (cj < n && (INVERSE || cj > k)) {
                    const cplx rk = rowk[cj];
                    cplx v[16], f[16];
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int i = ri + 2 * t;
                        if (i < n) { v[t] = O[i * n + cj]; f[t] = colk[i]; }
                    }
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int i = ri + 2 * t;
                        if (i < n && (INVERSE || i > k)) {
                            cplx o;
                            if (INVERSE && i == k) o = (cj == k) ? dinv : rk;
                            else if (INVERSE && cj == k) { const cplx m = cmul(f[t], dinv); o = cmake(-m.x, -m.y); }
                            else {
                                o = v[t];
                                o.x = fma(-f[t].x, rk.x, o.x); o.x = fma(f[t].y, rk.y, o.x);
                                o.y = fma(-f[t].x, rk.y, o.y); o.y = fma(-f[t].y, rk.x, o.y);
                            }
                            O[i * n + cj] = o;
                        }
                    }
                }
            } else if (INVERSE) {
                if (cj < n) {
                    const cplx rk = rowk[cj];
                    for (int i = ri; i < n; i += rstep) {
                        cplx v;
                        if (i == k) v = (cj == k) ? dinv : rk;
                        else {
                            const cplx f = colk[i];
                            if (cj == k) { const cplx t = cmul(f, dinv); v = cmake(-t.x, -t.y); }
                            else {
                                v = O[i * n + cj];
                                v.x = fma(-f.x, rk.x, v.x); v.x = fma(f.y, rk.y, v.x);
                                v.y = fma(-f.x, rk.y, v.y); v.y = fma(-f.y, rk.x, v.y);
                            }
                        }
                        O[i * n + cj] = v;
                    }
                }
            } else {
                if (cj < n && cj > k) {
                    const cplx rk = rowk[cj];
                    for (int i = k + 1 + ri; i < n; i += rstep) {
                        const cplx f = colk[i];
                        cplx v = O[i * n + cj];
                        v.x = fma(-f.x, rk.x, v.x); v.x = fma(f.y, rk.y, v.x);
                        v.y = fma(-f.x, rk.y, v.y); v.y = fma(-f.y, rk.x, v.y);
                        O[i * n + cj] = v;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (INVERSE) {
            for (int k = n - 1; k >= 0; --k) {
                const int p = piv[k];
                if (p != k) {
                    for (int i = lane; i < n; i += 64) {
                        const cplx t = O[i * n + k];
                        O[i * n + k] = O[i * n + p];
                        O[i * n + p] = t;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane == 0) { ph_s[g] = ph; la_s[g] = (double)la; }
    }
    GS_STAMP(4);
    __syncthreads();
    GS_STAMP(5);
    // (the last wave: its phase-3 item is the half-width one, the first wave's is full)
    if (tid == 448) {
        const cplx p2 = (a.dbg & 1) ? cmake(1.0, 0.0) : cmul(ph_s[0], ph_s[1]);
        const int e = (a.dbg & 1) ? 0 : (int)(la_s[0] + la_s[1]);
        a.det[w] = cmake(ldexp(p2.x, e), ldexp(p2.y, e));
        if (a.det_a) a.det_a[w] = cmake(ldexp(ph_s[0].x, (int)la_s[0]), ldexp(ph_s[0].y, (int)la_s[0]));
        if (wa.weight) weight_update_and_cap(wa, w);
    }
    GS_STAMP(6);
    if (INVERSE && a.oinv) {
        cplx *oo = a.oinv + ((long)w * 2 + g) * nmax * nmax;
        for (int e = tid & 255; e < n * n; e += 256) oo[(e / n) * nmax + (e % n)] = O[e];
    }
    // ---- phase 3: Ghalf = O^-1 phi^T.  Work item = (row tile of O^-1, PAIR of column tiles): the O^-1 fragments of the
    // row tile are read from LDS once per item and kept in registers, the two column tiles give six independent
    // accumulators (3-multiplication complex product: P1 = xr yr, P2 = xi yi, P3 = (xr + xi)(yr + yi)), and every LDS
    // read is unconditional (clamped index, zero by select) -- the previous loop re-read O^-1 for every column tile,
    // ran four dependent MFMAs per contraction step on two accumulators and branched around its operand reads.
    // With a.gsum (na == nb, host-checked) every wave of the work-group takes items of BOTH spins, so that the two Green's
    // functions of an element meet in one lane and their sum (what the force bias contracts when both spins share the
    // Cholesky block) is stored along.
    if (INVERSE && a.ghalf && !(a.dbg & 2)) {
        const bool both = a.gsum != nullptr;
        const int mt16 = (M + 15) >> 4, npair = (mt16 + 1) >> 1;
        const int first = both ? __builtin_amdgcn_readfirstlane(tid >> 6) : wave, stride = both ? 8 : 4;
        const int s_lo = both ? 0 : g, s_hi = both ? 2 : g + 1;
        const int nn = both ? a.na : n, nt16x = (nn + 15) >> 4;
        for (int it = first; nn > 0 && it < nt16x * npair; it += stride) {
            const int ti = it % nt16x, tc0 = 2 * (it / nt16x), tc1 = tc0 + 1;
            const bool two = tc1 < mt16;
            const int c0 = tc0 * 16 + lr, c1 = tc1 * 16 + lr;
            const int c0c = c0 < M ? c0 : M - 1, c1c = c1 < M ? c1 : M - 1;
            d4_t sra = {0, 0, 0, 0}, sia = {0, 0, 0, 0}, srb = {0, 0, 0, 0}, sib = {0, 0, 0, 0};
            for (int s = s_lo; s < (closed ? 1 : s_hi); ++s) {
                const int ns = s ? a.nb : a.na, offs = s ? a.na : 0;
                const cplx *Os = (const cplx *)smem + (long)s * (nmax * nmax + 2 * nmax);
                cplx *gh = a.ghalf + ((long)w * nt + offs) * M;
                const int nks3 = (ns + 3) >> 2;                      // ns <= 45: nks3 <= 12
                const int ia = ti * 16 + lr, iac = ia < ns ? ia : ns - 1;
                cplx xf[12];
#pragma unroll
                for (int ks = 0; ks < 12; ++ks) {
                    const int j = ks * 4 + lk, jc = j < ns ? j : ns - 1;
                    const cplx x = Os[iac * ns + jc];
                    xf[ks] = (ks < nks3 && j < ns && ia < ns) ? x : cmake(0.0, 0.0);
                }
                d4_t p1a = {0, 0, 0, 0}, p2a = {0, 0, 0, 0}, p3a = {0, 0, 0, 0};
                d4_t p1b = {0, 0, 0, 0}, p2b = {0, 0, 0, 0}, p3b = {0, 0, 0, 0};
                GS_STAMP(7 + 2 * (s & 1));
#pragma unroll
                for (int ks = 0; ks < 12; ++ks) {
                    if (ks < nks3) {
                        const int j = ks * 4 + lk, jc = j < ns ? j : ns - 1;
                        // (xf is zero wherever the contraction index runs past ns; columns past M are never stored)
                        const cplx y0 = phi_l[c0c * nt + offs + jc], y1 = phi_l[c1c * nt + offs + jc];
                        const cplx x = xf[ks];
                        const double xs = x.x + x.y;
                        p1a = mfma16(x.x, y0.x, p1a);
                        p2a = mfma16(x.y, y0.y, p2a);
                        p3a = mfma16(xs, y0.x + y0.y, p3a);
                        if (two) {
                            p1b = mfma16(x.x, y1.x, p1b);
                            p2b = mfma16(x.y, y1.y, p2b);
                            p3b = mfma16(xs, y1.x + y1.y, p3b);
                        }
                    }
                }
                GS_STAMP(8 + 2 * (s & 1));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = ti * 16 + lk + 4 * r;
                    const double ra = p1a[r] - p2a[r], ima = p3a[r] - p1a[r] - p2a[r];
                    const double rb = p1b[r] - p2b[r], imb = p3b[r] - p1b[r] - p2b[r];
                    if (i < ns && c0 < M && !a.skip_spin) gh[(long)i * M + c0] = cmake(ra, ima);
                    if (two && i < ns && c1 < M && !a.skip_spin) gh[(long)i * M + c1] = cmake(rb, imb);
                    sra[r] += ra; sia[r] += ima; srb[r] += rb; sib[r] += imb;
                    if (closed) {
                        // Ghalf_b = Ghalf_a: the same values into the beta rows, and twice into the spin sum (x + x: exact)
                        cplx *ghb = gh + (long)a.na * M;
                        if (i < ns && c0 < M && !a.skip_spin) ghb[(long)i * M + c0] = cmake(ra, ima);
                        if (two && i < ns && c1 < M && !a.skip_spin) ghb[(long)i * M + c1] = cmake(rb, imb);
                        sra[r] += ra; sia[r] += ima; srb[r] += rb; sib[r] += imb;
                    }
                }
            }
            if (both) {
                cplx *gs = a.gsum + (long)w * a.na * M;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = ti * 16 + lk + 4 * r;
                    if (i < a.na && c0 < M) gs[(long)i * M + c0] = cmake(sra[r], sia[r]);
                    if (two && i < a.na && c1 < M) gs[(long)i * M + c1] = cmake(srb[r], sib[r]);
                }
            }
        }
    }
    GS_STAMP(11);
}

// --------------------------------------------------------------------------
// Green's function / overlap for AT MOST 8 ELECTRONS PER SPIN (round 5: the electron gas of BASELINE configs[1] has 7 + 7,
// the 4 x 4 Hubbard lattice 8 + 8).  greens_small_kernel spends three dependent phases of 4-5 us each on such a walker --
// two 16 x 16 MFMA tiles that are 80 % padding per phase and a 7-step register Gauss-Jordan built for n = 32 -- and the
// kernel is one work-group per CU in a single round, i.e. pure latency.  Here, on the vector ALU, with every lane busy:
//   phase 0  walker AND trial into LDS by LDS-DMA (one latency)
//   phase 1  O_s[i, j] = sum_p phi[p, i] conj(psi[p, j]): thread = (spin, i, j, quarter of the p range), quad reduction
//   phase 2  in-place Gauss-Jordan inverse with implicit row pivoting, one wave per spin, ONE LANE PER ELEMENT
//            (lane = 8 row + column): per pivot step the column and the pivot row reach every lane by two lane
//            permutes, the pivot row is found by the DPP maximum of gj_wave.h, and every lane does one complex FMA
//   phase 3  Ghalf_s[i, q] = sum_j O^-1[i, j] phi[q, j]: thread per output element, stores along q
// Same arguments, outputs and riders (weight update + cap behind the determinant, spin sum, per-walker trial of the
// back-propagation, O^-1 for the Hirsch propagator) as greens_small_kernel; walkers/single_det.py:295-321, :170-199.
template <bool INVERSE>
__global__ __launch_bounds__(512) void greens_tiny_kernel(GreensArgs a, WeightArgs wa) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ cplx O_l[2][64], oinv_l[2][64], piv_l[2][32];
    __shared__ int prow_l[2][32];
    __shared__ cplx ph_s[2];
    __shared__ int la_s[2];
    const int w = blockIdx.x, tid = threadIdx.x;
    if (a.only_alive && !a.alive[w]) return;
    const int M = a.M, nt = a.nt, na = a.na, nb = a.nb;
    cplx *phi_l = (cplx *)smem, *psi_l = phi_l + (long)M * nt;
    const int lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    {   // ---- phase 0
        const unsigned total = (unsigned)(M * nt) * 16u;
        const char *phi_g = (const char *)(a.phi + (long)w * M * nt), *psi_g = (const char *)(a.psi + w * a.psi_stride);
        for (unsigned b0 = (unsigned)wave8 * 1024u; b0 < total; b0 += 8 * 1024u) {
            const unsigned bo = b0 + (unsigned)lane * 16u;
            if (bo < total) { glds16(phi_g + bo, (char *)phi_l + b0); glds16(psi_g + bo, (char *)psi_l + b0); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    {   // ---- phase 1: thread = (spin s, row i, column j, quarter c4 of the contraction)
        const int o = tid >> 2, c4 = tid & 3;
        const int s = o >> 6, i = (o >> 3) & 7, j = o & 7;
        const int ns = s ? nb : na, off = s ? na : 0;
        const bool live = i < ns && j < ns;
        const int ic = off + (live ? i : 0), jc = off + (live ? j : 0);
        double ar = 0.0, ai = 0.0;
        // (eight terms per trip: sixteen LDS reads in flight ahead of the FMAs instead of a read -> wait -> FMA chain per term)
        const int nterm = (M - c4 + 3) >> 2;
        for (int t0 = 0; t0 < nterm; t0 += 8) {
            cplx x[8], y[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p_ = c4 + 4 * (t0 + u), pc = p_ < M ? p_ : c4;
                x[u] = phi_l[pc * nt + ic]; y[u] = psi_l[pc * nt + jc];
                if (p_ >= M) x[u] = cmake(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {                          // x * conj(y)
                ar = fma(x[u].x, y[u].x, ar); ar = fma(x[u].y, y[u].y, ar);
                ai = fma(x[u].y, y[u].x, ai); ai = fma(-x[u].x, y[u].y, ai);
            }
        }
        ar += __shfl_xor(ar, 1); ai += __shfl_xor(ai, 1);
        ar += __shfl_xor(ar, 2); ai += __shfl_xor(ai, 2);
        // (rows and columns past the electron count: the identity, so that the padded matrix has the same determinant)
        if (c4 == 0) O_l[s][i * 8 + j] = live ? cmake(ar, ai) : cmake(i == j ? 1.0 : 0.0, 0.0);
        if (tid < 64) { piv_l[tid >> 5][tid & 31] = cmake(1.0, 0.0); prow_l[tid >> 5][tid & 31] = tid & 31; }
    }
    __syncthreads();
    // ---- phase 2: waves 0 and 1 (different SIMDs), spin = wave
    if (wave8 < 2) {
        const int s = wave8, n = s ? nb : na;
        const int r = lane >> 3, c = lane & 7;
        cplx v = O_l[s][lane];
        double wx = v.x, wy = v.y;
        bool used = r >= n;
        int mystep = r;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < n) {
                // column k to every lane of its row
                const int src = (lane & 0x38) | k;
                const double fx = __shfl(wx, src), fy = __shfl(wy, src);
                // pivot row: largest |re| + |im| among the rows not used yet (LAPACK's izamax metric on the top 26 bits of the
                // double, ties -> lowest row), as in gj_wave.h
                const unsigned mb = (unsigned)__double2hiint(fabs(fx) + fabs(fy));
                const unsigned key = used ? 0u : ((((mb >> 5) + 1u) << 5) | (unsigned)(31 - r));
                const int p_ = 31 - (int)(gj_wave_max_u32(key) & 31u);
                const bool isp = r == p_;
                // the pivot (uniform) and its reciprocal: v_rcp_f64 + two Newton steps
                const double dx = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(fx), p_ * 8),
                                                   __builtin_amdgcn_readlane(__double2loint(fx), p_ * 8));
                const double dy = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(fy), p_ * 8),
                                                   __builtin_amdgcn_readlane(__double2loint(fy), p_ * 8));
                // (every product-sum below is an explicit fma: the two instantiations of this kernel -- determinant only, and
                //  with the inverse -- must round alike, whatever the compiler would contract where)
                const double nn = fma(dx, dx, dy * dy);
                double dn = __builtin_amdgcn_rcp(nn);
                dn = fma(fma(-nn, dn, 1.0), dn, dn);
                dn = fma(fma(-nn, dn, 1.0), dn, dn);
                const double ix = dx * dn, iy = -dy * dn;
                // the pivot row, column k replaced by the unit entry, scaled by 1 / d
                const int psrc = p_ * 8 + c;
                double px = __shfl(wx, psrc), py = __shfl(wy, psrc);
                if (c == k) { px = 1.0; py = 0.0; }
                const double qx = fma(px, ix, -(py * iy)), qy = fma(px, iy, py * ix);
                if (isp) {
                    wx = qx; wy = qy; mystep = k;
                    if (c == 0) piv_l[s][k] = cmake(dx, dy);
                } else {
                    const double bx = c == k ? 0.0 : wx, by = c == k ? 0.0 : wy;
                    wx = fma(-fx, qx, bx); wx = fma(fy, qy, wx);
                    wy = fma(-fx, qy, by); wy = fma(-fy, qx, wy);
                }
                used = used || isp;
            }
        }
        if (c == 0 && r < n) prow_l[s][mystep] = r;
        __builtin_amdgcn_wave_barrier();
        // un-permuted inverse: A^-1[step(r)][prow[c]] = W[r][c]
        if (INVERSE && r < n && c < n) oinv_l[s][mystep * 8 + prow_l[s][c]] = cmake(wx, wy);
        cplx ph;
        int la;
        gj_wave_det(n, lane, piv_l[s], prow_l[s], ph, la);
        if (lane == 0) { ph_s[s] = ph; la_s[s] = la; }
    }
    __syncthreads();
    if (tid == 448) {
        const cplx p2 = cmul(ph_s[0], ph_s[1]);
        const int e = la_s[0] + la_s[1];
        a.det[w] = cmake(ldexp(p2.x, e), ldexp(p2.y, e));
        if (a.det_a) a.det_a[w] = cmake(ldexp(ph_s[0].x, la_s[0]), ldexp(ph_s[0].y, la_s[0]));
        if (wa.weight) weight_update_and_cap(wa, w);
    }
    if (!INVERSE) return;
    const int nmax = na > nb ? na : nb;
    if (a.oinv) {
        cplx *oo = a.oinv + (long)w * 2 * nmax * nmax;
        for (int e = tid; e < 2 * nmax * nmax; e += 512) {
            const int s = e / (nmax * nmax), rem = e - s * nmax * nmax, i = rem / nmax, j = rem - i * nmax;
            const int ns = s ? nb : na;
            if (i < ns && j < ns) oo[e] = oinv_l[s][i * 8 + j];
        }
    }
    if (!a.ghalf) return;
    // ---- phase 3
    auto ghalf_of = [&](const int s, const int i, const int q) {
        const int ns = s ? nb : na, off = s ? na : 0;
        double gr = 0.0, gi = 0.0;
        cplx x[8], y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                              // every read in flight before the first FMA
            const int jc = j < ns ? j : 0;
            x[j] = oinv_l[s][i * 8 + jc]; y[j] = phi_l[q * nt + off + jc];
            if (j >= ns) x[j] = cmake(0.0, 0.0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gr = fma(x[j].x, y[j].x, gr); gr = fma(-x[j].y, y[j].y, gr);
            gi = fma(x[j].x, y[j].y, gi); gi = fma(x[j].y, y[j].x, gi);
        }
        return cmake(gr, gi);
    };
    cplx *gh = a.ghalf + (long)w * nt * M;
    if (a.gsum) {                                              // na == nb (host-checked): both spins of an element in one thread
        cplx *gs = a.gsum + (long)w * na * M;
        for (int e = tid; e < na * M; e += 512) {
            const int i = e / M, q = e - i * M;
            const cplx ga = ghalf_of(0, i, q), gb = ghalf_of(1, i, q);
            if (!a.skip_spin) { gh[e] = ga; gh[(long)na * M + e] = gb; }
            gs[e] = cmake(ga.x + gb.x, ga.y + gb.y);
        }
    } else {
        for (int e = tid; e < nt * M; e += 512) {
            const int ii = e / M, q = e - ii * M;
            const int s = ii < na ? 0 : 1;
            gh[e] = ghalf_of(s, ii - (s ? na : 0), q);
        }
    }
}

static int launch_greens(afq_handle *h, cplx *ghalf, cplx *det, int only_alive, cplx *oinv = nullptr) {
    GreensArgs a;
    a.oinv = oinv; a.gsum = nullptr;
    if (ghalf) ++h->ghalf_version;
    a.M = h->M; a.na = h->na; a.nb = h->nb; a.nt = h->nt; a.nw = h->nw;
    a.phi = h->phi; a.psi = h->psi; a.psi_stride = h->psi_stride; a.ghalf = ghalf; a.det = det; a.ws = h->lu_ws;
    a.det_a = h->det_a_out;
    a.psi_real = h->psi_real && h->psi_stride == 0 && h->ndet <= 1;
    a.psi_closed = h->psi_closed && h->psi_stride == 0 && h->ndet <= 1 && !AFQ_KNOB_SET("AFQ_NO_CLOSED_GREENS");
    a.closed_bad = nullptr; a.closed_epoch = 0; a.counters = h->counters;
    if (ghalf == h->ghalf) h->closed_checked_version = 0;       // (set again below when THIS launch checks every walker)
    a.skip_spin = 0;
    const int nmax = h->na > h->nb ? h->na : h->nb;
    if (nmax > 256) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "more than 256 electrons per spin");
    if (k_greens_big_supported(h)) {
        // the step's weight update rides on the determinant kernel of the large path as well (conditions as below)
        if (h->fuse_weight_req && det == h->ovlp_new && !only_alive && !oinv && h->ndet <= 1) {
            const WeightArgs wa = weight_args(h, h->fuse_eshift);
            h->fuse_weight_done = true;
            h->fuse_weight_req = false;
            return k_greens_big(h, ghalf, det, oinv, &wa);
        }
        return k_greens_big(h, ghalf, det, oinv);
    }
    // (the fast kernel keeps the walker, both overlap matrices and their inverses in LDS: 160 KB per work-group)
    const size_t lds_small = sizeof(cplx) * (2 * ((size_t)nmax * nmax + 2 * nmax) + ((2 * nmax + 3) / 4 + 1) + (size_t)h->M * h->nt);
    if (nmax <= 45 && h->M <= 4 * GS_KSMAX && lds_small <= 160 * 1024) {
        a.o_in_lds = 1; a.only_alive = only_alive; a.alive = h->alive;
        // the step's weight update rides on this launch when afq_propagate asked for it and this IS the
        // overlap of the propagated walkers
        WeightArgs wa = no_weight_args();
        if (h->fuse_weight_req && det == h->ovlp_new && !only_alive && !oinv) {
            wa = weight_args(h, h->fuse_eshift);
            h->fuse_weight_done = true;
        }
        h->fuse_weight_req = false;
        static const int dbg = AFQ_KNOB_INT("AFQ_GREENS_DBG", 0);
        a.dbg = dbg;
#ifdef AFQ_TUNING
        static bool nopiv_set = false;
        if (!nopiv_set && AFQ_KNOB_SET("AFQ_GJ_NOPIV")) {
            const int one = 1;
            hipMemcpyToSymbol(HIP_SYMBOL(afq_gj_nopiv), &one, sizeof(int));
            nopiv_set = true;
        }
#endif
#ifdef AFQ_TUNING
        static unsigned long long *gsts = nullptr;
        static int gs_launch = 0;
        if (AFQ_KNOB_SET("AFQ_GS_TS")) {
            if (!gsts) { hipMalloc(&gsts, 96 * 8); hipMemset(gsts, 0, 96 * 8); hipMemcpyToSymbol(HIP_SYMBOL(afq_gs_ts), &gsts, sizeof(gsts)); }
            if (++gs_launch == 30) {
                unsigned long long t[96];
                hipStreamSynchronize(h->stream);
                hipMemcpy(t, gsts, sizeof(t), hipMemcpyDeviceToHost);
                for (int wv = 0; wv < 8; ++wv) {
                    fprintf(stderr, "gs_ts wave %d:", wv);
                    for (int i = 1; i < 12; ++i) fprintf(stderr, " %6lld", t[wv * 12 + i] ? (long long)(t[wv * 12 + i] - t[0]) : -1LL);
                    fprintf(stderr, "\n");
                }
            }
        }
#endif
        if (h->M > 4 * GS_KSMAX) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "fast Green's kernel supports M <= 128");
        const size_t lds = sizeof(cplx) * (2 * ((size_t)nmax * nmax + 2 * nmax) + ((2 * nmax + 3) / 4 + 1) +
                                           (size_t)h->M * h->nt);
        // raise the dynamic-LDS cap once per kernel and device, not per launch
        static size_t lds_set[4][AFQ_MAX_DEVICES] = {{0}};
        const bool wgj = nmax <= 32;
#ifdef AFQ_TUNING
        const bool wgj_on = wgj && !(dbg & 8);
#else
        const bool wgj_on = wgj;
#endif
        // at most 8 electrons per spin, walker + trial in LDS: the vector-ALU kernel above
        const size_t lds_tiny = sizeof(cplx) * 2 * (size_t)h->M * h->nt;
        static size_t lds_set_tiny[2][AFQ_MAX_DEVICES] = {{0}};
        const bool tiny = nmax <= 8 && h->nt >= 1 && lds_tiny <= 150 * 1024 && !dbg && !AFQ_KNOB_SET("AFQ_NO_GREENS_TINY");
        if (ghalf || oinv) {
            // the spin sum the force bias contracts (every walker written: not on the only_alive path)
            const bool want_sum = ghalf && ghalf == h->ghalf && !only_alive && k_fb_use_sum(h) && h->psi_stride == 0;
            if (want_sum) {
                if (!h->ghalf_sum) AFQ_HIP(h, hipMalloc(&h->ghalf_sum, sizeof(cplx) * (size_t)h->na * h->M * h->nw));
                a.gsum = h->ghalf_sum;
                if (h->ghalf_skip_store && (wgj_on || tiny) && !oinv) { a.skip_spin = 1; h->ghalf_skipped = true; }
            }
            if (want_sum && wgj_on && !tiny && a.psi_closed && !oinv) {
                // this launch compares the spin blocks of EVERY walker: its verdict holds for the Ghalf it leaves behind
                a.closed_bad = h->closed_bad; a.closed_epoch = ++h->closed_epoch;
                h->closed_checked_version = h->ghalf_version;
            }
            KernelTrace kt(h, AFQ_K_GREENS);
            if (tiny) {
                AFQ_HIP(h, afq_raise_lds((const void *)greens_tiny_kernel<true>, lds_tiny, lds_set_tiny[0]));
                AFQ_LAUNCH(h, (greens_tiny_kernel<true>), dim3(h->nw), dim3(512), lds_tiny, h->stream, a, wa);
            } else if (wgj_on) {
                AFQ_HIP(h, afq_raise_lds((const void *)greens_small_kernel<true, true>, lds, lds_set[0]));
                AFQ_LAUNCH(h, (greens_small_kernel<true, true>), dim3(h->nw), dim3(512), lds, h->stream, a, wa);
            } else {
                AFQ_HIP(h, afq_raise_lds((const void *)greens_small_kernel<true, false>, lds, lds_set[1]));
                AFQ_LAUNCH(h, (greens_small_kernel<true, false>), dim3(h->nw), dim3(512), lds, h->stream, a, wa);
            }
            if (want_sum) h->gsum_version = h->ghalf_version;
        } else if (tiny) {
            AFQ_HIP(h, afq_raise_lds((const void *)greens_tiny_kernel<false>, lds_tiny, lds_set_tiny[1]));
            AFQ_LAUNCH(h, (greens_tiny_kernel<false>), dim3(h->nw), dim3(512), lds_tiny, h->stream, a, wa);
        } else if (wgj_on) {
            AFQ_HIP(h, afq_raise_lds((const void *)greens_small_kernel<false, true>, lds, lds_set[2]));
            AFQ_LAUNCH(h, (greens_small_kernel<false, true>), dim3(h->nw), dim3(512), lds, h->stream, a, wa);
        } else {
            AFQ_HIP(h, afq_raise_lds((const void *)greens_small_kernel<false, false>, lds, lds_set[3]));
            AFQ_LAUNCH(h, (greens_small_kernel<false, false>), dim3(h->nw), dim3(512), lds, h->stream, a, wa);
        }
        AFQ_POST(h);
        return AFQ_OK;
    }
    if (oinv) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "inverse overlaps: the walker and both overlap matrices must fit 160 KB of LDS (N <= 45) or N > 45");
    const size_t need = sizeof(cplx) * (size_t)nmax * nmax;
    a.o_in_lds = need <= 64 * 1024;
    a.only_alive = only_alive; a.alive = h->alive;
    a.dbg = 0;
    if (!a.o_in_lds && !h->lu_ws)
        AFQ_HIP(h, hipMalloc(&h->lu_ws, sizeof(cplx) * (size_t)h->nw * nmax * nmax));
    a.ws = h->lu_ws;
    AFQ_LAUNCH(h, greens_kernel, dim3(h->nw), dim3(NTHR), a.o_in_lds ? need : 0, h->stream, a);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_greens(afq_handle *h, cplx *det_out) { return launch_greens(h, h->ghalf, det_out, 0); }
// det(psi^H phi) == det(phi^T conj(psi)) (transpose), so the same factorisation serves
// walkers/single_det.py:170-199
int k_overlap(afq_handle *h, cplx *det_out) { return launch_greens(h, nullptr, det_out, 0); }
// O^-1 of every walker and spin (+ determinant), no Ghalf: the discrete Hirsch propagator's inverse overlap
int k_inverse_overlap(afq_handle *h, cplx *oinv, cplx *det_out) {
    const int nmax = h->na > h->nb ? h->na : h->nb;
    if (!k_greens_big_supported(h) && (nmax > 45 || h->M > 4 * GS_KSMAX))
        AFQ_FAIL(h, AFQ_EUNSUPPORTED, "inverse overlaps: N <= 128 per spin (N <= 45: M <= 128)");
    return launch_greens(h, nullptr, det_out, 1, oinv);
}

// --------------------------------------------------------------------------
// force bias from the contraction output, per system
struct XbarArgs {
    int kind, flags, M, K, na, nb, nt, nw, nsplit, nq;
    double sqrt_dt, U;
    const cplx *vbias, *mf, *ghalf, *psi;
    const cplx *psicT;      // conj(psi)^T [nt, M] or null
    const cplx *gdiag;      // Hubbard: diag(G_s) partial sums [2 nw, gparts, M] left by the Ghalf GEMM, or null
    int gparts;
    cplx *xbar;
    int ndet;               // > 1: vbias holds ndet slices of det_stride elements, combined with detw
    long det_stride;
    const cplx *detw;       // [nw, ndet]
    // Hubbard with continuous fields: the kernel that makes the shifted fields also makes the HS potential's diagonal and
    // its degree-n Taylor factor per site (vhs_hubbard_kernel + exp_diag_factor_kernel of k_models.hip: two launches of
    // M nw elements each), hub_fac [nw, nv, M]; null otherwise
    cplx *hub_fac = nullptr;
    double hub_f = 0.0;     // sqrt(dt) sqrt(U) (charge decomposition: d = i f x) or sqrt(dt U) (spin: d = -+ f x)
    int hub_spin = 0, hub_order = 0;
};

__device__ inline cplx xbar_value(const XbarArgs &a, int w, int n) {
    cplx out = cmake(0.0, 0.0);
    if (a.flags & AFQ_PROP_FORCE_BIAS) {
        if (a.kind == AFQ_SYS_GENERIC) {
            // propagation/generic.py:150-152: -sqrt(dt) (i vbias - mf_shift)
            cplx v = cmake(0.0, 0.0);
            if (a.ndet > 1) {
                // propagation/generic.py:154-157 + walkers/multi_det.py:283-290:
                // vbias_n = sum_d w_d <V_n G_d> / sum_d w_d; <V_n G_d> = rchol_d^T vec(Ghalf_d)
                cplx num = cmake(0.0, 0.0), den = cmake(0.0, 0.0);
                for (int d = 0; d < a.ndet; ++d) {
                    const cplx wd = a.detw[(long)w * a.ndet + d];
                    if (wd.x == 0.0 && wd.y == 0.0) continue;     // skipped determinant (msd_combine_kernel): its Ghalf may hold anything
                    cplx x = cmake(0.0, 0.0);
                    const cplx *vb = a.vbias + (long)d * a.det_stride;
                    for (int b = 0; b < 2 * a.nsplit; ++b) x = cadd(x, vb[((long)b * a.nw + w) * a.K + n]);
                    cfma(num, wd, x);
                    den = cadd(den, wd);
                }
                v = cdiv(num, den);
            } else {
                // fixed summation order, four partial slices in flight at a time
                const int nb2 = 2 * a.nsplit;
                const cplx *vb = a.vbias + (long)w * a.K + n;
                const long bs = (long)a.nw * a.K;
                int b = 0;
                for (; b + 3 < nb2; b += 4) {
                    const cplx t0 = vb[b * bs], t1 = vb[(b + 1) * bs], t2 = vb[(b + 2) * bs], t3 = vb[(b + 3) * bs];
                    v = cadd(cadd(cadd(cadd(v, t0), t1), t2), t3);
                }
                for (; b < nb2; ++b) v = cadd(v, vb[b * bs]);
            }
            const cplx m = a.mf[n];
            out = cmake(-a.sqrt_dt * (-v.y - m.x), -a.sqrt_dt * (v.x - m.y));
        } else if (a.kind == AFQ_SYS_HUBBARD) {
            // diag of G_s = conj(psi_s) Ghalf_s at site n
            cplx g[2] = {cmake(0.0, 0.0), cmake(0.0, 0.0)};
            if (a.gdiag) {
                for (int s = 0; s < 2; ++s)
                    for (int pt = 0; pt < a.gparts; ++pt) g[s] = cadd(g[s], a.gdiag[((long)(2 * w + s) * a.gparts + pt) * a.M + n]);
            } else
            for (int s = 0; s < 2; ++s) {
                const int ns = s == 0 ? a.na : a.nb, off = s == 0 ? 0 : a.na;
                for (int i = 0; i < ns; ++i) {
                    // adjacent threads = adjacent sites n: both operands read along n
                    const cplx c = a.psicT ? a.psicT[(long)(off + i) * a.M + n] : cconj(a.psi[(long)n * a.nt + off + i]);
                    cfma(g[s], c, a.ghalf[((long)w * a.nt + off + i) * a.M + n]);
                }
            }
            const cplx m = a.mf[n];
            const double su = sqrt(a.U);
            cplx vb;
            if (a.flags & AFQ_PROP_HUBBARD_SPIN) {       // propagation/hubbard.py:472
                vb = cmake(su * (g[0].x - g[1].x), su * (g[0].y - g[1].y));
            } else {                                      // propagation/hubbard.py:406: i sqrt(U) (n_up + n_dn)
                const cplx t = cadd(g[0], g[1]);
                vb = cmake(-su * t.y, su * t.x);
            }
            out = cmake(-a.sqrt_dt * (vb.x - m.x), -a.sqrt_dt * (vb.y - m.y));
        } else {
            // UEG: propagation/planewave.py:76 (vbias filled by the sparse kernel)
            const cplx v = a.vbias[(long)w * a.K + n];
            out = cmake(-a.sqrt_dt * v.x, -a.sqrt_dt * v.y);
        }
    }
    return out;
}

__global__ void xbar_kernel(XbarArgs a) {
    const int w = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.K) return;
    a.xbar[(long)w * a.K + n] = xbar_value(a, w, n);
}

static XbarArgs xbar_args(afq_handle *h) {
    XbarArgs a;
    a.kind = h->kind; a.flags = h->flags; a.M = h->M; a.K = h->K; a.na = h->na; a.nb = h->nb;
    a.nt = h->nt; a.nw = h->nw; a.nsplit = h->fb_split; a.nq = h->nq;
    a.sqrt_dt = h->sqrt_dt; a.U = h->U;
    a.vbias = h->vbias; a.mf = h->mf_shift; a.ghalf = h->ghalf; a.psi = h->psi; a.xbar = h->xbar;
    a.psicT = (h->ndet <= 1 && h->psi_stride == 0) ? h->psicT : nullptr;
    a.gdiag = (h->kind == AFQ_SYS_HUBBARD && h->gdiag && h->gdiag_version == h->ghalf_version) ? h->gdiag : nullptr;
    a.gparts = h->gdiag_parts;
    a.ndet = h->ndet; a.detw = h->detw; a.det_stride = (long)2 * h->fb_split * h->nw * h->K;
    if (h->ndet > 1) a.vbias = h->vbias_all;
    // (the contraction with the determinant-averaged G left ONE set of partials, already weighted: k_force_bias_msd_gbar)
    if (h->ndet > 1 && h->msd_fb_gbar && h->kind == AFQ_SYS_GENERIC) a.ndet = 1;
    return a;
}

int k_xbar(afq_handle *h) {
    const XbarArgs a = xbar_args(h);
    AFQ_LAUNCH(h, xbar_kernel, dim3((h->K + 127) / 128, h->nw), dim3(128), 0, h->stream, a);
    AFQ_POST(h);
    return AFQ_OK;
}

// --------------------------------------------------------------------------
// multi-determinant trial: weights and weighted averages (walkers/multi_det.py:194-229,135-162)
// skip_small (Green's function, walkers/multi_det.py:209,218): a determinant whose overlap with the walker is below 1e-16 is
// left out -- the reference `continue`s past it, keeping whatever Gi / weight the walker object held before (zeros for a fresh
// walker); here it gets the weight 0 and every consumer of the weights (force bias, energy) passes over it, i.e. the
// reference's result for a fresh walker.  A singular overlap matrix may leave NaN as its determinant (0 x inf behind a zero
// pivot): that IS a zero overlap, in the Green's function and in calc_overlap (multi_det.py:135-162, which skips nothing);
// every such event is counted (afq_counters_ext [6]) so that a NaN from a genuine numerical failure does not pass unseen.
// The reference tests the alpha determinant first (:209) and the product of both second (:218): both tests are applied, the
// first on detd_a (the alpha determinant alone, written by the Green's function kernels beside the product).
__global__ void msd_combine_kernel(const cplx *detd, const cplx *detd_a, const cplx *coeffs, cplx *detw, cplx *det_out, int nw,
                                   int ndet, int skip_small, unsigned long long *counters) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    cplx tot = cmake(0.0, 0.0);
    for (int d = 0; d < ndet; ++d) {
        cplx dd = detd[(long)d * nw + w];
        const double mag = hypot(dd.x, dd.y);
        bool skip = skip_small && mag < 1e-16;
        if (skip_small && detd_a) {
            const cplx da = detd_a[(long)d * nw + w];
            const double ma = hypot(da.x, da.y);
            skip = skip || ma < 1e-16;
        }
        if (!(mag == mag)) { skip = true; if (counters) atomicAdd(&counters[6], 1ull); }
        if (skip) dd = cmake(0.0, 0.0);
        const cplx wd = cmul(cconj(coeffs[d]), dd);
        detw[(long)w * ndet + d] = wd;
        tot = cadd(tot, wd);
    }
    det_out[w] = tot;
}

int k_msd_combine(afq_handle *h, cplx *det_out, bool skip_small) {
    AFQ_LAUNCH(h, msd_combine_kernel, dim3((h->nw + 127) / 128), dim3(128), 0, h->stream, h->detd,
                       skip_small ? h->detd_a : nullptr, h->coeffs, h->detw, det_out, h->nw, h->ndet, skip_small ? 1 : 0, h->counters);
    AFQ_POST(h);
    return AFQ_OK;
}

// estimators/mixed.py:439-448: E = sum_d w_d E[G_d] / sum_d w_d, component-wise
__global__ void msd_energy_combine_kernel(const cplx *energy_all, const cplx *detw, cplx *energy, int nw,
                                          int ndet) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    cplx num[3] = {cmake(0.0, 0.0), cmake(0.0, 0.0), cmake(0.0, 0.0)}, den = cmake(0.0, 0.0);
    for (int d = 0; d < ndet; ++d) {
        const cplx wd = detw[(long)w * ndet + d];
        if (wd.x == 0.0 && wd.y == 0.0) continue;              // skipped determinant: its energy may be anything
        for (int c = 0; c < 3; ++c) cfma(num[c], wd, energy_all[((long)d * nw + w) * 3 + c]);
        den = cadd(den, wd);
    }
    for (int c = 0; c < 3; ++c) energy[(long)w * 3 + c] = cdiv(num[c], den);
}

int k_msd_energy_combine(afq_handle *h) {
    AFQ_LAUNCH(h, msd_energy_combine_kernel, dim3((h->nw + 127) / 128), dim3(128), 0, h->stream,
                       h->energy_all, h->detw, h->energy, h->nw, h->ndet);
    AFQ_POST(h);
    return AFQ_OK;
}

#include "philox.h"

// --------------------------------------------------------------------------
__device__ inline double block_sum(double v, double *red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

// propagation/continuous.py:140-158: clip, shift, constant factors
// FUSED: the force bias of (w, n) is evaluated here from the contraction output (xbar_value) instead
// of being read back from a separate xbar_kernel launch
// rng.on: the auxiliary fields are drawn here from the device stream instead of being read from xi (the same
// numbers rng_normal_kernel writes: element e = w K + n is member e & 1 of Philox pair e >> 1), and the alive
// flag of the step (qmc/afqmc.py:232) is set here too -- one launch less per step.

template <bool FUSED>
__global__ __launch_bounds__(NTHR) void fields_kernel(int K, double sqrt_dt, const double *xi, cplx *xbar,
                                                      const cplx *mf, cplx *xs, cplx *cmf, cplx *cfb,
                                                      unsigned long long *counters, const int *alive, XbarArgs xa,
                                                      FieldRng rng) {
    __shared__ double red[NTHR / 64][8];
    const int w = blockIdx.x;
#ifdef AFQ_TUNING
    if (rng.dbg & 8) return;
#endif
    if (rng.on) {
        const bool live = fabs(rng.weight[w]) > 1e-8;
        if (threadIdx.x == 0) rng.alive_out[w] = live ? 1 : 0;
        if (!live) return;
    } else if (alive && !alive[w]) return;
    // sums: mean-field shift (re, im), xi . xbar (re, im), xbar . xbar (re, im), clipped count
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    // (the force bias and the mean-field shift of a field are REQUESTED before its Philox arithmetic and used behind it:
    //  the kernel is one short latency chain per work-group, and the loads now fly under the ~1 us of integer / log / sincos)
    auto element = [&](const long e, const int n, const double xdev, cplx b, const cplx mm) {
        const double ab = hypot(b.x, b.y);
        if (ab > 1.0) { b.x /= ab; b.y /= ab; acc[6] += 1.0; }
        const double x = rng.on ? xdev : xi[e];
        const cplx sft = cmake(x - b.x, -b.y);
#ifdef AFQ_TUNING
        if (!(rng.dbg & 4))
#endif
        {
        xbar[e] = b;
        xs[e] = sft;
        }
        if (FUSED && xa.hub_fac) {
            // propagation/hubbard.py:409-413 / :475-480, continuous.py:104-107 with a diagonal potential: one factor per site
            auto taylor = [&](const cplx d) {
                cplx acc = cmake(1.0, 0.0), t = acc;
                for (int k = 1; k <= xa.hub_order; ++k) {
                    t = cmul(d, t);
                    t = cmake(t.x / k, t.y / k);
                    acc = cadd(acc, t);
                }
                return acc;
            };
            if (!xa.hub_spin) xa.hub_fac[(long)w * K + n] = taylor(cmake(-xa.hub_f * sft.y, xa.hub_f * sft.x));
            else {
                xa.hub_fac[((long)w * 2 + 0) * K + n] = taylor(cmake(-xa.hub_f * sft.x, -xa.hub_f * sft.y));
                xa.hub_fac[((long)w * 2 + 1) * K + n] = taylor(cmake(xa.hub_f * sft.x, xa.hub_f * sft.y));
            }
        }
        acc[0] += sft.x * mm.x - sft.y * mm.y;
        acc[1] += sft.x * mm.y + sft.y * mm.x;
        acc[2] += x * b.x; acc[3] += x * b.y;
        acc[4] += b.x * b.x - b.y * b.y;
        acc[5] += 2.0 * b.x * b.y;
    };
    const long e0 = (long)w * K;
    if (K <= NTHR) {
        // a thread per field (a Philox pair is then generated by both of its threads): with few fields the evaluation of the
        // force bias (xbar_value: a contraction per element for the lattice models) wants every thread of the work-group
        const int n = threadIdx.x;
        if (n < K) {
            const cplx b = FUSED ? xbar_value(xa, w, n) : xbar[e0 + n];
            const cplx mm = mf[n];
            double xn[2] = {0.0, 0.0};
            if (rng.on) philox_normal_pair((e0 + n) >> 1, rng.seed, rng.stream, rng.counter, xn[0], xn[1]);
            element(e0 + n, n, xn[(e0 + n) & 1], b, mm);
        }
    } else {
        // a thread takes the two members of one Philox pair (elements 2 p, 2 p + 1 of the stream: consecutive fields of
        // this walker, or its first / last field alone when the walker's K fields start or end inside a pair), so that a
        // pair is generated once -- with a thread per field every pair was generated twice and one normal of each thrown away
        // Generic Hamiltonian, one determinant, force bias on (the C3 step): the 2 x 2 nsplit split-K partial sums of the pair's
        // two fields and their mean-field shifts are loaded in ONE flight, no branch between them -- the partials come from the
        // other XCDs' force-bias work-groups (16 MB at C3), and this kernel is their round trips: 5.3 of its 10.7 us with the
        // loads of one field at a time, four in flight (round 6; the sums keep xbar_value's order: bit-identical)
        const bool flight = FUSED && xa.kind == AFQ_SYS_GENERIC && xa.ndet <= 1 && (xa.flags & AFQ_PROP_FORCE_BIAS) && !xa.hub_fac;
        for (long pr = (e0 >> 1) + threadIdx.x; pr <= ((e0 + K - 1) >> 1); pr += NTHR) {
            cplx bb[2] = {cmake(0.0, 0.0), cmake(0.0, 0.0)}, mm[2] = {cmake(0.0, 0.0), cmake(0.0, 0.0)};
            if (flight) {
                int nn[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int n = (int)(2 * pr + m - e0);
                    nn[m] = n < 0 ? 0 : n >= K ? K - 1 : n;          // (a pair that straddles the walker: a valid address, value unused)
                }
                const int nb2 = 2 * xa.nsplit;
                const long bs = (long)xa.nw * K;
                const cplx *vb0 = xa.vbias + (long)w * K + nn[0], *vb1 = xa.vbias + (long)w * K + nn[1];
                mm[0] = mf[nn[0]]; mm[1] = mf[nn[1]];
                cplx v0 = cmake(0.0, 0.0), v1 = cmake(0.0, 0.0);
                int b = 0;
                for (; b + 7 < nb2; b += 8) {
                    cplx t0[8], t1[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { t0[j] = vb0[(b + j) * bs]; t1[j] = vb1[(b + j) * bs]; }
#pragma unroll
                    for (int j = 0; j < 8; ++j) { v0 = cadd(v0, t0[j]); v1 = cadd(v1, t1[j]); }
                }
                for (; b + 3 < nb2; b += 4) {
                    cplx t0[4], t1[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { t0[j] = vb0[(b + j) * bs]; t1[j] = vb1[(b + j) * bs]; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v0 = cadd(v0, t0[j]); v1 = cadd(v1, t1[j]); }
                }
                for (; b < nb2; ++b) { v0 = cadd(v0, vb0[b * bs]); v1 = cadd(v1, vb1[b * bs]); }
                // propagation/generic.py:150-152 (xbar_value): -sqrt(dt) (i vbias - mf_shift)
                bb[0] = cmake(-xa.sqrt_dt * (-v0.y - mm[0].x), -xa.sqrt_dt * (v0.x - mm[0].y));
                bb[1] = cmake(-xa.sqrt_dt * (-v1.y - mm[1].x), -xa.sqrt_dt * (v1.x - mm[1].y));
            } else
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const long e = 2 * pr + m;
                const int n = (int)(e - e0);
#ifdef AFQ_TUNING
                if (rng.dbg & 2) { if (n >= 0 && n < K) { bb[m] = cmake(0.01 * n, 0.02); mm[m] = cmake(0.5, 0.1); } continue; }
#endif
                if (n >= 0 && n < K) { bb[m] = FUSED ? xbar_value(xa, w, n) : xbar[e]; mm[m] = mf[n]; }
            }
            double xn[2] = {0.0, 0.0};
#ifdef AFQ_TUNING
            if (rng.dbg & 1) { xn[0] = 0.3 + 1e-3 * threadIdx.x; xn[1] = -0.2; } else
#endif
            if (rng.on) philox_normal_pair(pr, rng.seed, rng.stream, rng.counter, xn[0], xn[1]);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const long e = 2 * pr + m;
                const int n = (int)(e - e0);
                if (n >= 0 && n < K) element(e, n, xn[m], bb[m], mm[m]);
            }
        }
    }
#ifdef AFQ_TUNING
    if (rng.dbg & 16) { if (threadIdx.x == 0) { cmf[w] = cmake(acc[0], acc[1]); cfb[w] = cmake(acc[2], acc[3] + acc[4] + acc[5] + acc[6]); } return; }
#endif
    // one reduction for all seven sums: wave shuffles, one barrier
#pragma unroll
    for (int q = 0; q < 7; ++q)
        for (int off = 32; off > 0; off >>= 1) acc[q] += __shfl_down(acc[q], off);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 7; ++q) red[threadIdx.x >> 6][q] = acc[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            t[q] = 0.0;
            for (int i = 0; i < NTHR / 64; ++i) t[q] += red[i][q];
        }
        cmf[w] = cmake(-sqrt_dt * t[0], -sqrt_dt * t[1]);
        cfb[w] = cmake(t[2] - 0.5 * t[4], t[3] - 0.5 * t[5]);
        if (t[6] > 0 && counters) atomicAdd(&counters[0], (unsigned long long)t[6]);
    }
}

int k_fields(afq_handle *h) {
    AFQ_LAUNCH(h, fields_kernel<false>, dim3(h->nw), dim3(NTHR), 0, h->stream, h->K, h->sqrt_dt, h->xi, h->xbar,
                       h->mf_shift, h->xs, h->cmf, h->cfb, h->counters, h->alive, XbarArgs(), FieldRng());
    AFQ_POST(h);
    return AFQ_OK;
}

// force bias from the contraction output + clip + shift in one launch (the step's hot path)
int k_xbar_fields(afq_handle *h, cplx *hubbard_factors) {
    FieldRng rng = FieldRng();
#ifdef AFQ_TUNING
    rng.dbg = AFQ_KNOB_INT("AFQ_FIELDS_DBG", 0);
#endif
    if (h->rng_inline) {
        rng.on = 1; rng.seed = h->rng_seed; rng.stream = h->rng_stream; rng.counter = h->rng_inline_counter;
        rng.weight = h->weight; rng.alive_out = h->alive;
        h->rng_inline = false;
    }
    XbarArgs xa = xbar_args(h);
    if (hubbard_factors) {
        const bool spin = (h->flags & AFQ_PROP_HUBBARD_SPIN) != 0;
        xa.hub_fac = hubbard_factors; xa.hub_spin = spin ? 1 : 0; xa.hub_order = h->exp_order;
        xa.hub_f = spin ? sqrt(h->dt * h->U) : h->sqrt_dt * sqrt(h->U);
    }
    AFQ_LAUNCH(h, fields_kernel<true>, dim3(h->nw), dim3(NTHR), 0, h->stream, h->K, h->sqrt_dt, h->xi, h->xbar,
                       h->mf_shift, h->xs, h->cmf, h->cfb, h->counters, h->alive, xa, rng);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_fields_explicit(afq_handle *h, const double *xi_d, const cplx *xbar_d, cplx *xs_d, cplx *cmf_d,
                      cplx *cfb_d) {
    AFQ_LAUNCH(h, fields_kernel<false>, dim3(h->nw), dim3(NTHR), 0, h->stream, h->K, h->sqrt_dt, xi_d,
                       (cplx *)xbar_d, h->mf_shift, xs_d, cmf_d, cfb_d, (unsigned long long *)nullptr,
                       (const int *)nullptr, XbarArgs(), FieldRng());
    AFQ_POST(h);
    return AFQ_OK;
}

// FieldConfig.update: append this step's shifted fields to the walker's history
__global__ void bp_push_kernel(const cplx *xs, cplx *hist, int *bp_n, const int *flag, int K, int nbp) {
    const int w = blockIdx.x;
    if (!flag[w]) return;
    const int n = bp_n[w];
    if (n < nbp)
        for (int k = threadIdx.x; k < K; k += blockDim.x) hist[((long)w * nbp + n) * K + k] = xs[(long)w * K + k];
    __syncthreads();
    if (threadIdx.x == 0 && n < nbp) bp_n[w] = n + 1;
}

// fields of back-propagation step i (most recent first): B(x)^H = B(-conj(x)) for real symmetric L_n
__global__ void bp_fields_kernel(const cplx *hist, const int *bp_n, cplx *xs, int *alive, int K, int nbp, int i) {
    const int w = blockIdx.x;
    const int n = bp_n[w];
    const bool on = i < n;
    if (threadIdx.x == 0) alive[w] = on ? 1 : 0;
    if (!on) return;
    const cplx *src = hist + ((long)w * nbp + (n - 1 - i)) * K;
    for (int k = threadIdx.x; k < K; k += blockDim.x) xs[(long)w * K + k] = cmake(-src[k].x, src[k].y);
}

__global__ void bp_init_kernel(const cplx *phi0, cplx *phi_bp, long per, int nw) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < per * nw) phi_bp[i] = phi0[i % per];
}

__global__ void conj_copy_kernel(const cplx *src, cplx *dst, long n) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) dst[i] = cconj(src[i]);
}

__global__ void conj_transpose_kernel(const cplx *A, cplx *At, int M) {      // per spin: At = A^H
    const int s = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * M) return;
    const int r = i / M, c = i % M;
    At[(long)s * M * M + c * M + r] = cconj(A[(long)s * M * M + i]);
}

// estimators/back_propagation.py:187-207: est[3] += w, est[4:] += w G_bp, w = weight (x restored factor)
__global__ void bp_accumulate_kernel(const cplx *G, const double *weight, const double *bp_cos, const cplx *bp_ph,
                                     cplx *est, int nw, long gsz, int restore, const cplx *energy) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e > gsz + 3) return;
    if (e > gsz && !energy) return;
    cplx acc = cmake(0.0, 0.0);
    for (int w = 0; w < nw; ++w) {
        cplx wt = cmake(weight[w], 0.0);
        if (restore == 1) wt = cmul(wt, bp_ph[w]);                                       // BP-PRes (partial)
        else if (restore == 2) wt = cmul(wt, cmake(bp_ph[w].x / bp_cos[w], bp_ph[w].y / bp_cos[w]));   // full
        if (e == gsz) acc = cadd(acc, wt);
        else if (e > gsz) cfma(acc, wt, energy[3 * w + (e - gsz - 1)]);      // estimates[:nreg] += weight * energies
        else cfma(acc, wt, G[(long)w * gsz + e]);
    }
    if (e == gsz) est[3] = cadd(est[3], acc);
    else if (e > gsz) est[e - gsz - 1] = cadd(est[e - gsz - 1], acc);
    else est[4 + e] = cadd(est[4 + e], acc);
}

// keep_mod > 0 (discrete fields, bp_n counts single fields): FieldConfig.reset zeroes `step` but not the position
// `ib` inside an unfinished configuration (walkers/stack.py:124-127)
__global__ void bp_reset_kernel(int *bp_n, double *bp_cos, cplx *bp_ph, int nw, int keep_mod) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nw) { bp_n[w] = keep_mod > 0 ? bp_n[w] % keep_mod : 0; bp_cos[w] = 1.0; bp_ph[w] = cmake(1.0, 0.0); }
}

int k_bp_push(afq_handle *h) {
    AFQ_LAUNCH(h, bp_push_kernel, dim3(h->nw), dim3(128), 0, h->stream, h->xs, h->bp_hist, h->bp_n, h->bp_flag,
                       h->K, h->nbp);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_bp_fields(afq_handle *h, int i) {
    AFQ_LAUNCH(h, bp_fields_kernel, dim3(h->nw), dim3(128), 0, h->stream, h->bp_hist, h->bp_n, h->xs, h->alive,
                       h->K, h->nbp, i);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_bp_init(afq_handle *h, const cplx *phi0_dev) {
    const long per = (long)h->M * h->nt, n = per * h->nw;
    AFQ_LAUNCH(h, bp_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, phi0_dev, h->phi_bp,
                       per, h->nw);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_conj_copy(afq_handle *h, const cplx *src, cplx *dst, long n) {
    AFQ_LAUNCH(h, conj_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, src, dst, n);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_conj_transpose(afq_handle *h, const cplx *A, cplx *At) {
    AFQ_LAUNCH(h, conj_transpose_kernel, dim3((h->M * h->M + 255) / 256, 2), dim3(256), 0, h->stream, A, At,
                       h->M);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_bp_accumulate(afq_handle *h, int restore, int with_energy) {
    const long gsz = 2L * h->M * h->M;
    AFQ_LAUNCH(h, bp_accumulate_kernel, dim3((unsigned)((gsz + 4 + 127) / 128)), dim3(128), 0, h->stream, h->G,
                       h->weight, h->bp_cos, h->bp_ph, h->bp_est, h->nw, gsz, restore,
                       with_energy ? h->energy : (const cplx *)nullptr);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_bp_reset(afq_handle *h, bool first) {
    AFQ_LAUNCH(h, bp_reset_kernel, dim3((h->nw + 127) / 128), dim3(128), 0, h->stream, h->bp_n, h->bp_cos,
                       h->bp_ph, h->nw, (h->hirsch && !first) ? h->K : 0);
    AFQ_POST(h);
    return AFQ_OK;
}


int k_update_weight(afq_handle *h, cplx eshift) {
    if (h->fuse_weight_done) { h->fuse_weight_done = false; return AFQ_OK; }   // rode on the Green's function kernel
    const WeightArgs a = weight_args(h, eshift);
    AFQ_LAUNCH(h, weight_kernel, dim3((h->nw + 127) / 128), dim3(128), 0, h->stream, a);
    AFQ_POST(h);
    return AFQ_OK;
}

static WeightArgs weight_args(afq_handle *h, cplx eshift) {
    WeightArgs a;
    a.nw = h->nw; a.flags = h->flags; a.dt = h->dt; a.eshift = eshift; a.alive = h->alive;
    a.ovlp_old = h->ovlp_old; a.ovlp_new = h->ovlp_new; a.cmf = h->cmf; a.cfb = h->cfb;
    a.weight = h->weight; a.ot = h->ot; a.ehyb = h->ehyb; a.phase = h->phase; a.counters = h->counters;
    a.eloc = h->eloc; a.energy = h->energy;
    a.bp_flag = h->nbp > 0 ? h->bp_flag : nullptr; a.bp_cos = h->bp_cos; a.bp_ph = h->bp_ph;
    a.cap_frac = h->cap_frac; a.cap_total = h->cap_total; a.cap_total_dev = h->scal;
    a.ot_scale = h->log_shift_on ? exp(-h->log_shift) : 1.0;
    a.est_acc = h->fuse_est_req ? h->est_acc : nullptr; a.unscaled = h->unscaled;
    return a;
}

// --------------------------------------------------------------------------
// Re-orthogonalisation: classical Gram-Schmidt applied twice per column.  With
// R_jj = ||v|| > 0 this is the unique QR with positive diagonal, i.e. exactly
// what the reference obtains from LAPACK QR followed by the sign fix
// (walkers/single_det.py:225-242); detR = prod_j R_jj over both spins.
struct QrArgs {
    int M, na, nb, nt, nw, in_lds, flags;
    cplx *phi;
    cplx *ws;           // global panel [nw, nmax, M] when it does not fit LDS
    double *detR, *weight;
    cplx *ot;
    const int *only;    // when set: redo only the walkers flagged by the Cholesky-QR path
    cplx *keep;         // cached overlap of a Green's function that stays valid across the QR (ovlp /= det R), or null
};

__global__ __launch_bounds__(NTHR) void reortho_kernel(QrArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ cplx coef[256];
    __shared__ double red[8];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.only && !a.only[w]) return;
    const int M = a.M, nt = a.nt;
    cplx *phi = a.phi + (long)w * M * nt;
    const int nmax = a.na > a.nb ? a.na : a.nb;
    cplx *Q = a.in_lds ? (cplx *)smem : a.ws + (long)w * nmax * M;     // Q[i][p], column i contiguous
    double logdet = 0.0;
    for (int s = 0; s < 2; ++s) {
        const int n = s == 0 ? a.na : a.nb, off = s == 0 ? 0 : a.na;
        if (n == 0) continue;
        for (int e = tid; e < n * M; e += NTHR) {
            const int p = e / n, i = e % n;
            Q[(long)i * M + p] = phi[(long)p * nt + off + i];
        }
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            cplx *v = Q + (long)j * M;
            for (int pass = 0; pass < 2 && j > 0; ++pass) {
                for (int i = wave; i < j; i += NTHR / 64) {
                    const cplx *q = Q + (long)i * M;
                    double sr = 0, si = 0;
                    for (int p = lane; p < M; p += 64) {
                        const cplx x = q[p], y = v[p];      // conj(x) * y
                        sr += x.x * y.x + x.y * y.y;
                        si += x.x * y.y - x.y * y.x;
                    }
                    for (int o = 32; o > 0; o >>= 1) { sr += __shfl_down(sr, o); si += __shfl_down(si, o); }
                    if (lane == 0) coef[i] = cmake(sr, si);
                }
                __syncthreads();
                for (int p = tid; p < M; p += NTHR) {
                    cplx y = v[p];
                    for (int i = 0; i < j; ++i) {
                        const cplx c = coef[i], x = Q[(long)i * M + p];
                        y.x = fma(-c.x, x.x, y.x); y.x = fma(c.y, x.y, y.x);
                        y.y = fma(-c.x, x.y, y.y); y.y = fma(-c.y, x.x, y.y);
                    }
                    v[p] = y;
                }
                __syncthreads();
            }
            double nn = 0.0;
            for (int p = tid; p < M; p += NTHR) nn += cabs2(v[p]);
            nn = block_sum(nn, red);
            const double r = sqrt(nn), inv = 1.0 / r;
            logdet += log(r);
            for (int p = tid; p < M; p += NTHR) v[p] = cscale(v[p], inv);
            __syncthreads();
        }
        for (int e = tid; e < n * M; e += NTHR) {
            const int p = e / n, i = e % n;
            phi[(long)p * nt + off + i] = Q[(long)i * M + p];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double d = exp(logdet);
        a.detR[w] = d;
        a.ot[w] = cmake(a.ot[w].x / d, a.ot[w].y / d);           // single_det.py:253
        if (a.keep) a.keep[w] = cmake(a.keep[w].x / d, a.keep[w].y / d);
        if (a.flags & AFQ_PROP_FREE_PROJECTION) a.weight[w] *= d;  // walkers/handler.py:178-181
    }
}

// --------------------------------------------------------------------------
// Cholesky-QR2 of one walker in ONE work-group (N <= 32 electrons per spin, the walker and both Gram matrices in LDS).
// The GEMM-engine version (k_reortho_big: Gram, Cholesky, Q = phi R^-1 as three launches per pass, twice) spends
// ~160 us per re-orthogonalisation at C3 sizes on six latency-bound launches of 25 us each for ~5 us of arithmetic; here
// the passes run back to back out of LDS:
//   Gram  S_s = phi_s^H phi_s          fp64 MFMA, waves 0-3 spin up, 4-7 spin down, one 16x16 tile per wave
//   Chol  T = R^-1 (S = R^H R)         one wave per spin, register resident (chol_block8, gj_wave.h)
//   Q     phi_s <- phi_s T             fp64 MFMA, in place: a wave owns a 16-row tile, whose fragments it holds in registers
// Same algorithm, same fallback contract as k_reortho_big: a walker whose Cholesky breaks down (fail[w] = 1) is left
// untouched for the Gram-Schmidt kernel.  (reference: scipy.linalg.qr + sign fix, walkers/single_det.py:228-255)
#define RF_LD 33
struct RfArgs {
    int M, na, nb, nt, nw, fp;
    cplx *phi;
    double *detR, *weight;
    cplx *ot;
    int *fail;
    cplx *keep;          // cached overlap of a Green's function that stays valid across the QR (ovlp /= det R), or null
};

#ifdef AFQ_TUNING
__device__ unsigned long long *afq_rf_ts = nullptr;
#define RF_STAMP(i) do { if (afq_rf_ts && blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); afq_rf_ts[i] = t_; } } while (0)
#else
#define RF_STAMP(i)
#endif
// RJ: registers per lane of the Cholesky wave (ceil(nmax / 2) rounded up to 4 / 8 / 13 / 16; one instantiation each: a kernel that
// carries all of them spills)
template <int RJ>
__global__ __launch_bounds__(512) void reortho_fused_kernel(RfArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ cplx rowk_s[2][32];
    __shared__ double piv_s[2][32];
    __shared__ double logd_s[2];
    __shared__ int bad_s[2];
    const int w = blockIdx.x, tid = threadIdx.x;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 8), wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
    const int lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int M = a.M, nt = a.nt;
    const int n = g == 0 ? a.na : a.nb, off = g == 0 ? 0 : a.na;
    cplx *phi_l = (cplx *)smem;                              // [M, nt]
    cplx *S = phi_l + (long)M * nt + (long)g * (32 * RF_LD);   // [32, RF_LD] Gram matrix, then T^T, of this spin (row stride 33: no bank conflicts for a lane per row)
    cplx *phi_g = a.phi + (long)w * M * nt;
    RF_STAMP(0);
    {
        // the walker by LDS-DMA: 1 KB per wave and instruction, all requests of a wave in flight at once
        const unsigned total = (unsigned)(M * nt) * 16u;
        const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
        for (unsigned b0 = (unsigned)wave8 * 1024u; b0 < total; b0 += 8 * 1024u) {
            const unsigned bo = b0 + (unsigned)lane * 16u;
            if (bo < total) glds16((const char *)phi_g + bo, (char *)phi_l + b0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (tid < 2) { logd_s[tid] = 0.0; bad_s[tid] = 0; }
    __syncthreads();
    RF_STAMP(1);
    const int nt16 = (n + 15) >> 4, mt16 = (M + 15) >> 4, nks = (M + 3) >> 2, nks3 = (n + 3) >> 2;
    for (int pass = 0; pass < 2; ++pass) {
        // ---- Gram matrix
        for (int t = wave; t < nt16 * nt16; t += 4) {
            const int ti = t / nt16, tj = t % nt16;
            const int ia = ti * 16 + lr, jb = tj * 16 + lr;
            const int iac = ia < n ? ia : n - 1, jbc = jb < n ? jb : n - 1;
            d4_t accR = {0, 0, 0, 0}, accI = {0, 0, 0, 0};
            // Rows / columns of the tile beyond n read a clamped (valid) column and produce numbers that are never stored: an
            // MFMA output element depends on its own A row and B column only.  Only the CONTRACTION index must not run past
            // M: whole steps of four are unconditional loads (no select for the compiler to turn into a branch around the
            // load), the last partial step selects.
            const cplx *xp = phi_l + lk * nt + off + iac, *yp = phi_l + lk * nt + off + jbc;
            const int nfull = M >> 2;
            // conj(x) * y by three multiplications: P1 = xr yr, P2 = xi yi, P3 = (xr - xi)(yr + yi); re = P1 + P2,
            // im = P3 - P1 + P2
            d4_t acc3 = {0, 0, 0, 0};
            // pairs of k-steps, the fragments of the next pair read from LDS ahead of the six MFMAs of this one (one k-step per
            // loop iteration put every pair of LDS reads directly in front of the MFMAs that wait for them)
            const int ngrp = nfull >> 1;
            cplx xa[2], ya[2], xb[2], yb[2];
            auto rd = [&](cplx (&x)[2], cplx (&y)[2], const int gq) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 2; ++u) { x[u] = xp[(gq * 2 + u) * 4 * nt]; y[u] = yp[(gq * 2 + u) * 4 * nt]; }
            };
            auto mf = [&](const cplx (&x)[2], const cplx (&y)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    accR = mfma16(x[u].x, y[u].x, accR);
                    accI = mfma16(x[u].y, y[u].y, accI);
                    acc3 = mfma16(x[u].x - x[u].y, y[u].x + y[u].y, acc3);
                }
            };
            if (ngrp > 0) rd(xa, ya, 0);
            for (int gq = 0; gq < ngrp; gq += 2) {
                if (gq + 1 < ngrp) rd(xb, yb, gq + 1);
                mf(xa, ya);
                if (gq + 1 < ngrp) {
                    if (gq + 2 < ngrp) rd(xa, ya, gq + 2);
                    mf(xb, yb);
                }
            }
            for (int ks = ngrp * 2; ks < nfull; ++ks) {
                const cplx x = xp[ks * 4 * nt], y = yp[ks * 4 * nt];
                accR = mfma16(x.x, y.x, accR);
                accI = mfma16(x.y, y.y, accI);
                acc3 = mfma16(x.x - x.y, y.x + y.y, acc3);
            }
            RF_STAMP(9 + pass);
            if (nfull < nks) {
                const int p = nfull * 4 + lk, pc = p < M ? p : M - 1;
                cplx x = phi_l[pc * nt + off + iac], y = phi_l[pc * nt + off + jbc];
                if (p >= M) { x = cmake(0.0, 0.0); y = cmake(0.0, 0.0); }
                accR = mfma16(x.x, y.x, accR);
                accI = mfma16(x.y, y.y, accI);
                acc3 = mfma16(x.x - x.y, y.x + y.y, acc3);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double p1 = accR[r], p2 = accI[r];
                accR[r] = p1 + p2; accI[r] = acc3[r] - p1 + p2;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + lk + 4 * r, j = tj * 16 + lr;
                if (i < n && j < n) S[i * RF_LD + j] = cmake(accR[r], accI[r]);
            }
        }
        RF_STAMP(11 + pass);
        __syncthreads();
        RF_STAMP(2 + 3 * pass);
        // ---- inverse Cholesky factor: T^T[r][c] = conj(Ltilde^-1[r][c]) / sqrt(D_r) (see chol_small_kernel)
        if (wave == g && n > 0) {                            // (spin up on SIMD 0, spin down on SIMD 1: not both on one)
            bool bad = false;
            if (lane < 32) piv_s[g][lane] = 1.0;
            chol_wave_rj<RJ>(S, RF_LD, S, RF_LD, n, lane, rowk_s[g], piv_s[g], bad);
            double l = lane < n ? log(piv_s[g][lane & 31]) : 0.0;
            if (lane >= 32) l = 0.0;
            for (int o = 16; o > 0; o >>= 1) l += __shfl_down(l, o);
            // a pivot that is not a positive finite number: breakdown
            const unsigned long long anybad = __ballot(bad || !(l == l));
            if (lane == 0) {
                logd_s[g] += 0.5 * l;
                if (anybad) bad_s[g] = 1;
            }
        }
        __syncthreads();
        RF_STAMP(3 + 3 * pass);
        if (bad_s[0] | bad_s[1]) {                           // leave the walker to the Gram-Schmidt kernel
            if (tid == 0) a.fail[w] = 1;
            return;
        }
        // ---- Q = phi T in place: a wave takes whole 16-row tiles (all their columns), the row fragments in registers.
        // The fragments of T are the same for every row tile: read once per pass, all loads unconditional on clamped
        // indices (a clamped entry meets a zero of the row fragment or a column that is not stored), trip counts from RJ --
        // no branch between an LDS read and the MFMAs that wait for it.  T is upper triangular: the column tile j < 16
        // contracts k < 16 only.
        if (n > 0) {
            constexpr int KSN = (2 * RJ + 3) / 4 > 8 ? 8 : (2 * RJ + 3) / 4, TJN = 2 * RJ > 16 ? 2 : 1;
            cplx yf[TJN][KSN];
#pragma unroll
            for (int tj = 0; tj < TJN; ++tj)
#pragma unroll
                for (int ks = 0; ks < KSN; ++ks) {
                    const int k = ks * 4 + lk, kc = k < n ? k : n - 1;
                    const int jq = tj * 16 + lr, jc = jq < n ? jq : n - 1;
                    yf[tj][ks] = S[jc * RF_LD + kc];         // T[k][j] = T^T[j][k]
                }
            auto load_x = [&](cplx (&xf)[KSN], const int ti) __attribute__((always_inline)) {
                const int pa = ti * 16 + lr, pac = pa < M ? pa : M - 1;
#pragma unroll
                for (int ks = 0; ks < KSN; ++ks) {
                    const int k = ks * 4 + lk, kc = k < n ? k : n - 1;
                    const cplx x = phi_l[pac * nt + off + kc];
                    xf[ks] = (k < n && pa < M) ? x : cmake(0.0, 0.0);
                }
            };
            auto mul_store = [&](const cplx (&xf)[KSN], const int ti) __attribute__((always_inline)) {
                d4_t p1[TJN], p2[TJN], p3[TJN];
#pragma unroll
                for (int tj = 0; tj < TJN; ++tj) {
                    p1[tj] = (d4_t){0, 0, 0, 0}; p2[tj] = (d4_t){0, 0, 0, 0}; p3[tj] = (d4_t){0, 0, 0, 0};
                }
#pragma unroll
                for (int ks = 0; ks < KSN; ++ks) {
                    const cplx x = xf[ks];
                    const double xs = x.x + x.y;
#pragma unroll
                    for (int tj = 0; tj < TJN; ++tj) {
                        if (tj == 0 && TJN == 2 && ks >= 4) continue;      // (compile time) T[k][j] = 0 for k >= 16 > j
                        const cplx y = yf[tj][ks];
                        p1[tj] = mfma16(x.x, y.x, p1[tj]);
                        p2[tj] = mfma16(x.y, y.y, p2[tj]);
                        p3[tj] = mfma16(xs, y.x + y.y, p3[tj]);
                    }
                }
#pragma unroll
                for (int tj = 0; tj < TJN; ++tj) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int p = ti * 16 + lk + 4 * r, jq = tj * 16 + lr;
                        if (p < M && jq < n)
                            phi_l[p * nt + off + jq] = cmake(p1[tj][r] - p2[tj][r], p3[tj][r] - p1[tj][r] - p2[tj][r]);
                    }
                }
            };
            for (int ti = wave; ti < mt16; ti += 4) {
                cplx xa[KSN];
                load_x(xa, ti);
                mul_store(xa, ti);
            }
        }
        __syncthreads();
        RF_STAMP(4 + 3 * pass);
    }
    for (int e = tid; e < M * nt; e += 512) phi_g[e] = phi_l[e];
    if (tid == 0) {
        const double d = exp(logd_s[0] + logd_s[1]);
        a.fail[w] = 0;
        a.detR[w] = d;
        a.ot[w] = cmake(a.ot[w].x / d, a.ot[w].y / d);       // single_det.py:253
        if (a.keep) a.keep[w] = cmake(a.keep[w].x / d, a.keep[w].y / d);
        if (a.fp) a.weight[w] *= d;                          // walkers/handler.py:178-181
    }
    RF_STAMP(8);
}

static bool reortho_fused_supported(afq_handle *h, size_t *lds_out) {
    const int nmax = h->na > h->nb ? h->na : h->nb;
    const size_t lds = sizeof(cplx) * ((size_t)h->M * h->nt + 2 * 32 * RF_LD);
    *lds_out = lds;
    static const bool off = AFQ_KNOB_SET("AFQ_NO_REORTHO_FUSED");
    return !off && nmax <= 32 && h->nb > 0 && h->M >= 16 && lds <= 150 * 1024;
}

static int k_reortho_fused(afq_handle *h, size_t lds, cplx *keep) {
    if (!h->qr_fail) AFQ_HIP(h, hipMalloc(&h->qr_fail, sizeof(int) * h->nw));
    RfArgs a;
    a.M = h->M; a.na = h->na; a.nb = h->nb; a.nt = h->nt; a.nw = h->nw;
    a.fp = (h->flags & AFQ_PROP_FREE_PROJECTION) ? 1 : 0;
    a.phi = h->phi; a.detR = h->detR; a.weight = h->weight; a.ot = h->ot; a.fail = h->qr_fail; a.keep = keep;
    const int nmax_rj = h->na > h->nb ? h->na : h->nb;
    const int rji = nmax_rj <= 8 ? 0 : nmax_rj <= 16 ? 1 : nmax_rj <= 26 ? 2 : 3;
    const void *kfn[4] = {(const void *)reortho_fused_kernel<4>, (const void *)reortho_fused_kernel<8>,
                          (const void *)reortho_fused_kernel<13>, (const void *)reortho_fused_kernel<16>};
    static size_t lds_set4[4][AFQ_MAX_DEVICES] = {{0}};
    AFQ_HIP(h, afq_raise_lds(kfn[rji], lds, lds_set4[rji]));
#ifdef AFQ_TUNING
    static unsigned long long *rfts = nullptr;
    static int rf_launch = 0;
    if (AFQ_KNOB_SET("AFQ_RF_TS")) {
        if (!rfts) { hipMalloc(&rfts, 13 * 8); hipMemset(rfts, 0, 13 * 8); hipMemcpyToSymbol(HIP_SYMBOL(afq_rf_ts), &rfts, sizeof(rfts)); }
        if (++rf_launch == 20) {
            unsigned long long t[13];
            hipStreamSynchronize(h->stream);
            hipMemcpy(t, rfts, sizeof(t), hipMemcpyDeviceToHost);
            fprintf(stderr, "RF_TS ticks: load %lld | pass 0: gram %lld chol %lld q %lld | pass 1: gram %lld chol %lld q %lld | store %lld\n",
                    (long long)(t[1] - t[0]), (long long)(t[2] - t[1]), (long long)(t[3] - t[2]), (long long)(t[4] - t[3]),
                    (long long)(t[5] - t[4]), (long long)(t[6] - t[5]), (long long)(t[7] - t[6]), (long long)(t[8] - t[7]));
            fprintf(stderr, "RF_TS gram detail: pass 0 k-loop %lld, to own end %lld, barrier %lld | pass 1 k-loop %lld, to own end %lld, barrier %lld\n",
                    (long long)(t[9] - t[1]), (long long)(t[11] - t[9]), (long long)(t[2] - t[11]),
                    (long long)(t[10] - t[4]), (long long)(t[12] - t[10]), (long long)(t[5] - t[12]));
        }
    }
#endif
    if (rji == 0) AFQ_LAUNCH(h, reortho_fused_kernel<4>, dim3(h->nw), dim3(512), lds, h->stream, a);
    else if (rji == 1) AFQ_LAUNCH(h, reortho_fused_kernel<8>, dim3(h->nw), dim3(512), lds, h->stream, a);
    else if (rji == 2) AFQ_LAUNCH(h, reortho_fused_kernel<13>, dim3(h->nw), dim3(512), lds, h->stream, a);
    else AFQ_LAUNCH(h, reortho_fused_kernel<16>, dim3(h->nw), dim3(512), lds, h->stream, a);
    AFQ_POST(h);
    return AFQ_OK;
}

// keep: the cached overlap to divide by det R along the way (null: none).  *keep_done tells the caller whether the kernels
// did that (fused Cholesky-QR path and its Gram-Schmidt fallback) or a separate pass is still due.
int k_reortho(afq_handle *h, cplx *keep, bool *keep_done) {
    QrArgs a;
    a.M = h->M; a.na = h->na; a.nb = h->nb; a.nt = h->nt; a.nw = h->nw; a.flags = h->flags;
    a.phi = h->phi; a.detR = h->detR; a.weight = h->weight; a.ot = h->ot;
    a.only = nullptr; a.keep = nullptr;
    if (keep_done) *keep_done = false;
    const int nmax = h->na > h->nb ? h->na : h->nb;
    static const bool no_cholqr = AFQ_KNOB_SET("AFQ_NO_CHOLQR");
    // Cholesky-QR2 on the GEMM engines: always for 45 < N <= 128; for smaller N once the population is
    // large enough that seven launches beat the one latency-bound Gram-Schmidt work-group per walker
    const bool small_ok = nmax <= 45 && h->nb > 0 && !h->no_ring && h->nw >= 64 && h->M >= 32;
    size_t lds_fused = 0;
    if (small_ok && !no_cholqr && reortho_fused_supported(h, &lds_fused)) {
        int rc = k_reortho_fused(h, lds_fused, keep);
        if (rc) return rc;
        a.only = h->qr_fail; a.keep = keep;
        if (keep_done) *keep_done = keep != nullptr;
    } else if ((k_greens_big_supported(h) || small_ok) && !no_cholqr) {
        int rc = k_reortho_big(h);
        if (rc) return rc;
        a.only = h->qr_fail;
    }
    const size_t need = sizeof(cplx) * (size_t)nmax * h->M;
    a.in_lds = need <= 64 * 1024;
    a.ws = h->phi_t2;       // scratch panel (nmax*M <= M*nt elements per walker)
    AFQ_LAUNCH(h, reortho_kernel, dim3(h->nw), dim3(NTHR), a.in_lds ? need : 0, h->stream, a);
    AFQ_POST(h);
    return AFQ_OK;
}

// --------------------------------------------------------------------------
__global__ void cap_kernel(double *weight, int nw, double cap) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nw && fabs(weight[w]) > cap) weight[w] = cap;       // qmc/afqmc.py:235-236
}

__global__ void cap_dev_kernel(double *weight, int nw, double frac, const double *scal) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const double cap = frac * scal[0];                          // total weight of the last comb
    if (w < nw && fabs(weight[w]) > cap) weight[w] = cap;
}

int k_cap_weights(afq_handle *h, double frac, double total_weight) {
    if (total_weight < 0.0) {
        AFQ_LAUNCH(h, cap_dev_kernel, dim3((h->nw + 255) / 256), dim3(256), 0, h->stream, h->weight, h->nw,
                           frac, h->scal);
        AFQ_POST(h);
        return AFQ_OK;
    }
    AFQ_LAUNCH(h, cap_kernel, dim3((h->nw + 255) / 256), dim3(256), 0, h->stream, h->weight, h->nw,
                       frac * total_weight);
    AFQ_POST(h);
    return AFQ_OK;
}

__global__ void scale_kernel(double *weight, double *unscaled, int nw, double scale) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nw) { unscaled[w] = weight[w]; weight[w] = weight[w] / scale; }   // handler.py:244-246
}

int k_scale_weights(afq_handle *h, double scale) {
    AFQ_LAUNCH(h, scale_kernel, dim3((h->nw + 255) / 256), dim3(256), 0, h->stream, h->weight,
                       h->unscaled, h->nw, scale);
    AFQ_POST(h);
    return AFQ_OK;
}

__global__ void reset_kernel(double *weight, int nw, const double *scal) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (scal && scal[1] < 0.0) return;                             // collapsed population: see comb_plan_kernel
    if (w < nw) weight[w] = 1.0;                                   // handler.py:337-338
}

int k_reset_weights(afq_handle *h, bool after_comb) {
    AFQ_LAUNCH(h, reset_kernel, dim3((h->nw + 255) / 256), dim3(256), 0, h->stream, h->weight, h->nw,
               after_comb ? (const double *)h->scal : (const double *)nullptr);
    AFQ_POST(h);
    return AFQ_OK;
}

// Single-rank comb (walkers/handler.py:225-301).  The cumulative sums and the
// walk over the comb teeth are inherently sequential and must reproduce the
// reference's left-to-right double additions, so one thread does them (nw adds).
// scal[0] = total weight (before scaling), scal[1] = number of (clone, kill) pairs.
// Comb population control, walkers/handler.py:225-338, decided by one work-group: every thread owns a
// contiguous chunk of walkers (sequential sums inside the chunk, prefix scans across chunks), every
// comb tooth is located in the cumulative weights by bisection, and the clone / kill lists are
// compacted with prefix counts so that the j-th walker with multiplicity > 1 overwrites the j-th walker
// with multiplicity 0 -- exactly the zip(clone, kill) pairing (one copy per parent, :295-301).
__global__ __launch_bounds__(256) void comb_plan_kernel(double *weight, double *unscaled, int nw, double r,
                                                         double target, int *parent_ix, int *pairs,
                                                         double *scal) {
    extern __shared__ __align__(16) unsigned char smem[];
    double *cs = (double *)smem;              // |w| / scale, then its inclusive cumulative sum
    int *pix = (int *)(cs + nw);
    int *clone_l = pix + nw, *kill_l = clone_l + nw;
    __shared__ double wtot_d[4];
    __shared__ int wtot_i[4];
    const int tid = threadIdx.x;
    const int per = (nw + 255) / 256;
    const int i0 = tid * per < nw ? tid * per : nw, i1 = (tid + 1) * per < nw ? (tid + 1) * per : nw;
    double loc = 0.0;
    for (int i = i0; i < i1; ++i) { const double a = fabs(weight[i]); cs[i] = a; pix[i] = 0; loc += a; }
    double total;
    (void)block_excl_scan256(loc, wtot_d, &total);            // sum(global_weights), handler.py:233
    if (tid == 0) scal[0] = total;
    // handler.py:236-241: the reference stops here.  Nothing is cloned, the weights are left alone, and the sticky
    // flag scal[2] makes the next synchronising call (afq_popcontrol_comb with outputs, afq_estimates_get)
    // return AFQ_EWEIGHT
    if (total < 1e-8) { if (tid == 0) { scal[1] = -1.0; scal[2] = 1.0; } return; }
    const double scale = total / target;
    loc = 0.0;
    for (int i = i0; i < i1; ++i) {
        unscaled[i] = weight[i];                              // handler.py:245
        weight[i] = weight[i] / scale;
        const double a = cs[i] / scale;                       // global_weights / scale, handler.py:248
        loc += a;
        cs[i] = loc;                                          // chunk-local running sum
    }
    double tot2;
    const double base = block_excl_scan256(loc, wtot_d, &tot2);   // sum(weights), handler.py:274
    for (int i = i0; i < i1; ++i) cs[i] += base;              // numpy.cumsum(weights)
    __syncthreads();
    const int ntarget = (int)target;
    const double step = tot2 / target;
    for (int ic = tid; ic < ntarget; ic += 256) {
        const double tooth = (ic + r) * step;
        int lo = 0, hi = nw;                                  // smallest iw with tooth < cs[iw]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tooth < cs[mid]) hi = mid; else lo = mid + 1;
        }
        if (lo < nw) atomicAdd(&pix[lo], 1);
    }
    __syncthreads();
    int nc = 0, nk = 0;
    for (int i = i0; i < i1; ++i) { nc += pix[i] > 1; nk += pix[i] == 0; }
    int totc, totk;
    int bc = block_excl_scan256(nc, wtot_i, &totc);
    int bk = block_excl_scan256(nk, wtot_i, &totk);
    for (int i = i0; i < i1; ++i) {
        if (pix[i] > 1) clone_l[bc++] = i;
        if (pix[i] == 0) kill_l[bk++] = i;
        parent_ix[i] = pix[i];
    }
    __syncthreads();
    const int np = totc < totk ? totc : totk;
    for (int j = tid; j < np; j += 256) { pairs[2 * j] = clone_l[j]; pairs[2 * j + 1] = kill_l[j]; }
    if (tid == 0) scal[1] = (double)np;
}

struct CloneArgs {
    long per;
    cplx *phi, *ot, *ehyb, *phase, *eloc;
    double *unscaled, *detR, *log_detR;
    const int *pairs;
    const double *scal;
    // back-propagation state travels with the walker (walkers/walker.py:89-90); null when off
    cplx *phi_old, *bp_hist, *bp_ph;
    double *bp_cos;
    int *bp_n;
    long hist_per;
    // a cached Green's function travels too; null when there is none
    cplx *ghalf, *ovlp_new;
    cplx *G;             // walker.G as walker state (mixed one_rdm); null otherwise
    long gsz;
    cplx *gsum;          // Ghalf_a + Ghalf_b of the force bias, kept in step with ghalf; null when it is not current
    long gsum_per;
    cplx *gdiag;         // Hubbard: diag(G) partial sums, kept in step with ghalf; null when they are not current
    long gdiag_per;
    double *weight;      // single-rank comb: every weight back to 1 (handler.py:337-338) in this launch; null otherwise
    int nw;
};

__global__ void clone_kernel(CloneArgs a) {
    const int pr = blockIdx.y;
    if (a.weight && blockIdx.x == 0 && !(a.scal[1] < 0.0)) {        // (collapsed population: see comb_plan_kernel)
        const int wr = blockIdx.y * blockDim.x + threadIdx.x;
        if (wr < a.nw) a.weight[wr] = 1.0;
    }
    if (pr >= (int)a.scal[1]) return;
    const int src = a.pairs[2 * pr], dst = a.pairs[2 * pr + 1];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < a.per; i += (long)gridDim.x * blockDim.x)
        a.phi[dst * a.per + i] = a.phi[src * a.per + i];
    if (a.ghalf) {
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < a.per; i += (long)gridDim.x * blockDim.x)
            a.ghalf[dst * a.per + i] = a.ghalf[src * a.per + i];
        if (blockIdx.x == 0 && threadIdx.x == 0) a.ovlp_new[dst] = a.ovlp_new[src];
    }
    if (a.gsum)
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < a.gsum_per; i += (long)gridDim.x * blockDim.x)
            a.gsum[dst * a.gsum_per + i] = a.gsum[src * a.gsum_per + i];
    if (a.gdiag)
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < a.gdiag_per; i += (long)gridDim.x * blockDim.x)
            a.gdiag[dst * a.gdiag_per + i] = a.gdiag[src * a.gdiag_per + i];
    if (a.G)
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < a.gsz; i += (long)gridDim.x * blockDim.x)
            a.G[dst * a.gsz + i] = a.G[src * a.gsz + i];
    if (a.phi_old) {
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < a.per; i += (long)gridDim.x * blockDim.x)
            a.phi_old[dst * a.per + i] = a.phi_old[src * a.per + i];
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < a.hist_per; i += (long)gridDim.x * blockDim.x)
            a.bp_hist[dst * a.hist_per + i] = a.bp_hist[src * a.hist_per + i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.ot[dst] = a.ot[src]; a.ehyb[dst] = a.ehyb[src]; a.phase[dst] = a.phase[src];
        a.eloc[dst] = a.eloc[src]; a.unscaled[dst] = a.unscaled[src]; a.detR[dst] = a.detR[src];
        a.log_detR[dst] = a.log_detR[src];
        if (a.phi_old) { a.bp_ph[dst] = a.bp_ph[src]; a.bp_cos[dst] = a.bp_cos[src]; a.bp_n[dst] = a.bp_n[src]; }
    }
}

__global__ void scale_by_inverse_kernel(cplx *x, const double *d, int nw) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nw) x[w] = cmake(x[w].x / d[w], x[w].y / d[w]);
}

// use_log_shift, walkers/single_det.py:250-253 on top of the plain QR bookkeeping the re-orthogonalisation kernels
// leave behind (detR = det R, ot /= det R, free projection: weight *= det R):
// detR -> exp(log det R - detR_shift), ot and the free-projection weight follow, log_detR += log(detR).
__global__ void log_shift_reortho_kernel(cplx *ot, double *detR, double *weight, double *log_detR, int nw,
                                         double detR_shift, int fp) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    const double raw = detR[w], d = exp(log(raw) - detR_shift), f = raw / d;
    detR[w] = d;
    ot[w] = cscale(ot[w], f);
    if (fp) weight[w] /= f;
    log_detR[w] += log(d);
}

int k_log_shift_reortho(afq_handle *h) {
    AFQ_LAUNCH(h, log_shift_reortho_kernel, dim3((h->nw + 127) / 128), dim3(128), 0, h->stream, h->ot, h->detR,
               h->weight, h->log_detR, h->nw, h->detR_shift, (h->flags & AFQ_PROP_FREE_PROJECTION) ? 1 : 0);
    AFQ_POST(h);
    return AFQ_OK;
}

// walkers/handler.py:457-462: sums of |ot|, |detR|, |log_detR| over this rank's walkers, one work-group
__global__ __launch_bounds__(256) void log_ovlp_sums_kernel(const cplx *ot, const double *detR, const double *log_detR,
                                                            int nw, double *out) {
    __shared__ double red[3][256];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int w = threadIdx.x; w < nw; w += 256) {
        s0 += hypot(ot[w].x, ot[w].y); s1 += fabs(detR[w]); s2 += fabs(log_detR[w]);
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 3) out[threadIdx.x] = red[threadIdx.x][0];
}

int k_log_ovlp_sums(afq_handle *h, double *out3) {
    double *tmp = (double *)h->pack_tmp;
    AFQ_LAUNCH(h, log_ovlp_sums_kernel, dim3(1), dim3(256), 0, h->stream, h->ot, h->detR, h->log_detR, h->nw, tmp);
    AFQ_POST(h);
    AFQ_HIP(h, hipMemcpyAsync(out3, tmp, 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    AFQ_HIP(h, hipStreamSynchronize(h->stream));
    return AFQ_OK;
}

int k_scale_by_inverse(afq_handle *h, cplx *x, const double *d) {
    AFQ_LAUNCH(h, scale_by_inverse_kernel, dim3((h->nw + 127) / 128), dim3(128), 0, h->stream, x, d, h->nw);
    AFQ_POST(h);
    return AFQ_OK;
}

// clone_kernel over the (src, dst) pairs in h->pack_tmp, count in scal[1]; at most nw / 2 pairs
int k_clone_pairs(afq_handle *h, bool with_greens, bool reset_weights) {
    // cloned walkers bring their Ghalf along, and its spin sum when that is current (it then stays current)
    const bool sum_too = with_greens && h->ghalf_sum && h->gsum_version == h->ghalf_version;
    const bool diag_too = with_greens && h->gdiag && h->gdiag_version == h->ghalf_version;
    const bool closed_too = with_greens && h->closed_checked_version == h->ghalf_version;   // clones are whole walkers
    ++h->ghalf_version;
    if (closed_too) h->closed_checked_version = h->ghalf_version;
    if (sum_too) h->gsum_version = h->ghalf_version;
    if (diag_too) h->gdiag_version = h->ghalf_version;
    CloneArgs a;
    a.gdiag = diag_too ? h->gdiag : nullptr; a.gdiag_per = 2L * h->gdiag_parts * h->M;
    a.gsum = sum_too ? h->ghalf_sum : nullptr; a.gsum_per = (long)h->na * h->M;
    a.weight = reset_weights ? h->weight : nullptr; a.nw = h->nw;
    a.per = (long)h->M * h->nt; a.phi = h->phi; a.ot = h->ot; a.ehyb = h->ehyb; a.phase = h->phase;
    a.eloc = h->eloc; a.unscaled = h->unscaled; a.detR = h->detR; a.log_detR = h->log_detR;
    a.pairs = (const int *)h->pack_tmp; a.scal = h->scal;
    a.phi_old = h->nbp > 0 ? h->phi_old : nullptr; a.bp_hist = h->bp_hist; a.bp_ph = h->bp_ph; a.bp_cos = h->bp_cos;
    a.bp_n = h->bp_n; a.hist_per = (long)h->nbp * h->K;
    a.ghalf = with_greens ? h->ghalf : nullptr; a.ovlp_new = h->ovlp_new;
    a.G = (h->rdm_on && h->G) ? h->G : nullptr; a.gsz = 2L * h->M * h->M;
    // (grid.y * 256 threads of the x == 0 blocks cover every walker for the weight reset)
    AFQ_LAUNCH(h, clone_kernel, dim3(4, (h->nw + 1) / 2), dim3(256), 0, h->stream, a);
    AFQ_POST(h);
    return AFQ_OK;
}

// ---- closed-shell walkers on the large-system path (afq_internal.h: closed_large) ----
// one work-group per walker: are the alpha and the beta block bitwise equal?  (na == nb)
__global__ __launch_bounds__(256) void closed_flags_kernel(const cplx *phi, int M, int na, int nt, const int *alive, int *flags,
                                                           unsigned long long *counters) {
    const int w = blockIdx.x;
    if (alive && !alive[w]) { if (threadIdx.x == 0) flags[w] = 0; return; }
    const cplx *p = phi + (long)w * M * nt;
    int same = 1;
    for (int e = threadIdx.x; e < M * na; e += 256) {
        const int r = e / na, c = e - r * na;
        const cplx x = p[r * nt + c], y = p[r * nt + na + c];
        same &= (int)((__double_as_longlong(x.x) == __double_as_longlong(y.x)) & (__double_as_longlong(x.y) == __double_as_longlong(y.y)));
    }
    const int all = __syncthreads_and(same);
    if (threadIdx.x == 0) {
        flags[w] = all ? 1 : 0;
        if (all && counters) atomicAdd(&counters[7], 1ull);       // afq_counters_ext [7]
    }
}

__global__ __launch_bounds__(256) void closed_copy_beta_kernel(cplx *phi, int M, int na, int nt, const int *flags) {
    const int w = blockIdx.x;
    if (!flags[w]) return;
    cplx *p = phi + (long)w * M * nt;
    for (int e = threadIdx.x; e < M * na; e += 256) {
        const int r = e / na, c = e - r * na;
        p[r * nt + na + c] = p[r * nt + c];
    }
}

int k_closed_flags(afq_handle *h) {
    if (!h->closed_w || h->closed_w_n < h->nw) {
        if (h->closed_w) hipFree(h->closed_w);
        h->closed_w = nullptr;
        AFQ_HIP(h, hipMalloc(&h->closed_w, sizeof(int) * (size_t)h->nw));
        h->closed_w_n = h->nw;
    }
    AFQ_LAUNCH(h, closed_flags_kernel, dim3(h->nw), dim3(256), 0, h->stream, h->phi, h->M, h->na, h->nt, h->alive, h->closed_w, h->counters);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_closed_copy_beta(afq_handle *h) {

    AFQ_LAUNCH(h, closed_copy_beta_kernel, dim3(h->nw), dim3(256), 0, h->stream, h->phi, h->M, h->na, h->nt, h->closed_w);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_comb(afq_handle *h, double r, double target, bool with_greens) {
    if (k_comm_size(h) > 1 || h->comm) return k_comm_popcontrol(h, r, target, with_greens);
    int *pairs = (int *)h->pack_tmp;
    AFQ_LAUNCH(h, comb_plan_kernel, dim3(1), dim3(256), (sizeof(double) + 3 * sizeof(int)) * (size_t)h->nw,
               h->stream, h->weight, h->unscaled, h->nw, r, target, h->parent_ix, pairs, h->scal);
    AFQ_POST(h);
    return k_clone_pairs(h, with_greens, true);       // the weights go back to 1 in the same launch
}

// --------------------------------------------------------------------------
// estimators/mixed.py:211-225: one workgroup, deterministic tree sums.
// estimators/mixed.py:151-175, 211-225: one workgroup, deterministic tree sums.
__global__ __launch_bounds__(NTHR) void estimates_kernel(int nw, int have_energy, int fp, const double *weight,
                                                         const double *unscaled, const cplx *ot,
                                                         const cplx *ehyb, const cplx *phase, const cplx *energy,
                                                         cplx *est, double *acc, int fold_only, EstPublish pub) {
    __shared__ double red[NTHR / 64][18];
    // v[0]=uweight  v[1..2]=weight  v[3]=ovlp  v[4..5]=ehyb  v[6..7]=enumer  v[8..9]=e1b  v[10..11]=e2b
    // v[12..17] = the first six again for the steps whose sums rode on their weight update (afq_estimates_fuse_next):
    // kept apart, the energy denominator of THIS step is this step's weight sum only
    double v[18];
    for (int k = 0; k < 18; ++k) v[k] = 0.0;
    for (int w = threadIdx.x; w < nw; w += NTHR) {
        if (acc) {       // fold and clear
#pragma unroll
            for (int k = 0; k < 6; ++k) { v[12 + k] += acc[6 * w + k]; acc[6 * w + k] = 0.0; }
        }
        if (fold_only) continue;
        const double x = weight[w];
        // importance sampling: wfac = weight (mixed.py:217-225); free projection: weight*ot*phase (:154)
        cplx wf = cmake(x, 0.0);
        if (fp) wf = cscale(cmul(ot[w], phase[w]), x);
        v[0] += unscaled[w];
        v[1] += wf.x; v[2] += wf.y;
        v[3] += x * hypot(ot[w].x, ot[w].y);
        const cplx eh = cmul(wf, ehyb[w]);
        v[4] += eh.x; v[5] += eh.y;
        if (have_energy) {
            if (fp) {
                const cplx e0 = cmul(wf, energy[3 * w]), e1 = cmul(wf, energy[3 * w + 1]), e2 = cmul(wf, energy[3 * w + 2]);
                v[6] += e0.x; v[7] += e0.y; v[8] += e1.x; v[9] += e1.y; v[10] += e2.x; v[11] += e2.y;
            } else {
                v[6] += x * energy[3 * w].x; v[8] += x * energy[3 * w + 1].x; v[10] += x * energy[3 * w + 2].x;
            }
        }
    }
    // all sums in one pass: wave shuffles, one barrier, thread 0 adds the per-wave partials in wave order
    const int nk = have_energy ? 12 : 6;                     // the energy sums stay zero (and unused) on the other steps
#pragma unroll
    for (int k = 0; k < 18; ++k)
        if (k < nk || (k >= 12 && acc))
            for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 18; ++k) red[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 18; ++k) {
            v[k] = 0.0;
            for (int i = 0; i < NTHR / 64; ++i) v[k] += red[i][k];
        }
        est[AFQ_EST_UWEIGHT].x += v[0] + v[12];
        est[AFQ_EST_WEIGHT].x += v[1] + v[13]; est[AFQ_EST_WEIGHT].y += v[2] + v[14];
        est[AFQ_EST_OVLP].x += v[3] + v[15];
        est[AFQ_EST_EHYB].x += v[4] + v[16]; est[AFQ_EST_EHYB].y += v[5] + v[17];
        if (have_energy) {
            est[AFQ_EST_ENUMER].x += v[6]; est[AFQ_EST_ENUMER].y += v[7];
            est[AFQ_EST_E1B].x += v[8]; est[AFQ_EST_E1B].y += v[9];
            est[AFQ_EST_E2B].x += v[10]; est[AFQ_EST_E2B].y += v[11];
            est[AFQ_EST_EDENOM].x += v[1]; est[AFQ_EST_EDENOM].y += v[2];
        }
    }
    if (pub.host_out) {
        // the block's sums go to the host from this launch (afq_api.hip: est_publish_kernel, the same steps): one launch less
        // at every block boundary
        __syncthreads();                                 // thread 0's sums, for the threads that copy them out
        const int t = threadIdx.x;
        if (t < pub.nest) pub.host_out[t] = ((const double *)est)[t];
        if (t < AFQ_NSCAL) pub.host_out[pub.nest + t] = pub.scal[t];
        if (t == 0) pub.host_seq[1] = pub.closed_bad ? *pub.closed_bad : 0ull;
        __threadfence_system();
        __syncthreads();
        if (t < pub.nest && pub.zero) ((double *)est)[t] = 0.0;
        if (t == 0) __hip_atomic_store(pub.host_seq, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// estimators/mixed.py:226-229: estimates[one_rdm] += weight * walker.G.real, summed over walkers in a fixed order
__global__ void rdm_accumulate_kernel(const cplx *G, const double *weight, double *acc, int nw, long gsz) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= gsz) return;
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s = fma(weight[w], G[(long)w * gsz + e].x, s);
    acc[e] += s;
}

int k_rdm_accumulate(afq_handle *h) {
    const long gsz = 2L * h->M * h->M;
    AFQ_LAUNCH(h, rdm_accumulate_kernel, dim3((unsigned)((gsz + 127) / 128)), dim3(128), 0, h->stream, h->G, h->weight,
               h->rdm_acc, h->nw, gsz);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_estimates(afq_handle *h, int have_energy, bool fold_only, const EstPublish *pub) {
    const int fp = (h->flags & AFQ_PROP_FREE_PROJECTION) ? 1 : 0;
    if (fold_only && !h->est_acc_pending) return AFQ_OK;
    AFQ_LAUNCH(h, estimates_kernel, dim3(1), dim3(NTHR), 0, h->stream, h->nw, have_energy, fp, h->weight,
                       h->unscaled, h->ot, h->ehyb, h->phase, h->energy, h->estimates,
                       h->est_acc_pending ? h->est_acc : (double *)nullptr, fold_only ? 1 : 0, pub ? *pub : EstPublish());
    AFQ_POST(h);
    h->est_acc_pending = false;
    return AFQ_OK;
}

// also refreshes the alive flags of the step (qmc/afqmc.py:232) so that the device-RNG path needs no
// separate alive_kernel launch
__global__ void rng_normal_kernel(double *xi, long n, unsigned long long seed, unsigned long long stream,
                                  unsigned long long counter, const double *weight, int *alive, int nw) {
    const long pair = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (pair < nw) alive[pair] = fabs(weight[pair]) > 1e-8 ? 1 : 0;
    if (2 * pair >= n) return;
    double x0, x1;
    philox_normal_pair(pair, seed, stream, counter, x0, x1);
    xi[2 * pair] = x0;
    if (2 * pair + 1 < n) xi[2 * pair + 1] = x1;
}

// uniforms in [0, 1) (53 bits), same counter-based stream
__global__ void rng_uniform_kernel(double *u, long n, unsigned long long seed, unsigned long long stream,
                                   unsigned long long counter) {
    const long pair = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (2 * pair >= n) return;
    unsigned int c0 = (unsigned int)pair, c1 = (unsigned int)(pair >> 32);
    unsigned int c2 = (unsigned int)counter, c3 = (unsigned int)(counter >> 32) ^ (unsigned int)(stream * 0x9E3779B9u);
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
    for (int rd = 0; rd < 10; ++rd) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const unsigned long long a = (((unsigned long long)c0 << 32) | c1) >> 11;
    const unsigned long long b = (((unsigned long long)c2 << 32) | c3) >> 11;
    u[2 * pair] = (double)a * (1.0 / 9007199254740992.0);
    if (2 * pair + 1 < n) u[2 * pair + 1] = (double)b * (1.0 / 9007199254740992.0);
}

int k_rng_uniform(afq_handle *h, double *u, long n) {
    const long pairs = (n + 1) / 2;
    AFQ_LAUNCH(h, rng_uniform_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, h->stream, u, n,
                       (unsigned long long)h->rng_seed, (unsigned long long)h->rng_stream,
                       (unsigned long long)h->rng_counter);
    AFQ_POST(h);
    h->rng_counter += 1;
    return AFQ_OK;
}

// test hooks (afq_rng_normal / afq_rng_philox4x32): the same stream into a caller-provided device buffer, and
// the bare Philox4x32-10 block function for the Random123 known-answer vectors
int k_rng_normal_into(afq_handle *h, double *out_d, long n) {
    const long pairs = (n + 1) / 2;
    AFQ_LAUNCH(h, rng_normal_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, h->stream, out_d, n,
               (unsigned long long)h->rng_seed, (unsigned long long)h->rng_stream,
               (unsigned long long)h->rng_counter, (const double *)nullptr, (int *)nullptr, 0);
    AFQ_POST(h);
    h->rng_counter += 1;
    return AFQ_OK;
}

__global__ void philox_raw_kernel(const unsigned int *in, unsigned int *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned int c0 = in[6 * i], c1 = in[6 * i + 1], c2 = in[6 * i + 2], c3 = in[6 * i + 3];
    unsigned int k0 = in[6 * i + 4], k1 = in[6 * i + 5];
    for (int rd = 0; rd < 10; ++rd) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[4 * i] = c0; out[4 * i + 1] = c1; out[4 * i + 2] = c2; out[4 * i + 3] = c3;
}

int k_philox_raw(afq_handle *h, const unsigned int *in_d, unsigned int *out_d, int n) {
    AFQ_LAUNCH(h, philox_raw_kernel, dim3((n + 63) / 64), dim3(64), 0, h->stream, in_d, out_d, n);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_rng_normal(afq_handle *h) {
    const long n = (long)h->nw * h->K;
    const long pairs = std::max((n + 1) / 2, (long)h->nw);
    AFQ_LAUNCH(h, rng_normal_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, h->stream, h->xi, n,
                       (unsigned long long)h->rng_seed, (unsigned long long)h->rng_stream,
                       (unsigned long long)h->rng_counter, h->weight, h->alive, h->nw);
    AFQ_POST(h);
    h->rng_counter += 1;
    return AFQ_OK;
}
