// Plane-wave (UEG) step without the M x M intermediates (propagation/planewave.py:57-112, continuous.py:133-171,251-258).
//
// The reference forms, per walker and step, the full Green's function G[2, M, M], gathers the force bias from it through
// the sparse iA / iB (a column per field), and scatters the shifted fields back through iA / iB into a dense M x M HS
// potential that the Taylor series then multiplies six times.  Two structural facts of a plane-wave Hamiltonian make all
// three M x M objects unnecessary (both are CHECKED at upload on the matrices the caller hands over, not assumed):
//   * an element (i, j) of the HS potential receives a handful of entries of [iA | iB] (the momentum transfer k_i - k_j
//     fixes the fields: +q and -q of iA and of iB, four terms), and the DISTINCT term lists are few -- one per momentum
//     transfer: V[w][i, j] = c[w][id(i, j)] with c[w][id] = sqrt(dt) sum_t v_t xs[w][n_t] -- a few hundred coefficients
//     per walker (9 KB at C2) and one int16 map id(i, j) shared by all walkers;
//   * the trial occupies few orbitals: G_s = conj(psi_s) Ghalf_s has non-zero rows only where psi has non-zero rows (the
//     7 + 7 occupied plane waves of the Hartree-Fock trial), so the force bias gathers from `nrows` rows of
//     G_up + G_dn, built from Ghalf in LDS, with a per-field list restricted to those rows (~14 entries instead of ~190).
// ueg_fields_kernel   Ghalf -> compact (G_up + G_dn) rows in LDS -> force bias, clip, fields (device Philox or xi),
//                     shifted fields, the mean-field / force-bias factors, and the coefficients c[w][.]
//                     (replaces FullGProb x 2, vbias_ueg_lds_kernel, fields_kernel, vhs_ueg_kernel)
// prop_ueg_kernel     phi <- B exp(V) B phi with B = exp(-dt/2 H1) DIAGONAL (plane waves diagonalise the kinetic
//                     energy; checked at afq_set_propagator) and V applied from c[w] and the id map, both in LDS: one
//                     work-group per walker, both spins in ONE 16-column MFMA tile (na + nb <= 16), operand fragments
//                     of V gathered from LDS, 3-multiplication complex products, the walker never leaves the CU.
//                     HBM traffic per walker and step: Ghalf 21 KB + phi in/out 42 KB + fields 72 KB.
// Anything the checks do not cover (a trial with many occupied rows, one_rdm, local-energy weights, M > 112,
// na + nb > 16, a non-diagonal one-body propagator) takes the general kernels of k_models.hip / k_fused.hip.
#include <algorithm>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

#include "lds_dma.h"
#include "philox.h"

namespace {

constexpr int UF_NT = 512;              // threads of the propagator
#ifndef UF_NTF_OVERRIDE
#define UF_NTF_OVERRIDE 1024
#endif
constexpr int UF_NTF = UF_NTF_OVERRIDE; // threads of the field kernel: latency bound (dependent table / LDS gathers, the
                                        // Philox + Box-Muller chain), one work-group per CU -- 16 waves hide twice as much as 8
constexpr int UF_MAX_COEF = 4095;       // coefficients per walker that fit the LDS budget with room to spare
constexpr int UF_MAX_ROWS = 32;
constexpr int UF_TERMS = 8;             // entries of [iA | iB] one element of the HS potential may collect

struct UegFast {
    // host copies of the sparse operators (the trial-dependent tables are rebuilt at afq_set_trial)
    int M = 0, nq = 0;
    std::vector<int64_t> Acp, Arow, Bcp, Brow;
    std::vector<double> Aval, Bval;
    // element structure
    bool elem_ok = false;
    int ncoef = 0, Mp = 0;
    short *elem_id = nullptr;       // [nrt][nks][64] coefficient of element (i, k) in MFMA A-fragment order: row tile rt,
                                    // k-step ks, lane (k & 3) * 16 + (i & 15); ncoef = the zero coefficient
    int nterms = 0;                 // terms per coefficient (padded with field -1)
    int *coef_q = nullptr;          // [ncoef][nterms] fields n_t (index into the K shifted fields), -1 = no term
    cplx *coef_v = nullptr;         // [ncoef][nterms] values v_t
    // trial structure
    bool trial_ok = false;
    int nrows = 0;
    cplx *psic_rows = nullptr;      // [nrows][nt] conj(psi[row, :])
    int *fb_off = nullptr;          // [K + 1] entries of field n: fb_off[n] .. fb_off[n + 1]  (1.7 on average at C2, at most 14)
    int nfb = 0;                    // entries of the lists (fb_off[K])
    int *fb_idx = nullptr;          // index into the compact G (row slot * M + column)
    cplx *fb_val = nullptr;
    // walkers
    cplx *vcoef = nullptr;          // [nw][ncoef + 1]
    int vcoef_nw = 0;
    // one-body propagator
    cplx *bdiag = nullptr;          // [2][M] diagonal of BH1 when BH1 is diagonal
    bool bdiag_ok = false;
};

UegFast *uf_of(afq_handle *h) { return (UegFast *)h->ueg_fast; }

template <class T> int upload_vec(afq_handle *h, T **dst, const std::vector<T> &v) {
    if (*dst) { hipFree(*dst); *dst = nullptr; }
    const size_t n = v.empty() ? 1 : v.size();
    AFQ_HIP(h, hipMalloc((void **)dst, sizeof(T) * n));
    if (!v.empty()) AFQ_HIP(h, hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return AFQ_OK;
}

// ------------------------------------------------------------------------------------------------ fields
struct UegFieldArgs {
    int M, nt, K, nq, nrows, ncoef, nterms;
    double sqrt_dt;
    const cplx *ghalf;          // [nw, nt, M]
    const cplx *psic_rows;      // [nrows, nt]
    const int *fb_off;
    const int *fb_idx;
    const cplx *fb_val;
    const int *coef_q;
    const cplx *coef_v;
    const double *xi;
    const cplx *mf;
    cplx *xbar, *xs, *cmf, *cfb, *vcoef;
    unsigned long long *counters;
    const int *alive;
    int force_bias;             // AFQ_PROP_FORCE_BIAS set
    // round 5: the walker's Ghalf, the trial rows and the two gather tables go to LDS by LDS-DMA at kernel start
    int nfb;                    // entries of the force-bias lists (fb_off[K])
    int fb_lds, coef_lds;       // the force-bias lists / the coefficient lists are staged in LDS (they fit)
    int abl;                    // tuning builds, timing ablations (WRONG results): 1 no Philox, 2 no gather + clip, 4 no G rows,
                                // 8 no coefficients
};

// copy `bytes` (a multiple of 16, the source allocation padded accordingly) from memory to LDS with every wave's requests
// in flight at once; the caller waits (s_waitcnt vmcnt(0)) and synchronises
__device__ inline void uf_dma(const void *src, void *dst, unsigned bytes, int wave, int nwaves, int lane) {
    for (unsigned b0 = (unsigned)wave * 1024u; b0 < bytes; b0 += (unsigned)nwaves * 1024u) {
        const unsigned bo = b0 + (unsigned)lane * 16u;
        if (bo < bytes) glds16((const char *)src + bo, (char *)dst + b0);
    }
}

// One 1024-thread work-group per live walker (propagation/planewave.py:57-112, continuous.py:133-158): force bias from the
// occupied rows of G_up + G_dn, fields (device Philox stream of fields_kernel, or the host's), clipping, shifts, and the
// coefficients that ARE the plane-wave HS potential.  The kernel is one work-group per CU in a single round -- its time is
// the latency of its dependent chain -- so (round 5) EVERYTHING it reads from memory is requested in its first
// instructions: Ghalf of the walker, the trial rows, the force-bias lists and the coefficient lists by LDS-DMA into LDS
// (20 + 2 + 50 + 25 KB at C2), the mean-field shift and the list offsets of the thread's own fields into registers; the
// Philox / Box-Muller arithmetic runs under that one memory latency, and every later gather (14 dependent table reads
// per occupied row, up to 14 per field, the terms of a coefficient) is an LDS read.  20.4 -> ~10 us at C2.
// (NT threads: 1024 as a kernel of its own, 512 as the head of ueg_step_kernel; coef_lds_out: the coefficients also go to
//  this LDS array -- the propagator's operand -- and the function returns false for a dead walker)
template <int NT>
__device__ __attribute__((always_inline)) inline bool ueg_fields_body(const UegFieldArgs &a, const FieldRng &rng, unsigned char *smem,
                                                                      cplx *coef_lds_out) {
    constexpr int UF_NTF = NT;
    cplx *gc = (cplx *)smem;                                 // [nrows][M] rows of G_up + G_dn
    cplx *xl = gc + (size_t)a.nrows * a.M;                   // [K] shifted fields
    cplx *ghl = xl + a.K;                                    // [nt][M] Ghalf of the walker
    cplx *psr = ghl + (size_t)a.nt * a.M;                    // [nrows][nt] (+ pad to 16 bytes: complex, none needed)
    const unsigned nfb4 = ((unsigned)a.nfb + 3u) & ~3u, ncq = (unsigned)a.ncoef * a.nterms, ncq4 = (ncq + 3u) & ~3u;
    cplx *fbv = psr + (size_t)a.nrows * a.nt;                // [nfb] when fb_lds
    int *fbi = (int *)(fbv + (a.fb_lds ? a.nfb : 0));        // [nfb4]
    cplx *cvl = (cplx *)(fbi + (a.fb_lds ? nfb4 : 0));       // [ncoef * nterms] when coef_lds
    int *cql = (int *)(cvl + (a.coef_lds ? ncq : 0));        // [ncq4]
    __shared__ double red[UF_NTF / 64][8];
    const int w = blockIdx.x, tid = threadIdx.x;
    if (rng.on) {
        const bool live = fabs(rng.weight[w]) > 1e-8;
        if (tid == 0) rng.alive_out[w] = live ? 1 : 0;
        if (!live) return false;
    } else if (a.alive && !a.alive[w]) return false;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NWV = UF_NTF / 64;
    // ---- every memory request of the kernel, up front
    uf_dma(a.ghalf + (long)w * a.nt * a.M, ghl, (unsigned)(a.nt * a.M) * 16u, wave, NWV, lane);
    uf_dma(a.psic_rows, psr, (unsigned)(a.nrows * a.nt) * 16u, wave, NWV, lane);
    if (a.fb_lds) {
        uf_dma(a.fb_val, fbv, (unsigned)a.nfb * 16u, wave, NWV, lane);
        uf_dma(a.fb_idx, fbi, nfb4 * 4u, wave, NWV, lane);
    }
    if (a.coef_lds) {
        uf_dma(a.coef_v, cvl, ncq * 16u, wave, NWV, lane);
        uf_dma(a.coef_q, cql, ncq4 * 4u, wave, NWV, lane);
    }
    const int K = a.K;
    const long e0 = (long)w * K;
    // a thread takes the two members of one Philox pair (the stream of fields_kernel: element e = w K + n is member e & 1
    // of pair e >> 1); pairs beyond the first UF_NTF of a walker go round the loop below again
    const long pr0 = (e0 >> 1) + tid, pr_last = (e0 + K - 1) >> 1;
    int nA = (int)(2 * pr0 - e0), zA[3] = {0, 0, 0};
    cplx mfA[2] = {cmake(0.0, 0.0), cmake(0.0, 0.0)};
    double xiA[2] = {0.0, 0.0};
    if (pr0 <= pr_last) {
#pragma unroll
        for (int m = 0; m < 3; ++m) { const int n = nA + m; zA[m] = a.fb_off[n < 0 ? 0 : (n > K ? K : n)]; }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int n = nA + m;
            if (n >= 0 && n < K) { mfA[m] = a.mf[n]; if (!rng.on) xiA[m] = a.xi[e0 + n]; }
        }
    }
    double xn0[2] = {0.0, 0.0};
    if (rng.on && pr0 <= pr_last && !(a.abl & 1)) philox_normal_pair(pr0, rng.seed, rng.stream, rng.counter, xn0[0], xn0[1]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // rows of G_up + G_dn that are not identically zero: sum over ALL columns of the trial (both spins) of conj(psi[row, c]) Ghalf[c, :]
    for (int e = tid; e < ((a.abl & 4) ? 0 : a.nrows * a.M); e += UF_NTF) {
        const int rr = e / a.M, j = e - rr * a.M;
        cplx acc = cmake(0.0, 0.0);
        for (int c0 = 0; c0 < a.nt; c0 += 8) {
            cplx pc[8], gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u < a.nt ? c0 + u : 0;
                pc[u] = psr[rr * a.nt + c]; gv[u] = ghl[c * a.M + j];
                if (c0 + u >= a.nt) pc[u] = cmake(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) cfma(acc, pc[u], gv[u]);
        }
        gc[e] = acc;
    }
    __syncthreads();
    // sums: mean-field shift (re, im), xi . xbar (re, im), xbar . xbar (re, im), clipped count  (continuous.py:140-158)
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    auto element = [&](const long e, const int n, const double x, const int z0, const int z1, const cplx mm) {
        cplx b = cmake(0.0, 0.0);
        if (a.force_bias && !(a.abl & 2)) {
            // propagation/planewave.py:70-76: vbias[n] = (G_up + G_dn) . column n of [iA | iB], xbar = -sqrt(dt) vbias
            cplx v = cmake(0.0, 0.0);
            if (a.fb_lds) { for (int z = z0; z < z1; ++z) cfma(v, fbv[z], gc[fbi[z]]); }
            else { for (int z = z0; z < z1; ++z) cfma(v, a.fb_val[z], gc[a.fb_idx[z]]); }
            b = cmake(-a.sqrt_dt * v.x, -a.sqrt_dt * v.y);
        }
        const double ab = (a.abl & 2) ? 0.0 : hypot(b.x, b.y);
        if (ab > 1.0) { b.x /= ab; b.y /= ab; acc[6] += 1.0; }
        const cplx sft = cmake(x - b.x, -b.y);
        a.xbar[e] = b;
        a.xs[e] = sft;
        xl[n] = sft;
        acc[0] += sft.x * mm.x - sft.y * mm.y;
        acc[1] += sft.x * mm.y + sft.y * mm.x;
        acc[2] += x * b.x; acc[3] += x * b.y;
        acc[4] += b.x * b.x - b.y * b.y;
        acc[5] += 2.0 * b.x * b.y;
    };
    if (pr0 <= pr_last) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int n = nA + m;
            if (n >= 0 && n < K) element(e0 + n, n, rng.on ? xn0[m] : xiA[m], zA[m], zA[m + 1], mfA[m]);
        }
    }
    for (long pr = pr0 + UF_NTF; pr <= pr_last; pr += UF_NTF) {       // (more than 2 UF_NTF fields per walker)
        double xn[2] = {0.0, 0.0};
        if (rng.on) philox_normal_pair(pr, rng.seed, rng.stream, rng.counter, xn[0], xn[1]);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const long e = 2 * pr + m;
            const int n = (int)(e - e0);
            if (n >= 0 && n < K) element(e, n, rng.on ? xn[m] : a.xi[e], a.fb_off[n], a.fb_off[n + 1], a.mf[n]);
        }
    }
#pragma unroll
    for (int q = 0; q < 7; ++q)
        for (int off = 32; off > 0; off >>= 1) acc[q] += __shfl_down(acc[q], off);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 7; ++q) red[tid >> 6][q] = acc[q];
    }
    __syncthreads();                                         // (also: every shifted field is in xl)
    if (tid == 0) {
        double t[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            t[q] = 0.0;
            for (int i = 0; i < UF_NTF / 64; ++i) t[q] += red[i][q];
        }
        a.cmf[w] = cmake(-a.sqrt_dt * t[0], -a.sqrt_dt * t[1]);
        a.cfb[w] = cmake(t[2] - 0.5 * t[4], t[3] - 0.5 * t[5]);
        if (t[6] > 0 && a.counters) atomicAdd(&a.counters[0], (unsigned long long)t[6]);
    }
    // coefficients of the HS potential (propagation/planewave.py:109-112): c[id] = sqrt(dt) sum_t v_t xs[n_t]
    cplx *vc = a.vcoef + (long)w * (a.ncoef + 1);
    for (int id = tid; id <= a.ncoef; id += UF_NTF) {
        cplx c = cmake(0.0, 0.0);
        if (id < a.ncoef && !(a.abl & 8)) {
            for (int t = 0; t < a.nterms; ++t) {
                const int n = a.coef_lds ? cql[id * a.nterms + t] : a.coef_q[id * a.nterms + t];
                if (n >= 0) cfma(c, a.coef_lds ? cvl[id * a.nterms + t] : a.coef_v[id * a.nterms + t], xl[n]);
            }
            c = cmake(a.sqrt_dt * c.x, a.sqrt_dt * c.y);
        }
        vc[id] = c;
        if (coef_lds_out) coef_lds_out[id] = c;
    }
    return true;
}

__global__ __launch_bounds__(UF_NTF) void ueg_fields_kernel(UegFieldArgs a, FieldRng rng) {
    extern __shared__ __align__(16) unsigned char smem[];
    (void)ueg_fields_body<UF_NTF>(a, rng, smem, nullptr);
}

// ------------------------------------------------------------------------------------------------ propagator
struct PropUegArgs {
    int M, na, nb, nt, order, ncoef, Mp, nrt, nks;
    const short *elem_id;       // [nrt][nks][64] fragment order
    const cplx *vcoef;          // [nw][ncoef + 1]
    const cplx *bdiag;          // [2][M]
    cplx *phi;                  // [nw][M][nt], updated in place
    const int *alive;
};

// One 512-thread work-group per live walker.  Matrix product C = V T with V [Mp x Mp] (never stored in memory: element
// (i, k) is coef[id[i][k]], both staged in LDS) and T [Mp x 16] (both spins side by side in the one column tile; LDS,
// B-fragment order: k-step ks = 4 rows of T, lane (k & 3) * 16 + column, 16 bytes each).  The nrt row tiles of C times
// two halves of the contraction are the 2 nrt units of a product, dealt to the 8 waves (unit u -> wave u & 7): with 6 row
// tiles every SIMD carries 3 units.  A wave's units are the same in all `order` products, so it gathers its operand
// fragments of V ONCE (id map and coefficients through LDS, bank-conflicted 16-byte gathers) and keeps them -- re, im and
// re + im for the 3-multiplication product -- in registers: per product only the T fragments (conflict-free, lane-linear)
// come from LDS.  The second-half units park their partial tile in LDS, the first-half unit of the same row tile adds it,
// scales by 1 / n (Taylor term n), adds it to the running sum it keeps in registers and writes it back into T for the
// next product: two barriers per product.  KSU = k-steps per unit (ceil(nks / 2), compile time: static register indices).
// what the propagator part requests from memory before anything else: its rows of the walker and the coefficient ids of its
// operand fragments.  ueg_step_kernel issues these loads ahead of the field part, so that they fly under it.
template <int KSU> struct PropUegPrefetch { cplx phi[4]; int id0[KSU], id1[KSU]; };

template <int KSU>
__device__ __attribute__((always_inline)) inline void prop_ueg_prefetch(const PropUegArgs &a, PropUegPrefetch<KSU> &pf) {
    const int w = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int M = a.M, nt = a.nt, nrt = a.nrt, nks = a.nks;
    const int nunits = 2 * nrt, u0 = wave, u1 = wave + 8;
    const bool has0 = u0 < nunits, has1 = u1 < nunits;
    const int rt0 = has0 ? u0 % nrt : 0, kh0 = has0 ? u0 / nrt : 0;
    const int rt1 = has1 ? u1 % nrt : 0, kh1 = has1 ? u1 / nrt : 0;
    const int ksplit = (nks + 1) >> 1;
    const int k00 = kh0 ? ksplit : 0, n0 = has0 ? (kh0 ? nks - ksplit : ksplit) : 0;
    const int k10 = kh1 ? ksplit : 0, n1 = has1 ? (kh1 ? nks - ksplit : ksplit) : 0;
    const cplx *phi = a.phi + (long)w * M * nt;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = rt0 * 16 + 4 * r + lk;
        pf.phi[r] = (has0 && kh0 == 0 && k < M && lr < nt) ? phi[(long)k * nt + lr] : cmake(0.0, 0.0);
    }
#pragma unroll
    for (int j = 0; j < KSU; ++j) {
        pf.id0[j] = j < n0 ? (int)a.elem_id[((long)rt0 * nks + k00 + j) * 64 + lane] : a.ncoef;
        pf.id1[j] = j < n1 ? (int)a.elem_id[((long)rt1 * nks + k10 + j) * 64 + lane] : a.ncoef;
    }
}

// (coef_ready: the coefficients are in LDS already -- written by ueg_fields_body in the same kernel, behind a barrier;
//  pf: operands requested earlier by prop_ueg_prefetch, or null)
template <int KSU, bool PRE>
__device__ __attribute__((always_inline)) inline void prop_ueg_body(const PropUegArgs &a, unsigned char *smem, const bool coef_ready,
                                                                    const PropUegPrefetch<KSU> &pf) {
    const int w = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int M = a.M, nt = a.nt, Mp = a.Mp, nrt = a.nrt, nks = a.nks;
    // ---- LDS carve: coefficients | T | partial tiles
    cplx *coef = (cplx *)smem;                                                   // [ncoef + 1]
    unsigned char *Tb = smem + (((size_t)(a.ncoef + 1) * 16 + 15) & ~(size_t)15);   // [nks][1024]
    unsigned char *Pb = Tb + (size_t)nks * 1024;                                  // [nrt][2][4][64] doubles (re / im)
    if (!coef_ready) {
        const cplx *vc = a.vcoef + (long)w * (a.ncoef + 1);
        for (int i = tid; i <= a.ncoef; i += UF_NT) coef[i] = vc[i];
    }
    cplx *phi = a.phi + (long)w * M * nt;
    // ---- units of this wave: u = wave and wave + 8; unit u: half kh = u / nrt of the contraction, row tile rt = u % nrt.
    // The kh == 0 unit of a row tile owns that tile of the running sum S and of T.
    const int nunits = 2 * nrt;
    const int u0 = wave, u1 = wave + 8;
    const bool has0 = u0 < nunits, has1 = u1 < nunits;
    const int rt0 = has0 ? u0 % nrt : 0, kh0 = has0 ? u0 / nrt : 0;
    const int rt1 = has1 ? u1 % nrt : 0, kh1 = has1 ? u1 / nrt : 0;
    const int ksplit = (nks + 1) >> 1;                           // k-steps [0, ksplit) and [ksplit, nks); ksplit <= KSU
    // element (row 4 r + lk of the tile, column lr) of a tile in accumulator layout <-> T entry (k-step rt * 4 + r, lane)
    const bool col_ok = lr < nt;
    const int spin = lr < a.na ? 0 : 1;
    // (u0 is a kh == 0 unit whenever wave < nrt; u1 = wave + 8 never is: nrt <= 7)
    const bool own = has0 && kh0 == 0;
    d4_t Sr = {0, 0, 0, 0}, Si = {0, 0, 0, 0};
    // ---- T_0 = B phi (row scaling), S = T_0
    if (own) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = rt0 * 16 + 4 * r + lk;
            cplx v = cmake(0.0, 0.0);
            if (k < M && col_ok) v = cmul(a.bdiag[spin * M + k], PRE ? pf.phi[r] : phi[(long)k * nt + lr]);
            Sr[r] = v.x; Si[r] = v.y;
            *(d2_t *)(Tb + (size_t)(rt0 * 4 + r) * 1024 + lane * 16) = (d2_t){v.x, v.y};
        }
    }
    __syncthreads();
    // ---- operand fragments of V for this wave's units: lane (lr, lk) holds V[rt * 16 + lr][4 ks + lk] for its k-steps.
    // The coefficient ids come lane-contiguous from the fragment-ordered map (one 128-byte line per fragment, all loads in
    // flight together, L2 resident: the map is the same for every walker), the coefficients from LDS.
    double ar0[KSU], ai0[KSU], as0[KSU], ar1[KSU], ai1[KSU], as1[KSU];
    const int k00 = kh0 ? ksplit : 0, n0 = has0 ? (kh0 ? nks - ksplit : ksplit) : 0;
    const int k10 = kh1 ? ksplit : 0, n1 = has1 ? (kh1 ? nks - ksplit : ksplit) : 0;
    {
        int id0[KSU], id1[KSU];
#pragma unroll
        for (int j = 0; j < KSU; ++j) {
            if constexpr (PRE) { id0[j] = pf.id0[j]; id1[j] = pf.id1[j]; }
            else {
                id0[j] = j < n0 ? (int)a.elem_id[((long)rt0 * nks + k00 + j) * 64 + lane] : a.ncoef;
                id1[j] = j < n1 ? (int)a.elem_id[((long)rt1 * nks + k10 + j) * 64 + lane] : a.ncoef;
            }
        }
#pragma unroll
        for (int j = 0; j < KSU; ++j) {
            const cplx v0 = coef[id0[j]], v1 = coef[id1[j]];
            ar0[j] = v0.x; ai0[j] = v0.y; as0[j] = v0.x + v0.y;
            ar1[j] = v1.x; ai1[j] = v1.y; as1[j] = v1.x + v1.y;
        }
    }
    const unsigned t_l = lds_addr(Tb);
    // one unit: its rows of V (registers) times T over its k-steps (k-steps beyond the unit's count multiply zeros: never
    // more than one, nks = 2 KSU or 2 KSU - 1)
    auto unit = [&](const double (&ar)[KSU], const double (&ai)[KSU], const double (&as)[KSU], const int k0, const int cnt,
                    d4_t &Cr, d4_t &Ci) __attribute__((always_inline)) {
        d4_t P1 = {0, 0, 0, 0}, P2 = {0, 0, 0, 0}, P3 = {0, 0, 0, 0};
        d2_t b[KSU];
#pragma unroll
        for (int j = 0; j < KSU; ++j) {
            const int ks = k0 + (j < cnt ? j : 0);
            b[j] = lds_read_b128(t_l + (unsigned)ks * 1024 + lane * 16);
        }
        // LDS returns in order: the first half of the fragments has landed when at most KSU - KSU / 2 reads are outstanding
        // (the sched_barriers keep the MFMAs, which the compiler sees no dependence for, behind the waits)
        constexpr int H = KSU / 2;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(KSU - H) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < H; ++j) {
            P1 = mfma16(ar[j], b[j][0], P1);
            P2 = mfma16(ai[j], b[j][1], P2);
            P3 = mfma16(as[j], b[j][0] + b[j][1], P3);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = H; j < KSU; ++j) {
            P1 = mfma16(ar[j], b[j][0], P1);
            P2 = mfma16(ai[j], b[j][1], P2);
            P3 = mfma16(as[j], b[j][0] + b[j][1], P3);
        }
        Cr = P1 - P2;
        Ci = P3 - P1 - P2;
    };
    for (int n = 1; n <= a.order; ++n) {
        d4_t C0r = {0, 0, 0, 0}, C0i = {0, 0, 0, 0}, C1r, C1i;
        if (has0) unit(ar0, ai0, as0, k00, n0, C0r, C0i);
        if (has1) {
            unit(ar1, ai1, as1, k10, n1, C1r, C1i);
            // (u1 is always a second-half unit: park the partial tile)
            double *p = (double *)Pb + (size_t)rt1 * 512;
#pragma unroll
            for (int r = 0; r < 4; ++r) { p[r * 64 + lane] = C1r[r]; p[256 + r * 64 + lane] = C1i[r]; }
        }
        if (has0 && kh0) {
            double *p = (double *)Pb + (size_t)rt0 * 512;
#pragma unroll
            for (int r = 0; r < 4; ++r) { p[r * 64 + lane] = C0r[r]; p[256 + r * 64 + lane] = C0i[r]; }
        }
        __syncthreads();                 // partial tiles parked; every wave is done reading T_{n-1}
        if (own) {
            const double *p = (const double *)Pb + (size_t)rt0 * 512;
            const double inv = 1.0 / n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double tr = (C0r[r] + p[r * 64 + lane]) * inv, ti = (C0i[r] + p[256 + r * 64 + lane]) * inv;
                Sr[r] += tr; Si[r] += ti;
                *(d2_t *)(Tb + (size_t)(rt0 * 4 + r) * 1024 + lane * 16) = (d2_t){tr, ti};
            }
        }
        __syncthreads();                 // T_n in place
    }
    // ---- phi = B S
    if (own) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = rt0 * 16 + 4 * r + lk;
            if (k < M && col_ok) phi[(long)k * nt + lr] = cmul(a.bdiag[spin * M + k], cmake(Sr[r], Si[r]));
        }
    }
}

template <int KSU>
__global__ __launch_bounds__(UF_NT) void prop_ueg_kernel(PropUegArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (!a.alive[blockIdx.x]) return;
    PropUegPrefetch<KSU> none;
    prop_ueg_body<KSU, false>(a, smem, false, none);
}

// Round 5: force bias + fields + coefficients AND the propagator in ONE launch of 512 threads per live walker (the two
// kernels above back to back in one work-group: the field part's arithmetic costs the same with 8 waves as with 16 -- it
// is one memory latency + three barriers -- and what the second launch cost is gone: 4.6 us of launch floor, the
// coefficients' trip through memory, the latency of the propagator's own first loads).  The propagator's LDS arrays
// (coefficients | T | partial tiles) start where the field part's walker-dependent staging (Ghalf, trial rows, force-bias
// lists) lies: all of it dead when the coefficients are written, behind the barrier that ends the field loop.
template <int KSU>
__global__ __launch_bounds__(UF_NT) void ueg_step_kernel(UegFieldArgs fa, FieldRng rng, PropUegArgs pa, unsigned prop_off) {
    extern __shared__ __align__(16) unsigned char smem[];
    PropUegPrefetch<KSU> pf;
    prop_ueg_prefetch<KSU>(pa, pf);                               // (in flight under the field part)
    if (!ueg_fields_body<UF_NT>(fa, rng, smem, (cplx *)(smem + prop_off))) return;      // (dead walker: the whole work-group)
    __syncthreads();                                              // coefficients in LDS; the field part's LDS is free
    prop_ueg_body<KSU, true>(pa, smem + prop_off, true, pf);
}

}  // namespace

void k_ueg_fast_free(afq_handle *h) {
    UegFast *f = uf_of(h);
    if (!f) return;
    for (void *p : {(void *)f->elem_id, (void *)f->coef_q, (void *)f->coef_v, (void *)f->psic_rows, (void *)f->fb_off, (void *)f->fb_idx,
                    (void *)f->fb_val, (void *)f->vcoef, (void *)f->bdiag})
        if (p) hipFree(p);
    delete f;
    h->ueg_fast = nullptr;
}

// afq_set_system_ueg: keep the operators and analyse the element structure (one coefficient per element)
int k_ueg_fast_system(afq_handle *h, int M, int nq, const int64_t *Acp, const int64_t *Arow, const double *Aval,
                      const int64_t *Bcp, const int64_t *Brow, const double *Bval) {
    k_ueg_fast_free(h);
    UegFast *f = new UegFast();
    h->ueg_fast = f;
    f->M = M; f->nq = nq;
    f->Acp.assign(Acp, Acp + nq + 1); f->Bcp.assign(Bcp, Bcp + nq + 1);
    f->Arow.assign(Arow, Arow + Acp[nq]); f->Brow.assign(Brow, Brow + Bcp[nq]);
    f->Aval.assign(Aval, Aval + 2 * Acp[nq]); f->Bval.assign(Bval, Bval + 2 * Bcp[nq]);
    const long mm = (long)M * M;
    if (M > 112) return AFQ_OK;          // general kernels
    // terms of every element, in the order of the fields (iA columns 0 .. nq - 1, then iB columns as fields nq ..)
    typedef std::vector<std::tuple<int, double, double>> Terms;
    std::vector<Terms> terms(mm);
    for (int pass = 0; pass < 2; ++pass) {
        const int64_t *cp = pass ? Bcp : Acp, *row = pass ? Brow : Arow;
        const double *val = pass ? Bval : Aval;
        for (int q = 0; q < nq; ++q)
            for (int64_t z = cp[q]; z < cp[q + 1]; ++z) {
                const int64_t e = row[z];
                if (e < 0 || e >= mm) return AFQ_OK;
                terms[e].emplace_back(pass * nq + q, val[2 * z], val[2 * z + 1]);
                if ((int)terms[e].size() > UF_TERMS) return AFQ_OK;
            }
    }
    std::map<Terms, int> ids;
    std::vector<const Terms *> order;
    const int Mp = ((M + 15) / 16) * 16;
    std::vector<short> eid((size_t)Mp * Mp, (short)-1);
    int nterms = 1;
    for (long e = 0; e < mm; ++e) {
        if (terms[e].empty()) continue;
        auto it = ids.find(terms[e]);
        int id;
        if (it == ids.end()) {
            id = (int)ids.size();
            if (id >= UF_MAX_COEF) return AFQ_OK;
            it = ids.emplace(terms[e], id).first;
            order.push_back(&it->first);
            nterms = std::max(nterms, (int)terms[e].size());
        } else id = it->second;
        eid[(size_t)(e / M) * Mp + (e % M)] = (short)id;
    }
    std::vector<int> cq((size_t)ids.size() * nterms, -1);
    std::vector<cplx> cv((size_t)ids.size() * nterms, cmake(0.0, 0.0));
    for (size_t id = 0; id < order.size(); ++id)
        for (size_t t = 0; t < order[id]->size(); ++t) {
            cq[id * nterms + t] = std::get<0>((*order[id])[t]);
            cv[id * nterms + t] = cmake(std::get<1>((*order[id])[t]), std::get<2>((*order[id])[t]));
        }
    while (cq.size() % 4) cq.push_back(-1);                      // (whole 16-byte pieces for the LDS-DMA copy of the field kernel)
    f->nterms = nterms;
    f->ncoef = (int)ids.size(); f->Mp = Mp;
    for (short &s : eid) if (s < 0) s = (short)f->ncoef;            // the zero coefficient
    // A-fragment order: what lane (lr, lk) of a wave multiplying row tile rt at k-step ks needs is element
    // (rt * 16 + lr, 4 ks + lk); stored lane-contiguous so that a wave's load of one fragment is one 128-byte line
    std::vector<short> frag((size_t)Mp * Mp);
    for (int rt = 0; rt < Mp / 16; ++rt)
        for (int ks = 0; ks < Mp / 4; ++ks)
            for (int l = 0; l < 64; ++l)
                frag[((size_t)rt * (Mp / 4) + ks) * 64 + l] = eid[(size_t)(rt * 16 + (l & 15)) * Mp + 4 * ks + (l >> 4)];
    int rc;
    if ((rc = upload_vec(h, &f->elem_id, frag))) return rc;
    if ((rc = upload_vec(h, &f->coef_q, cq))) return rc;
    if ((rc = upload_vec(h, &f->coef_v, cv))) return rc;
    f->elem_ok = true;
    return AFQ_OK;
}

// afq_set_trial: rows of G that are not identically zero and the force-bias lists restricted to them
int k_ueg_fast_trial(afq_handle *h, const double *psi) {
    UegFast *f = uf_of(h);
    if (!f) return AFQ_OK;
    f->trial_ok = false;
    if (!f->elem_ok) return AFQ_OK;
    const int M = f->M, nt = h->nt, nq = f->nq, K = 2 * nq;
    std::vector<int> slot(M, -1), rows;
    for (int i = 0; i < M; ++i) {
        bool nz = false;
        for (int c = 0; c < nt && !nz; ++c) nz = psi[2 * ((size_t)i * nt + c)] != 0.0 || psi[2 * ((size_t)i * nt + c) + 1] != 0.0;
        if (nz) { slot[i] = (int)rows.size(); rows.push_back(i); }
    }
    if (rows.empty() || (int)rows.size() > UF_MAX_ROWS) return AFQ_OK;
    f->nrows = (int)rows.size();
    std::vector<cplx> pr((size_t)f->nrows * nt);
    for (int rr = 0; rr < f->nrows; ++rr)
        for (int c = 0; c < nt; ++c)
            pr[(size_t)rr * nt + c] = cmake(psi[2 * ((size_t)rows[rr] * nt + c)], -psi[2 * ((size_t)rows[rr] * nt + c) + 1]);
    // per field n (column of [iA | iB]): the entries whose row of G is occupied
    std::vector<int> fo(K + 1, 0), fi;
    std::vector<cplx> fv;
    for (int n = 0; n < K; ++n) {
        const bool isB = n >= nq;
        const int q = isB ? n - nq : n;
        const std::vector<int64_t> &cp = isB ? f->Bcp : f->Acp, &row = isB ? f->Brow : f->Arow;
        const std::vector<double> &val = isB ? f->Bval : f->Aval;
        for (int64_t z = cp[q]; z < cp[q + 1]; ++z) {
            const int i = (int)(row[z] / M), j = (int)(row[z] % M);
            if (slot[i] >= 0) { fi.push_back(slot[i] * M + j); fv.push_back(cmake(val[2 * z], val[2 * z + 1])); }
        }
        fo[n + 1] = (int)fi.size();
    }
    f->nfb = (int)fi.size();
    while (fi.size() % 4) fi.push_back(0);                       // (whole 16-byte pieces for the LDS-DMA copy of the field kernel)
    int rc;
    if ((rc = upload_vec(h, &f->psic_rows, pr))) return rc;
    if ((rc = upload_vec(h, &f->fb_off, fo))) return rc;
    if ((rc = upload_vec(h, &f->fb_idx, fi))) return rc;
    if ((rc = upload_vec(h, &f->fb_val, fv))) return rc;
    f->trial_ok = true;
    return AFQ_OK;
}

// afq_set_propagator: a diagonal one-body propagator is applied as a row scaling
int k_ueg_fast_propagator(afq_handle *h, const double *BH1) {
    UegFast *f = uf_of(h);
    if (!f) return AFQ_OK;
    f->bdiag_ok = false;
    const int M = h->M;
    std::vector<cplx> d((size_t)2 * M);
    for (int s = 0; s < 2; ++s)
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < M; ++j) {
                const double re = BH1[2 * (((size_t)s * M + i) * M + j)], im = BH1[2 * (((size_t)s * M + i) * M + j) + 1];
                if (i == j) d[(size_t)s * M + i] = cmake(re, im);
                else if (re != 0.0 || im != 0.0) return AFQ_OK;
            }
    const int rc = upload_vec(h, &f->bdiag, d);
    if (rc) return rc;
    f->bdiag_ok = true;
    return AFQ_OK;
}

int k_ueg_fast_supported(afq_handle *h) {
    UegFast *f = uf_of(h);
    if (!f || h->kind != AFQ_SYS_UEG || h->no_fused) return 0;
    if (!f->elem_ok || !f->trial_ok || !f->bdiag_ok) return 0;
    if (h->ndet != 1 || h->rdm_on || h->psi_stride != 0 || h->nt > 16 || h->nb <= 0 || h->M > 112) return 0;
    if (!(h->flags & AFQ_PROP_HYBRID) || (h->flags & AFQ_PROP_FREE_PROJECTION)) return 0;
    const size_t lds1 = sizeof(cplx) * ((size_t)f->nrows * h->M + h->K + (size_t)h->nt * h->M + (size_t)f->nrows * h->nt);
    return lds1 <= 150 * 1024;
}

// argument blocks of the two halves of the plane-wave step; lds: the dynamic LDS either needs as a kernel of its own
static int ueg_field_args(afq_handle *h, UegFieldArgs &a, FieldRng &rng, size_t &lds, size_t &live_end) {
    UegFast *f = uf_of(h);
    if (f->vcoef_nw != h->nw || !f->vcoef) {
        if (f->vcoef) hipFree(f->vcoef);
        f->vcoef = nullptr;
        AFQ_HIP(h, hipMalloc(&f->vcoef, sizeof(cplx) * (size_t)h->nw * (f->ncoef + 1)));
        f->vcoef_nw = h->nw;
    }
    rng = FieldRng();
    if (h->rng_inline) {
        rng.on = 1; rng.seed = h->rng_seed; rng.stream = h->rng_stream; rng.counter = h->rng_inline_counter;
        rng.weight = h->weight; rng.alive_out = h->alive;
        h->rng_inline = false;
    }
    a.M = h->M; a.nt = h->nt; a.K = h->K; a.nq = h->nq; a.nrows = f->nrows; a.ncoef = f->ncoef;
    a.sqrt_dt = h->sqrt_dt; a.ghalf = h->ghalf; a.psic_rows = f->psic_rows; a.fb_off = f->fb_off; a.fb_idx = f->fb_idx; a.fb_val = f->fb_val;
    a.nterms = f->nterms; a.coef_q = f->coef_q; a.coef_v = f->coef_v; a.xi = h->xi; a.mf = h->mf_shift; a.xbar = h->xbar; a.xs = h->xs;
    a.cmf = h->cmf; a.cfb = h->cfb; a.vcoef = f->vcoef; a.counters = h->counters; a.alive = h->alive;
    a.force_bias = (h->flags & AFQ_PROP_FORCE_BIAS) ? 1 : 0;
    // walker-independent tables into LDS as far as they fit beside the walker's own arrays: force-bias lists first
    lds = sizeof(cplx) * ((size_t)f->nrows * h->M + h->K + (size_t)h->nt * h->M + (size_t)f->nrows * h->nt);
    const size_t fb_bytes = (size_t)f->nfb * 16 + (((size_t)f->nfb + 3) & ~(size_t)3) * 4;
    const size_t ncq = (size_t)f->ncoef * f->nterms, cq_bytes = ncq * 16 + ((ncq + 3) & ~(size_t)3) * 4;
    a.nfb = f->nfb;
    a.abl = AFQ_KNOB_INT("AFQ_UEG_ABL", 0);
    // (the work-group is alone on its CU: everything but 2 KB of the 160 KB for the kernel's static arrays may be used)
    const size_t cap = 158 * 1024;
    a.fb_lds = lds + fb_bytes <= cap && !AFQ_KNOB_SET("AFQ_UEG_NO_TABLE_LDS");
    if (a.fb_lds) lds += fb_bytes;
    live_end = lds;                     // what lies beyond is still read while the coefficients are written
    a.coef_lds = lds + cq_bytes <= cap && !AFQ_KNOB_SET("AFQ_UEG_NO_TABLE_LDS");
    if (a.coef_lds) lds += cq_bytes;
#ifdef AFQ_TUNING
    static bool said = false;
    if (!said && AFQ_KNOB_SET("AFQ_UEG_SAY")) {
        said = true;
        fprintf(stderr, "ueg_fields: nrows %d nfb %d ncoef %d nterms %d lds %zu fb_lds %d coef_lds %d\n", f->nrows, f->nfb, f->ncoef, f->nterms, lds, a.fb_lds, a.coef_lds);
    }
#endif
    return AFQ_OK;
}

static void ueg_prop_args(afq_handle *h, PropUegArgs &a, size_t &lds) {
    UegFast *f = uf_of(h);
    a.M = h->M; a.na = h->na; a.nb = h->nb; a.nt = h->nt; a.order = h->exp_order; a.ncoef = f->ncoef; a.Mp = f->Mp;
    a.nrt = f->Mp / 16; a.nks = f->Mp / 4;
    a.elem_id = f->elem_id; a.vcoef = f->vcoef; a.bdiag = f->bdiag; a.phi = h->phi; a.alive = h->alive;
    lds = (((size_t)(f->ncoef + 1) * 16 + 15) & ~(size_t)15) + (size_t)a.nks * 1024 + (size_t)a.nrt * 4096;
    // matrix-pipe flops: order products x 2 nrt units x KSU k-steps x 3 multiplications x 2048
    const int ksu = (a.nks + 1) / 2;
    h->issued_flops[AFQ_K_PROPAGATOR] = 3.0 * h->exp_order * 2.0 * a.nrt * ksu * 2048.0 * h->nw;
}

// force bias + fields + HS coefficients in one launch (replaces k_full_G, k_vbias_ueg, k_xbar_fields, k_vhs_ueg)
int k_ueg_fields(afq_handle *h) {
    UegFieldArgs a;
    FieldRng rng;
    size_t lds, live_end;
    const int rc = ueg_field_args(h, a, rng, lds, live_end);
    if (rc) return rc;
    static size_t lds_set[AFQ_MAX_DEVICES] = {0};
    AFQ_HIP(h, afq_raise_lds((const void *)ueg_fields_kernel, lds, lds_set));
    AFQ_LAUNCH(h, ueg_fields_kernel, dim3(h->nw), dim3(UF_NTF), lds, h->stream, a, rng);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_prop_ueg(afq_handle *h) {
    PropUegArgs a;
    size_t lds;
    ueg_prop_args(h, a, lds);
    KernelTrace kt(h, AFQ_K_PROPAGATOR);
    const int ksu = (a.nks + 1) / 2;
#define PU_LAUNCH_(KSU_, SLOT_)                                                                                \
    do {                                                                                                      \
        static size_t lds_set_[AFQ_MAX_DEVICES] = {0};                                                        \
        AFQ_HIP(h, afq_raise_lds((const void *)prop_ueg_kernel<KSU_>, lds, lds_set_));                        \
        AFQ_LAUNCH(h, prop_ueg_kernel<KSU_>, dim3(h->nw), dim3(UF_NT), lds, h->stream, a);                    \
    } while (0)
    switch (ksu) {
    case 2: PU_LAUNCH_(2, 0); break;
    case 4: PU_LAUNCH_(4, 1); break;
    case 6: PU_LAUNCH_(6, 2); break;
    case 8: PU_LAUNCH_(8, 3); break;
    case 10: PU_LAUNCH_(10, 4); break;
    case 12: PU_LAUNCH_(12, 5); break;
    default: PU_LAUNCH_(14, 6); break;
    }
#undef PU_LAUNCH_
    AFQ_POST(h);
    return AFQ_OK;
}

// the whole head of a plane-wave step (continuous.py:133-171, :251, :258): ONE launch when the propagator's LDS arrays fit
// where the field part's dead staging lies, else the two launches above
int k_ueg_step(afq_handle *h) {
    UegFieldArgs fa;
    FieldRng rng;
    PropUegArgs pa;
    size_t lds_f, live_end, lds_p;
    if (AFQ_KNOB_SET("AFQ_UEG_NO_STEP_FUSION")) { const int rc = k_ueg_fields(h); return rc ? rc : k_prop_ueg(h); }
    const bool was_inline = h->rng_inline;
    int rc = ueg_field_args(h, fa, rng, lds_f, live_end);
    if (rc) return rc;
    ueg_prop_args(h, pa, lds_p);
    // the propagator's arrays start at the field part's Ghalf copy: everything from there to live_end (Ghalf, trial rows,
    // force-bias lists) is dead once the field loop is over; the coefficient lists beyond it are not
    const size_t prop_off = sizeof(cplx) * ((size_t)fa.nrows * fa.M + fa.K);
    const bool fits = fa.coef_lds ? prop_off + lds_p <= live_end : prop_off + lds_p <= 158 * 1024;
    if (!fits) {
        h->rng_inline = was_inline;                 // (ueg_field_args consumed the flag: the two-launch path reads it again)
        rc = k_ueg_fields(h);
        return rc ? rc : k_prop_ueg(h);
    }
    const size_t lds = std::max(lds_f, prop_off + lds_p);
    KernelTrace kt(h, AFQ_K_PROPAGATOR);
    const int ksu = (pa.nks + 1) / 2;
#define US_LAUNCH_(KSU_)                                                                                       \
    do {                                                                                                      \
        static size_t lds_set_[AFQ_MAX_DEVICES] = {0};                                                        \
        AFQ_HIP(h, afq_raise_lds((const void *)ueg_step_kernel<KSU_>, lds, lds_set_));                        \
        AFQ_LAUNCH(h, ueg_step_kernel<KSU_>, dim3(h->nw), dim3(UF_NT), lds, h->stream, fa, rng, pa, (unsigned)prop_off); \
    } while (0)
    switch (ksu) {
    case 2: US_LAUNCH_(2); break;
    case 4: US_LAUNCH_(4); break;
    case 6: US_LAUNCH_(6); break;
    case 8: US_LAUNCH_(8); break;
    case 10: US_LAUNCH_(10); break;
    case 12: US_LAUNCH_(12); break;
    default: US_LAUNCH_(14); break;
    }
#undef US_LAUNCH_
    AFQ_POST(h);
    return AFQ_OK;
}
