// Plane-wave (UEG) step without the M x M intermediates (propagation/planewave.py:57-112, continuous.py:133-171,251-258).
//
// The reference forms, per walker and step, the full Green's function G[2, M, M], gathers the force bias from it through
// the sparse iA / iB (a column per field), and scatters the shifted fields back through iA / iB into a dense M x M HS
// potential that the Taylor series then multiplies six times.  Two structural facts of a plane-wave Hamiltonian make all
// three M x M objects unnecessary (both are CHECKED at upload on the matrices the caller hands over, not assumed):
//   * an element (i, j) of the HS potential receives at most ONE entry of iA and ONE of iB (the momentum transfer
//     k_i - k_j fixes the field), and the distinct (field, value) combinations are few: V[w][i, j] = c[w][id(i, j)] with
//     c[w][id] = sqrt(dt) (vA_id xs[w][qA_id] + vB_id xs[w][nq + qB_id]) -- about 2 nq coefficients per walker (24 KB at
//     C2) and one int16 map id(i, j) shared by all walkers;
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

constexpr int UF_NT = 512;              // threads of both kernels
constexpr int UF_MAX_COEF = 4095;       // coefficients per walker that fit the LDS budget with room to spare
constexpr int UF_MAX_ROWS = 32;

struct UegFast {
    // host copies of the sparse operators (the trial-dependent tables are rebuilt at afq_set_trial)
    int M = 0, nq = 0;
    std::vector<int64_t> Acp, Arow, Bcp, Brow;
    std::vector<double> Aval, Bval;
    // element structure
    bool elem_ok = false;
    int ncoef = 0, Mp = 0;
    short *elem_id = nullptr;       // [Mp][Mp] coefficient of element (i, j); ncoef = the zero coefficient
    int *coef_q = nullptr;          // [ncoef][2] field of the iA / iB entry, -1 without
    cplx *coef_v = nullptr;         // [ncoef][2] their values
    // trial structure
    bool trial_ok = false;
    int nrows = 0, fb_len = 0;
    cplx *psic_rows = nullptr;      // [nrows][nt] conj(psi[row, :])
    int *fb_idx = nullptr;          // [fb_len][K] index into the compact G (row slot * M + column); padded entries carry a zero value
    cplx *fb_val = nullptr;         // [fb_len][K]
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
    int M, nt, K, nq, nrows, fb_len, ncoef;
    double sqrt_dt;
    const cplx *ghalf;          // [nw, nt, M]
    const cplx *psic_rows;      // [nrows, nt]
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
};

__global__ __launch_bounds__(UF_NT) void ueg_fields_kernel(UegFieldArgs a, FieldRng rng) {
    extern __shared__ __align__(16) unsigned char smem[];
    cplx *gc = (cplx *)smem;                                 // [nrows][M] rows of G_up + G_dn
    cplx *xl = gc + (size_t)a.nrows * a.M;                   // [K] shifted fields
    __shared__ double red[UF_NT / 64][8];
    const int w = blockIdx.x, tid = threadIdx.x;
    if (rng.on) {
        const bool live = fabs(rng.weight[w]) > 1e-8;
        if (tid == 0) rng.alive_out[w] = live ? 1 : 0;
        if (!live) return;
    } else if (a.alive && !a.alive[w]) return;
    // rows of G_up + G_dn that are not identically zero: sum over ALL columns of the trial (both spins) of conj(psi[row, c]) Ghalf[c, :]
    const cplx *gh = a.ghalf + (long)w * a.nt * a.M;
    for (int e = tid; e < a.nrows * a.M; e += UF_NT) {
        const int rr = e / a.M, j = e - rr * a.M;
        cplx acc = cmake(0.0, 0.0);
        for (int c = 0; c < a.nt; ++c) cfma(acc, a.psic_rows[rr * a.nt + c], gh[(long)c * a.M + j]);
        gc[e] = acc;
    }
    __syncthreads();
    // sums: mean-field shift (re, im), xi . xbar (re, im), xbar . xbar (re, im), clipped count  (continuous.py:140-158)
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    const int K = a.K;
    const long e0 = (long)w * K;
    auto element = [&](const long e, const int n, const double xdev) {
        cplx b = cmake(0.0, 0.0);
        if (a.force_bias) {
            // propagation/planewave.py:70-76: vbias[n] = (G_up + G_dn) . column n of [iA | iB], xbar = -sqrt(dt) vbias
            cplx v = cmake(0.0, 0.0);
            for (int k = 0; k < a.fb_len; ++k) cfma(v, a.fb_val[(long)k * K + n], gc[a.fb_idx[(long)k * K + n]]);
            b = cmake(-a.sqrt_dt * v.x, -a.sqrt_dt * v.y);
        }
        const double ab = hypot(b.x, b.y);
        if (ab > 1.0) { b.x /= ab; b.y /= ab; acc[6] += 1.0; }
        const double x = rng.on ? xdev : a.xi[e];
        const cplx sft = cmake(x - b.x, -b.y);
        a.xbar[e] = b;
        a.xs[e] = sft;
        xl[n] = sft;
        const cplx mm = a.mf[n];
        acc[0] += sft.x * mm.x - sft.y * mm.y;
        acc[1] += sft.x * mm.y + sft.y * mm.x;
        acc[2] += x * b.x; acc[3] += x * b.y;
        acc[4] += b.x * b.x - b.y * b.y;
        acc[5] += 2.0 * b.x * b.y;
    };
    // a thread takes the two members of one Philox pair (the stream of fields_kernel: element e = w K + n is member e & 1
    // of pair e >> 1)
    for (long pr = (e0 >> 1) + tid; pr <= ((e0 + K - 1) >> 1); pr += UF_NT) {
        double xn[2] = {0.0, 0.0};
        if (rng.on) philox_normal_pair(pr, rng.seed, rng.stream, rng.counter, xn[0], xn[1]);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const long e = 2 * pr + m;
            const int n = (int)(e - e0);
            if (n >= 0 && n < K) element(e, n, xn[m]);
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
            for (int i = 0; i < UF_NT / 64; ++i) t[q] += red[i][q];
        }
        a.cmf[w] = cmake(-a.sqrt_dt * t[0], -a.sqrt_dt * t[1]);
        a.cfb[w] = cmake(t[2] - 0.5 * t[4], t[3] - 0.5 * t[5]);
        if (t[6] > 0 && a.counters) atomicAdd(&a.counters[0], (unsigned long long)t[6]);
    }
    // coefficients of the HS potential (propagation/planewave.py:109-112): c[id] = sqrt(dt) (vA xs[qA] + vB xs[nq + qB])
    cplx *vc = a.vcoef + (long)w * (a.ncoef + 1);
    for (int id = tid; id <= a.ncoef; id += UF_NT) {
        cplx c = cmake(0.0, 0.0);
        if (id < a.ncoef) {
            const int qa = a.coef_q[2 * id], qb = a.coef_q[2 * id + 1];
            if (qa >= 0) cfma(c, a.coef_v[2 * id], xl[qa]);
            if (qb >= 0) cfma(c, a.coef_v[2 * id + 1], xl[a.nq + qb]);
            c = cmake(a.sqrt_dt * c.x, a.sqrt_dt * c.y);
        }
        vc[id] = c;
    }
}

// ------------------------------------------------------------------------------------------------ propagator
struct PropUegArgs {
    int M, na, nb, nt, order, ncoef, Mp, nrt, nks;
    const short *elem_id;       // [Mp][Mp]
    const cplx *vcoef;          // [nw][ncoef + 1]
    const cplx *bdiag;          // [2][M]
    cplx *phi;                  // [nw][M][nt], updated in place
    const int *alive;
};

// One 512-thread work-group per live walker.  Matrix product C = V T with V [Mp x Mp] (never stored: element (i, k) is
// coef[id[i][k]], both in LDS) and T [Mp x 16] (both spins side by side in the one column tile; LDS, B-fragment order:
// k-step ks = 4 rows of T, lane (k & 3) * 16 + column, 16 bytes each).  The nrt row tiles of C times two halves of the
// contraction are the 2 nrt units of a product, dealt to the 8 waves (unit u -> wave u & 7): with 6 row tiles every SIMD
// carries 3 units.  The second-half units park their partial tile in LDS, the first-half unit of the same row tile adds
// it, scales by 1 / n (Taylor term n), adds it to the running sum it keeps in registers and writes it back into T for the
// next product: two barriers per product.
__global__ __launch_bounds__(UF_NT) void prop_ueg_kernel(PropUegArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int w = blockIdx.x;
    if (!a.alive[w]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int M = a.M, nt = a.nt, Mp = a.Mp, nrt = a.nrt, nks = a.nks;
    // ---- LDS carve: coefficients | id map | T | partial tiles
    cplx *coef = (cplx *)smem;                                                   // [ncoef + 1]
    short *idm = (short *)(smem + (((size_t)(a.ncoef + 1) * 16 + 15) & ~(size_t)15));   // [Mp][Mp]
    unsigned char *Tb = (unsigned char *)idm + (((size_t)Mp * Mp * 2 + 15) & ~(size_t)15);   // [nks][1024]
    unsigned char *Pb = Tb + (size_t)nks * 1024;                                  // [nrt][2][4][64] doubles (re / im)
    const cplx *vc = a.vcoef + (long)w * (a.ncoef + 1);
    for (int i = tid; i <= a.ncoef; i += UF_NT) coef[i] = vc[i];
    for (int i = tid; i < Mp * Mp / 2; i += UF_NT) ((int *)idm)[i] = ((const int *)a.elem_id)[i];
    cplx *phi = a.phi + (long)w * M * nt;
    // ---- units of this wave: u = wave and wave + 8; unit u: half kh = u / nrt of the contraction, row tile rt = u % nrt.
    // The kh == 0 unit of a row tile owns that tile of the running sum S and of T.
    const int nunits = 2 * nrt;
    const int u0 = wave, u1 = wave + 8;
    const bool has0 = u0 < nunits, has1 = u1 < nunits;
    const int rt0 = has0 ? u0 % nrt : 0, kh0 = has0 ? u0 / nrt : 0;
    const int rt1 = has1 ? u1 % nrt : 0, kh1 = has1 ? u1 / nrt : 0;
    const int ksplit = (nks + 1) >> 1;                           // k-steps [0, ksplit) and [ksplit, nks)
    // element (row 4 r + lk of the tile, column lr) of a tile in accumulator layout <-> T entry (k-step rt * 4 + r, lane)
    const bool col_ok = lr < nt;
    const int spin = lr < a.na ? 0 : 1;
    // S tile(s) this wave owns (kh == 0 units): at most two (u0 always kh == 0 when it exists and wave < nrt; u1 = wave + 8
    // is a kh == 0 unit only when nrt > 8, which M <= 112 excludes)
    const bool own = has0 && kh0 == 0;
    d4_t Sr = {0, 0, 0, 0}, Si = {0, 0, 0, 0};
    // ---- T_0 = B phi (row scaling), S = T_0
    if (own) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = rt0 * 16 + 4 * r + lk;
            cplx v = cmake(0.0, 0.0);
            if (k < M && col_ok) v = cmul(a.bdiag[spin * M + k], phi[(long)k * nt + lr]);
            Sr[r] = v.x; Si[r] = v.y;
            *(d2_t *)(Tb + (size_t)(rt0 * 4 + r) * 1024 + lane * 16) = (d2_t){v.x, v.y};
        }
    }
    __syncthreads();
    const unsigned coef_l = lds_addr(coef), id_l = lds_addr(idm), t_l = lds_addr(Tb);
    // one unit: rows rt * 16 .. + 15 of V times T over k-steps [k0, k1): returns the (re, im) partial tile
    auto unit = [&](const int rt, const int k0, const int k1, d4_t &Cr, d4_t &Ci) __attribute__((always_inline)) {
        d4_t P1 = {0, 0, 0, 0}, P2 = {0, 0, 0, 0}, P3 = {0, 0, 0, 0};
        const unsigned idrow = id_l + (unsigned)((rt * 16 + lr) * Mp + lk) * 2;      // id[row][4 ks + lk]
        // software pipeline, depth 2: the id of k-step ks + 2 and the coefficient of ks + 1 are in flight under the MFMAs of ks
        auto ld_id = [&](int ks) -> int {
            int v;
            asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(idrow + (unsigned)ks * 8));
            return v;
        };
        int idn = 0, idnn = 0;
        d2_t an = {0, 0}, bn = {0, 0};
        idn = ld_id(k0);
        if (k0 + 1 < k1) idnn = ld_id(k0 + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        an = lds_read_b128(coef_l + (unsigned)idn * 16);
        bn = lds_read_b128(t_l + (unsigned)k0 * 1024 + lane * 16);
        for (int ks = k0; ks < k1; ++ks) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const d2_t av = an, bv = bn;
            const int idc = idnn;
            if (ks + 1 < k1) {
                an = lds_read_b128(coef_l + (unsigned)idc * 16);
                bn = lds_read_b128(t_l + (unsigned)(ks + 1) * 1024 + lane * 16);
            }
            if (ks + 2 < k1) idnn = ld_id(ks + 2);
            P1 = mfma16(av[0], bv[0], P1);
            P2 = mfma16(av[1], bv[1], P2);
            P3 = mfma16(av[0] + av[1], bv[0] + bv[1], P3);
        }
        Cr = P1 - P2;
        Ci = P3 - P1 - P2;
    };
    for (int n = 1; n <= a.order; ++n) {
        d4_t C0r = {0, 0, 0, 0}, C0i = {0, 0, 0, 0}, C1r, C1i;
        if (has0) unit(rt0, kh0 ? ksplit : 0, kh0 ? nks : ksplit, C0r, C0i);
        if (has1) {
            unit(rt1, kh1 ? ksplit : 0, kh1 ? nks : ksplit, C1r, C1i);
            // (u1 is always a second-half unit here: park the partial tile)
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

}  // namespace

void k_ueg_fast_free(afq_handle *h) {
    UegFast *f = uf_of(h);
    if (!f) return;
    for (void *p : {(void *)f->elem_id, (void *)f->coef_q, (void *)f->coef_v, (void *)f->psic_rows, (void *)f->fb_idx,
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
    std::vector<int> qa(mm, -1), qb(mm, -1);
    std::vector<double> va(2 * mm, 0.0), vb(2 * mm, 0.0);
    bool ok = M <= 112;
    for (int pass = 0; pass < 2 && ok; ++pass) {
        const int64_t *cp = pass ? Bcp : Acp, *row = pass ? Brow : Arow;
        const double *val = pass ? Bval : Aval;
        std::vector<int> &q_ = pass ? qb : qa;
        std::vector<double> &v_ = pass ? vb : va;
        for (int q = 0; q < nq && ok; ++q)
            for (int64_t z = cp[q]; z < cp[q + 1]; ++z) {
                const int64_t e = row[z];
                if (e < 0 || e >= mm || q_[e] >= 0) { ok = false; break; }       // a second entry for this element
                q_[e] = q; v_[2 * e] = val[2 * z]; v_[2 * e + 1] = val[2 * z + 1];
            }
    }
    if (!ok) return AFQ_OK;              // general kernels
    typedef std::tuple<int, double, double, int, double, double> Key;
    std::map<Key, int> ids;
    std::vector<int> cq;
    std::vector<cplx> cv;
    const int Mp = ((M + 15) / 16) * 16;
    std::vector<short> eid((size_t)Mp * Mp, (short)-1);
    for (long e = 0; e < mm; ++e) {
        if (qa[e] < 0 && qb[e] < 0) continue;
        const Key key(qa[e], va[2 * e], va[2 * e + 1], qb[e], vb[2 * e], vb[2 * e + 1]);
        auto it = ids.find(key);
        int id;
        if (it == ids.end()) {
            id = (int)ids.size();
            if (id >= UF_MAX_COEF) return AFQ_OK;
            ids.emplace(key, id);
            cq.push_back(qa[e]); cq.push_back(qb[e]);
            cv.push_back(cmake(va[2 * e], va[2 * e + 1])); cv.push_back(cmake(vb[2 * e], vb[2 * e + 1]));
        } else id = it->second;
        eid[(size_t)(e / M) * Mp + (e % M)] = (short)id;
    }
    f->ncoef = (int)ids.size(); f->Mp = Mp;
    for (short &s : eid) if (s < 0) s = (short)f->ncoef;            // the zero coefficient
    int rc;
    if ((rc = upload_vec(h, &f->elem_id, eid))) return rc;
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
    std::vector<std::vector<std::pair<int, cplx>>> lists(K);
    int L = 0;
    for (int n = 0; n < K; ++n) {
        const bool isB = n >= nq;
        const int q = isB ? n - nq : n;
        const std::vector<int64_t> &cp = isB ? f->Bcp : f->Acp, &row = isB ? f->Brow : f->Arow;
        const std::vector<double> &val = isB ? f->Bval : f->Aval;
        for (int64_t z = cp[q]; z < cp[q + 1]; ++z) {
            const int i = (int)(row[z] / M), j = (int)(row[z] % M);
            if (slot[i] >= 0) lists[n].emplace_back(slot[i] * M + j, cmake(val[2 * z], val[2 * z + 1]));
        }
        L = std::max(L, (int)lists[n].size());
    }
    if (L == 0) L = 1;
    f->fb_len = L;
    std::vector<int> fi((size_t)L * K, 0);
    std::vector<cplx> fv((size_t)L * K, cmake(0.0, 0.0));
    for (int n = 0; n < K; ++n)
        for (size_t k = 0; k < lists[n].size(); ++k) { fi[k * K + n] = lists[n][k].first; fv[k * K + n] = lists[n][k].second; }
    int rc;
    if ((rc = upload_vec(h, &f->psic_rows, pr))) return rc;
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
    const size_t lds1 = sizeof(cplx) * ((size_t)f->nrows * h->M + h->K);
    return lds1 <= 150 * 1024;
}

// force bias + fields + HS coefficients in one launch (replaces k_full_G, k_vbias_ueg, k_xbar_fields, k_vhs_ueg)
int k_ueg_fields(afq_handle *h) {
    UegFast *f = uf_of(h);
    if (f->vcoef_nw != h->nw || !f->vcoef) {
        if (f->vcoef) hipFree(f->vcoef);
        f->vcoef = nullptr;
        AFQ_HIP(h, hipMalloc(&f->vcoef, sizeof(cplx) * (size_t)h->nw * (f->ncoef + 1)));
        f->vcoef_nw = h->nw;
    }
    FieldRng rng = FieldRng();
    if (h->rng_inline) {
        rng.on = 1; rng.seed = h->rng_seed; rng.stream = h->rng_stream; rng.counter = h->rng_inline_counter;
        rng.weight = h->weight; rng.alive_out = h->alive;
        h->rng_inline = false;
    }
    UegFieldArgs a;
    a.M = h->M; a.nt = h->nt; a.K = h->K; a.nq = h->nq; a.nrows = f->nrows; a.fb_len = f->fb_len; a.ncoef = f->ncoef;
    a.sqrt_dt = h->sqrt_dt; a.ghalf = h->ghalf; a.psic_rows = f->psic_rows; a.fb_idx = f->fb_idx; a.fb_val = f->fb_val;
    a.coef_q = f->coef_q; a.coef_v = f->coef_v; a.xi = h->xi; a.mf = h->mf_shift; a.xbar = h->xbar; a.xs = h->xs;
    a.cmf = h->cmf; a.cfb = h->cfb; a.vcoef = f->vcoef; a.counters = h->counters; a.alive = h->alive;
    a.force_bias = (h->flags & AFQ_PROP_FORCE_BIAS) ? 1 : 0;
    const size_t lds = sizeof(cplx) * ((size_t)f->nrows * h->M + h->K);
    static size_t lds_set[AFQ_MAX_DEVICES] = {0};
    AFQ_HIP(h, afq_raise_lds((const void *)ueg_fields_kernel, lds, lds_set));
    AFQ_LAUNCH(h, ueg_fields_kernel, dim3(h->nw), dim3(UF_NT), lds, h->stream, a, rng);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_prop_ueg(afq_handle *h) {
    UegFast *f = uf_of(h);
    PropUegArgs a;
    a.M = h->M; a.na = h->na; a.nb = h->nb; a.nt = h->nt; a.order = h->exp_order; a.ncoef = f->ncoef; a.Mp = f->Mp;
    a.nrt = f->Mp / 16; a.nks = f->Mp / 4;
    a.elem_id = f->elem_id; a.vcoef = f->vcoef; a.bdiag = f->bdiag; a.phi = h->phi; a.alive = h->alive;
    const size_t lds = (((size_t)(f->ncoef + 1) * 16 + 15) & ~(size_t)15) + (((size_t)f->Mp * f->Mp * 2 + 15) & ~(size_t)15) +
                       (size_t)a.nks * 1024 + (size_t)a.nrt * 4096;
    static size_t lds_set[AFQ_MAX_DEVICES] = {0};
    AFQ_HIP(h, afq_raise_lds((const void *)prop_ueg_kernel, lds, lds_set));
    KernelTrace kt(h, AFQ_K_PROPAGATOR);
    // matrix-pipe flops: order products x row tiles x k-steps x 3 multiplications x 2048
    h->issued_flops[AFQ_K_PROPAGATOR] = 3.0 * h->exp_order * a.nrt * a.nks * 2048.0 * h->nw;
    AFQ_LAUNCH(h, prop_ueg_kernel, dim3(h->nw), dim3(UF_NT), lds, h->stream, a);
    AFQ_POST(h);
    return AFQ_OK;
}
