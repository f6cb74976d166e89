// Walker-batched dense contractions on the fp64 MFMA tile engine:
//   one-body propagator  phi_s <- BH1[s] phi_s        (propagation/operations.py:29-52)
//   force bias           X_s[w,n] = sum_q Ghalf_s[w,q] rchol_s[q,n]   (propagation/generic.py:130-152)
//   HS potential         VHS[w] = i sqrt(dt) reshape(hs_pot xs[w])    (propagation/generic.py:164-179)
//   Taylor propagator    phi <- sum_{n<=order} VHS^n/n! phi           (propagation/continuous.py:82-111)
//   full Green's fn      G_s = conj(psi_s) Ghalf_s                    (walkers/single_det.py:312,319)
#include <cstdlib>
#include "mfma_gemm_wg.h"

// Pick the (TM, TN) register tiling that needs the fewest whole rounds of the
// chip's 1024 SIMDs (one wave-task per SIMD per round), with a small bias
// towards larger tiles (more MFMAs per fragment load).
struct TileChoice { int tm, tn; };
static TileChoice pick_tiles(int batch, int rows, int cols, const TileChoice *cand, int ncand) {
    double best = 1e300; TileChoice bc = cand[0];
    for (int c = 0; c < ncand; ++c) {
        const long tasks = mfma_gemm_tasks(batch, rows, cols, cand[c].tm, cand[c].tn);
        const long rounds = (tasks + 1023) / 1024;
        const double cost = (double)rounds * (cand[c].tm * cand[c].tn + 0.35 * (cand[c].tm + cand[c].tn));
        if (cost < best) { best = cost; bc = cand[c]; }
    }
    return bc;
}
#define DISPATCH_TILES(h, p, tc, MAP, WPB)                                                         \
    do {                                                                                           \
        hipError_t e__;                                                                            \
        typedef decltype(p) P__;                                                                   \
        afq_note_launch((h), __func__);                                                            \
        if (tc.tm == 1 && tc.tn == 1) e__ = launch_mfma_gemm<1, 1, P__, MAP>(p, (h)->stream, WPB);      \
        else if (tc.tm == 1 && tc.tn == 2) e__ = launch_mfma_gemm<1, 2, P__, MAP>(p, (h)->stream, WPB); \
        else if (tc.tm == 2 && tc.tn == 1) e__ = launch_mfma_gemm<2, 1, P__, MAP>(p, (h)->stream, WPB); \
        else if (tc.tm == 2 && tc.tn == 2) e__ = launch_mfma_gemm<2, 2, P__, MAP>(p, (h)->stream, WPB); \
        else if (tc.tm == 1 && tc.tn == 4) e__ = launch_mfma_gemm<1, 4, P__, MAP>(p, (h)->stream, WPB); \
        else if (tc.tm == 2 && tc.tn == 4) e__ = launch_mfma_gemm<2, 4, P__, MAP>(p, (h)->stream, WPB); \
        else e__ = launch_mfma_gemm<2, 2, P__, MAP>(p, (h)->stream, WPB);                               \
        if (e__ == hipSuccess) e__ = afq_post_launch(h);                                           \
        AFQ_HIP(h, e__);                                                                           \
    } while (0)
static const TileChoice kCplxTiles[] = {{2, 2}, {2, 1}, {1, 2}, {1, 1}};
static const TileChoice kMixedTiles[] = {{2, 4}, {2, 2}, {1, 4}, {1, 2}, {1, 1}};

// ---------------------------------------------------------------- one body
// AR: BH1 has no imaginary part (real hopping / kinetic matrix and real mean-field terms: checked at
// afq_set_propagator) -- two real multiplications per element pair instead of three
template <bool AR>
struct OneBodyProbT {
    static constexpr bool A_CPLX = true, B_CPLX = true, A_REAL = AR;
    int batch, rows, cols, kdim;     // batch = nw, rows = M, cols = ns, kdim = M
    int nt, off;                     // phi row stride, first column of this spin
    const cplx *B1;                  // BH1[s]  [M, M]
    const cplx *src;                 // phi     [nw, M, nt]
    cplx *dst;
    const int *alive;
    // optional: row p of walker b's product is multiplied by rowscale[b * rs_stride + p] on the way out (the diagonal
    // Taylor propagator of the Hubbard HS potential folded into the one-body product ahead of it)
    const cplx *rowscale;
    long rs_stride;
    // closed-shell walkers on the large-system path (round 6; afq_internal.h: closed_large): closed_w[b] != 0 says walker b's
    // spin blocks are bitwise equal; the work-group tiles that lie wholly in its beta columns (col0 >= na_cols) are not
    // computed -- every column of the product depends on its own input column only, and the propagated alpha block is
    // copied over the beta block at the end of the step (closed_copy_beta_kernel)
    const int *closed_w;
    int na_cols;
    static constexpr bool BTILE_SKIP = true;
    __device__ bool skip_tile_b(int b, int, int col0) const { return closed_w && col0 >= na_cols && closed_w[b] != 0; }
    __device__ bool active(int b) const { return alive[b] != 0; }
    // dead walkers are not propagated (qmc/afqmc.py:232) but source and destination are ping-pong buffers: their columns
    // are copied through by the work-groups / waves that would have multiplied them (a separate copy launch before)
    static constexpr bool INACTIVE_COPY = true;
    __device__ void inactive_tile(int b, int row0, int nr, int col0, int nc, int t, int nthr) const {
        for (int e = t; e < nr * nc; e += nthr) {
            const int r = row0 + e / nc, c = col0 + e % nc;
            if (r < rows && c < cols) {
                const long idx = ((long)b * rows + r) * nt + off + c;
                dst[idx] = src[idx];
            }
        }
    }
    __device__ cplx loadA(int, int row, int k) const { return B1[(long)row * kdim + k]; }
    __device__ cplx loadB(int b, int k, int col) const {
        return src[((long)b * kdim + k) * nt + off + col];
    }
    __device__ const cplx *ptrA(int, int row, int k) const { return B1 + (long)row * kdim + k; }
    __device__ const cplx *ptrB(int b, int k, int col) const { return src + ((long)b * kdim + k) * nt + off + col; }
    static constexpr bool INCR = true;       // incremental refill of the ring engine
    __device__ int klimit(int) const { return kdim; }
    __device__ const cplx *baseA(int, int row) const { return B1 + (long)row * kdim; }
    __device__ const cplx *baseB(int b, int col) const { return src + (long)b * kdim * nt + off + col; }
    __device__ long kstepA() const { return 1; }
    __device__ long kstepB(int) const { return nt; }
    __device__ bool rowok(int, int) const { return true; }
    __device__ bool colok(int, int) const { return true; }
    __device__ void store(int b, int row, int col, double re, double im) const {
        cplx v = cmake(re, im);
        if (rowscale) v = cmul(rowscale[b * rs_stride + row], v);
        dst[((long)b * rows + row) * nt + off + col] = v;
    }
};

typedef OneBodyProbT<false> OneBodyProb;

// s = 0 / 1: the columns of one spin; s = 2: both spins in one launch (BH1[0] == BH1[1], one row-scale set)
template <bool AR>
static int onebody_spin(afq_handle *h, int s, const cplx *rowscale) {
    const int M = h->M;
    const int ns = s == 2 ? h->nt : s == 0 ? h->na : h->nb;
    OneBodyProbT<AR> p;
    p.batch = h->nw; p.rows = M; p.cols = ns; p.kdim = M;
    p.nt = h->nt; p.off = s == 1 ? h->na : 0;
    if (s == 2) s = 0;
    p.B1 = h->BH1 + (long)s * M * M;
    p.src = h->phi; p.dst = h->phi_t; p.alive = h->alive;
    // rowscale: [nw, nv, M] factors; spin s takes its own row of them when there are two (spin decomposition)
    p.rowscale = rowscale ? rowscale + (h->nv == 2 ? (long)s * M : 0L) : nullptr;
    p.rs_stride = (long)h->nv * M;
    // (only the launch over the columns of both spins can leave the beta tiles of closed-shell walkers out)
    p.closed_w = (h->closed_large && p.off == 0 && ns == h->nt && !rowscale) ? h->closed_w : nullptr;
    p.na_cols = h->na;
    if (!h->no_ring && M > 64 && M <= 128 && ns > 16 && ns <= 32 && h->nw >= 64) {
        // one work-group = one walker-spin: BH1 and phi fragments through the LDS ring once
        AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 1, 2, 2, 4, OneBodyProbT<AR>, MAP_COLS_FAST, true>(p, h->stream, h->zero_page)));
    } else if (!h->no_ring && M > 128 && ns > 32 && h->nw >= 64) {
        // large systems: 128 x 64 or 64 x 128 work-group tiles, whichever pads the output less (M = 400, 100 columns:
        // 448 x 128 against 512 x 128), 3M complex products (2 real ones when BH1 is real); the 64 x 128 shape runs the
        // half-chunk pipelined loop (measured at C5 on the Taylor product: 813 -> 683 us)
        const long padA = (long)((M + 127) / 128) * 128 * ((ns + 63) / 64) * 64, padB = (long)((M + 63) / 64) * 64 * ((ns + 127) / 128) * 128;
        // 128 x 128 tiles (two 16-row and four 16-column MFMA tiles per wave, pipelined loop) when they pad no more than the
        // smaller shapes: twice the MFMAs per chunk and barrier (C4, 256 x 256 per walker: 302 -> 272 us, 1.84 -> 1.77 ms per step)
        const long padC = (long)((M + 127) / 128) * 128 * ((ns + 127) / 128) * 128;
#ifdef AFQ_TUNING
        const int ocfg = AFQ_KNOB_INT("AFQ_OB_CFG", 0);
        if (ocfg == 1) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 4, 4, OneBodyProbT<AR>, MAP_COLS_FAST, true, 1, 3>(p, h->stream, h->zero_page)));
        else if (ocfg == 2) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 1, 1, 8, 4, OneBodyProbT<AR>, MAP_COLS_FAST, true, 1, 3>(p, h->stream, h->zero_page)));
        else if (ocfg == 3) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, OneBodyProbT<AR>, MAP_COLS_FAST, true, 1, 3>(p, h->stream, h->zero_page)));
        else if (AFQ_KNOB_SET("AFQ_OB_NOLOADER")) {
            if (padC <= padA && padC <= padB) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 2, 4, 4, OneBodyProbT<AR>, MAP_COLS_FAST, true, 1, 2>(p, h->stream, h->zero_page)));
            else if (padB <= padA) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 4, 4, OneBodyProbT<AR>, MAP_COLS_FAST, true, 1, 2>(p, h->stream, h->zero_page)));
            else AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 2, 2, 4, OneBodyProbT<AR>, MAP_COLS_FAST, true>(p, h->stream, h->zero_page)));
        }
        else if (AFQ_KNOB_SET("AFQ_OB_NOLEAN")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, OneBodyProbT<AR>, MAP_COLS_FAST, true, 1, 3>(p, h->stream, h->zero_page)));
        else
#endif
        (void)padA, (void)padB, (void)padC;
        // round 4: 64 x 64 tiles on four compute waves (2 x 2 MFMA tiles each) + four loader waves that do nothing but the
        // ring refill (STAG = 3): C4 (256 x 256 per walker, real BH1) 338 -> 300 us against the 128 x 128 tiles above, C5
        // sizes (400 x 100) 444 -> 384 us against 64 x 128; the small tile also pads least
        // round 5: a complex BH1 (3-multiplication products: 142 VGPRs, one work-group per CU) runs the lean loop at 122 VGPRs
        // so that two work-groups share a CU (see k_apply_exponential); the real one (97 VGPRs) is two per CU as it is
        if constexpr (!AR) {
            if (p.closed_w) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, OneBodyProbT<AR>, MAP_COLTILE_SLOW, true, 1, 5, 4>(p, h->stream, h->zero_page)));
            else AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, OneBodyProbT<AR>, MAP_COLS_FAST, true, 1, 5, 4>(p, h->stream, h->zero_page)));
        } else {
            // (closed-shell walkers: column tile slowest, see k_apply_exponential)
            if (p.closed_w) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, OneBodyProbT<AR>, MAP_COLTILE_SLOW, true, 1, 3>(p, h->stream, h->zero_page)));
            else AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, OneBodyProbT<AR>, MAP_COLS_FAST, true, 1, 3>(p, h->stream, h->zero_page)));
        }
    } else {
        OneBodyProb q;          // (small shapes: the register engine, which has no real-operand variant)
        q.batch = p.batch; q.rows = p.rows; q.cols = p.cols; q.kdim = p.kdim; q.nt = p.nt; q.off = p.off;
        q.B1 = p.B1; q.src = p.src; q.dst = p.dst; q.alive = p.alive; q.rowscale = p.rowscale; q.rs_stride = p.rs_stride;
        q.closed_w = nullptr; q.na_cols = 0;
        const TileChoice tc = pick_tiles(q.batch, q.rows, q.cols, kCplxTiles, 4);
        DISPATCH_TILES(h, q, tc, MAP_COLS_FAST, 4);
    }
    return AFQ_OK;
}

int k_onebody(afq_handle *h, const cplx *rowscale) {
    const int M = h->M;
    // both spins share the matrix (and the row-scale set, if any): one launch over all na + nb columns pads the column
    // tiles once instead of twice (M = 400, 50 + 50 columns: 448 x 128 against 2 x 512 x 64)
    // ... and on small systems (M <= 64: the register engine, launch-latency bound) one launch instead of two
    const bool merged = h->bh1_same && h->na > 0 && h->nb > 0 && (!rowscale || h->nv == 1) &&
                        ((M > 128 && h->nt > 32 && h->nw >= 64 && !h->no_ring) || M <= 64);
    for (int s = merged ? 2 : 0; s < (merged ? 3 : 2); ++s) {
        if (s < 2 && (s == 0 ? h->na : h->nb) == 0) continue;
        const int rc = h->bh1_real ? onebody_spin<true>(h, s, rowscale) : onebody_spin<false>(h, s, rowscale);
        if (rc) return rc;
    }
    // (dead walkers are not propagated, qmc/afqmc.py:232: the products above copied their phi through)
    cplx *t = h->phi; h->phi = h->phi_t; h->phi_t = t;
    return AFQ_OK;
}

// -------------------------------------------------------------- force bias
// rows = walkers, cols = fields, contraction over q = (i, p) of one spin,
// split into `nsplit` slices; partial sums go to vbias[(slice*2 + s), w, n].
#define FB_MAX_BATCH 32
template <bool RC>
struct ForceBiasProb {
    static constexpr bool A_CPLX = true, B_CPLX = RC;
    int batch, rows, cols, kdim;     // batch = 2*nsplit, rows = nw, cols = K, kdim = longest slice
    long astride;                    // elements between walkers in ghalf
    int K;
    long ldr;                        // leading dimension of rre / rim
    // per batch: first contraction index inside [0, nt*M) and slice length (host-computed)
    long q0[FB_MAX_BATCH];
    int len[FB_MAX_BATCH];
    const cplx *ghalf;               // [nw, nt*M]
    const double *rre, *rim;         // [nt*M, K]
    cplx *out;                       // [batch, nw, K]
    __device__ bool active(int) const { return true; }
    __device__ cplx loadA(int b, int row, int k) const {
        if (k >= len[b]) return cmake(0.0, 0.0);
        return ghalf[row * astride + q0[b] + k];
    }
    __device__ cplx loadB(int b, int k, int col) const {
        if (k >= len[b]) return cmake(0.0, 0.0);
        const long idx = (q0[b] + k) * ldr + col;
        return cmake(rre[idx], RC ? rim[idx] : 0.0);
    }
    // ring engine (real rchol only): slices are zero-padded by clamping k to the slice
    __device__ const cplx *ptrA(int b, int row, int k) const {
        return k < len[b] ? ghalf + row * astride + q0[b] + k : zero;
    }
    __device__ const double *ptrB(int b, int k, int col) const {
        return k < len[b] ? rre + (q0[b] + k) * ldr + col : (const double *)zero;
    }
    // incremental refill of the ring engine (real rchol)
    static constexpr bool INCR = !RC;
    __device__ int klimit(int b) const { return len[b]; }
    __device__ const cplx *baseA(int b, int row) const { return ghalf + row * astride + q0[b]; }
    __device__ const double *baseB(int b, int col) const { return rre + q0[b] * ldr + col; }
    __device__ long kstepA() const { return 1; }
    __device__ long kstepB(int) const { return ldr; }
    __device__ bool rowok(int, int) const { return true; }
    __device__ bool colok(int, int) const { return true; }
    const cplx *zero;
    int imag_pass;                   // second pass of a complex rchol on the real-B engine: out += i (A . Im B)
    __device__ void store(int b, int row, int col, double re, double im) const {
        cplx *o = out + ((long)b * rows + row) * K + col;
        if (imag_pass) { const cplx t = *o; *o = cmake(t.x - im, t.y + re); }
        else *o = cmake(re, im);
    }
};

template <bool RC>
static void fill_force_bias(ForceBiasProb<RC> &p, afq_handle *h) {
    const int nsplit = h->fb_split, M = h->M;
    p.batch = 2 * nsplit; p.rows = h->nw; p.cols = h->K; p.K = h->K;
    p.astride = (long)h->nt * M;
    p.ldr = h->ld_rc;
    int kmax = 0;
    for (int b = 0; b < p.batch; ++b) {
        const int s = b & 1, sl = b >> 1;
        const long tot = (long)(s == 0 ? h->na : h->nb) * M;
        const long per = (tot + nsplit - 1) / nsplit;
        long l = tot - sl * per; if (l > per) l = per; if (l < 0) l = 0;
        p.q0[b] = (long)(s ? h->na : 0) * M + sl * per;
        p.len[b] = (int)l;
        if (l > kmax) kmax = (int)l;
    }
    p.kdim = kmax;
    p.ghalf = h->ghalf; p.rre = h->rchol_re; p.rim = h->rchol_im; p.out = h->vbias;
    p.zero = (const cplx *)h->zero_page;
    p.imag_pass = 0;
}

__global__ void ghalf_sum_kernel(const cplx *ghalf, cplx *out, long half, long n) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const long w = e / half, q = e % half;
    const cplx a = ghalf[w * 2 * half + q], b = ghalf[w * 2 * half + half + q];
    out[e] = cmake(a.x + b.x, a.y + b.y);
}

// Both spins contract with the SAME half-rotated Cholesky block (RHF-type trial: rchol_same, checked bitwise at upload):
// sum_q R[q,k] Ga[q] + sum_q R[q,k] Gb[q] = sum_q R[q,k] (Ga + Gb)[q] -- half the contraction.  The 2 * nsplit output
// partials keep their layout (every consumer sums all of them): they become 2 * nsplit slices of the one contraction.
bool k_fb_use_sum(afq_handle *h) {
    return h->rchol_same && h->rchol_real && h->ndet == 1 && h->na == h->nb && h->nw > 32 && !h->no_ring &&
           !AFQ_KNOB_SET("AFQ_FB_NOSUM");
}

static int force_bias_generic_impl(afq_handle *h);

// The Coulomb vectors of the energy evaluation at the end of step n and the force bias at the start of step n + 1 are
// the same contraction of the same (cached) Ghalf: the second call is skipped while ghalf_version has not moved.
int k_force_bias_generic(afq_handle *h) {
    // (multi-determinant trial: one set of partials, and one version, per determinant -- the Coulomb vectors of an energy
    //  evaluation serve the force bias of the next step as they do for one determinant)
    unsigned long long &ver = h->ndet == 1 ? h->vbias_version : h->dets[h->cur_det].vbias_version;
    if (ver == h->ghalf_version && !AFQ_KNOB_SET("AFQ_FB_NOREUSE")) return AFQ_OK;
    const int rc = force_bias_generic_impl(h);
    ver = rc == AFQ_OK ? h->ghalf_version : 0;
    return rc;
}

// every determinant's partials are current (left behind by the energy evaluation on the same Green's functions)
bool k_msd_vbias_current(afq_handle *h) {
    if (h->ndet <= 1 || AFQ_KNOB_SET("AFQ_FB_NOREUSE")) return false;
    for (int d = 0; d < h->ndet; ++d) if (h->dets[d].vbias_version != h->ghalf_version) return false;
    return true;
}

static int force_bias_generic_impl(afq_handle *h) {
    if (2 * h->fb_split > FB_MAX_BATCH) AFQ_FAIL(h, AFQ_EINVAL, "force-bias split too large");
    if (h->rchol_real) {
        ForceBiasProb<false> p;
        fill_force_bias(p, h);
        if (k_fb_use_sum(h)) {
            const long half = (long)h->na * h->M, n = half * h->nw;
            if (!h->ghalf_sum) { AFQ_HIP(h, hipMalloc(&h->ghalf_sum, sizeof(cplx) * (size_t)n)); h->gsum_version = 0; }
            if (h->gsum_version != h->ghalf_version) {      // not written by the Green's function kernel itself
                AFQ_LAUNCH(h, ghalf_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->ghalf,
                           h->ghalf_sum, half, n);
                AFQ_POST(h);
                h->gsum_version = h->ghalf_version;
            }
            const int ns2 = 2 * h->fb_split;
            const long per = (half + ns2 - 1) / ns2;
            int kmax = 0;
            for (int b = 0; b < ns2; ++b) {
                long l = half - b * per; if (l > per) l = per; if (l < 0) l = 0;
                p.q0[b] = b * per; p.len[b] = (int)l;
                if (l > kmax) kmax = (int)l;
            }
            p.kdim = kmax; p.ghalf = h->ghalf_sum; p.astride = half;
        }
        const TileChoice tc = pick_tiles(p.batch, p.rows, p.cols, kMixedTiles, 5);
        if (h->nw > 32 && !h->no_ring) {
            // work-group tile 64 walkers x 64 fields (cfg 2), operands shared through the LDS ring.  Measured at C3
            // together with the reduction of the split-K partial sums in fields_kernel (step time, us):
            // 64x128 tile / 16 slices 553.7, 64x64 / 8 slices 545.4, 32x64 / 8 slices 545.3, 64x64 / 4 slices 553.6
#ifdef AFQ_TUNING
            static const int cfg = AFQ_KNOB_INT("AFQ_FB_CFG", 2);
            static const int kc = AFQ_KNOB_INT("AFQ_GEMM_KC", 1);
            if (cfg == 1) {
                KernelTrace kt(h, AFQ_K_FORCE_BIAS);
                if (kc == 2) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 4, 2, 2, 4, ForceBiasProb<false>, MAP_BATCH_XCD, false, 2>(p, h->stream, h->zero_page)));
                else AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 4, 2, 2, 4, ForceBiasProb<false>, MAP_BATCH_XCD>(p, h->stream, h->zero_page)));
            }
            else if (cfg == 3) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 2, 4, ForceBiasProb<false>, MAP_BATCH_XCD>(p, h->stream, h->zero_page)));
            else if (cfg != 2) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ForceBiasProb<false>, MAP_BATCH_XCD>(p, h->stream, h->zero_page)));
            else
#endif
            {
                KernelTrace kt(h, AFQ_K_FORCE_BIAS);
                h->issued_flops[AFQ_K_FORCE_BIAS] = mfma_gemm_wg_issued_flops<4, 2, 1, 2, ForceBiasProb<false>>(
                    p, [&](int, int, int) -> long { return p.kdim; });
                // half-chunk pipelined loop (STAG = 2): 56 us at C3 against 58 (staggered halves) / 61 (plain loop)
#ifdef AFQ_TUNING
                if (AFQ_KNOB_SET("AFQ_GEMM_NOSTAG")) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 2, 4, ForceBiasProb<false>, MAP_BATCH_XCD>(p, h->stream, h->zero_page)));
                else if (AFQ_KNOB_SET("AFQ_GEMM_STAG1")) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 2, 4, ForceBiasProb<false>, MAP_BATCH_XCD, false, 1, 1>(p, h->stream, h->zero_page)));
                else if (AFQ_KNOB_SET("AFQ_FB_LOADER")) {
                    const int v = AFQ_KNOB_INT("AFQ_FB_LOADER", 0);
                    if (v == 2) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, ForceBiasProb<false>, MAP_BATCH_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
                    else if (v == 3) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 1, 1, 4, 4, ForceBiasProb<false>, MAP_BATCH_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
                    else AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 2, 4, ForceBiasProb<false>, MAP_BATCH_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
                }
                else if (AFQ_KNOB_SET("AFQ_FB_NOLOADER")) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 2, 4, ForceBiasProb<false>, MAP_BATCH_XCD, false, 1, 2>(p, h->stream, h->zero_page)));
                else
#endif
                // round 4: the ring refill on eight loader waves of its own (STAG = 3; see the HS-potential GEMM): 32.9 -> 31.8 us
                AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 2, 4, ForceBiasProb<false>, MAP_BATCH_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
            }
        } else {
            DISPATCH_TILES(h, p, tc, MAP_BATCH_XCD, 4);
        }
    } else if (h->nw > 32 && !h->no_ring) {
        // complex half-rotated Cholesky vectors (complex trial): rchol is stored planar, so the real-B ring
        // engine runs twice, out = A . Re(rchol) then out += i A . Im(rchol)
        ForceBiasProb<false> p;
        fill_force_bias(p, h);
        for (int pass = 0; pass < 2; ++pass) {
            p.imag_pass = pass;
            p.rre = pass ? h->rchol_im : h->rchol_re;
            AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 4, 2, 2, 4, ForceBiasProb<false>, MAP_BATCH_XCD>(p, h->stream, h->zero_page)));
        }
    } else {
        ForceBiasProb<true> p;
        fill_force_bias(p, h);
        const TileChoice tc = pick_tiles(p.batch, p.rows, p.cols, kCplxTiles, 4);
        DISPATCH_TILES(h, p, tc, MAP_BATCH_XCD, 4);
    }
    return AFQ_OK;
}

// ------------------------------------------ multi-determinant force bias through the determinant-averaged G
// The reference contracts the Cholesky vectors with ONE matrix per walker (propagation/generic.py:154-157,
// walkers/multi_det.py:283-290):  vbias_n = sum_pq L_n[p,q] Gbar[p,q],  Gbar = sum_d w_d (G_d,a + G_d,b) / sum_d w_d,
// G_d = conj(psi_d) Ghalf_d.  One contraction per determinant with its half-rotated vectors (k_force_bias_generic) costs
// ndet (x 2 for complex vectors) x K nt M products; through Gbar it is K M (M + 1) / 2 for symmetric L_n -- the mirror image
// of the HS-potential GEMM, on the same packed columns -- plus the build of Gbar:
//   1. msd_scale_ghalf_kernel: gs[w] = [ (w_d / sum w) Ghalf_d[w] ]_d stacked over the determinants, [ndet nt, M]
//   2. GbarSymProb: S = Gbar + Gbar^T on the upper triangle only, as ONE product with a doubled contraction,
//        S[p,q] = sum_k X[k,p] Y[k,q],  X = [conj(psi)^T ; gs],  Y = [gs ; conj(psi)^T]   (k over 2 ndet nt),
//      strictly lower work-group tiles skipped, the diagonal halved, stored straight into the packed columns
//   3. ForceBiasProb with A = S [nw, P] and B = the packed hs_pot with the field index contiguous [P, K]
__global__ void msd_scale_ghalf_kernel(const cplx *__restrict__ ghalf_all, const cplx *__restrict__ detw,
                                       cplx *__restrict__ gs, int nw, int ndet, long per, long wstride) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int w = blockIdx.y, d = blockIdx.z;
    if (e >= per) return;
    cplx tot = cmake(0.0, 0.0);
    for (int dd = 0; dd < ndet; ++dd) tot = cadd(tot, detw[(long)w * ndet + dd]);
    const cplx wd = detw[(long)w * ndet + d];
    // (a skipped determinant -- weight exactly 0, msd_combine_kernel -- contributes zeros, whatever its Ghalf holds)
    const bool dead = wd.x == 0.0 && wd.y == 0.0;
    gs[(long)w * wstride + (long)d * per + e] = dead ? cmake(0.0, 0.0) : cmul(cdiv(wd, tot), ghalf_all[((long)d * nw + w) * per + e]);
}

// out[c][n] = in[n][c]: the packed hs_pot^T [K, ld_in] -> [P, ld_out]
__global__ void transpose_f64_kernel(const double *__restrict__ in, double *__restrict__ out, int nrow_in, long ncol_in,
                                     long ld_in, long ld_out) {
    __shared__ double t[32][33];
    const long c0 = (long)blockIdx.x * 32;
    const int r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const long c = c0 + threadIdx.x; const int r = r0 + j;
        t[j][threadIdx.x] = (r < nrow_in && c < ncol_in) ? in[(long)r * ld_in + c] : 0.0;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const long c = c0 + j; const int r = r0 + threadIdx.x;
        if (c < ncol_in && r < nrow_in) out[c * ld_out + r] = t[threadIdx.x][j];
    }
}

struct GbarSymProb {
    static constexpr bool A_CPLX = true, B_CPLX = true, TILE_SKIP = true;
    int batch, rows, cols, kdim;     // nw, M, M, 2 KK
    int KK, M;
    long ldS;
    const cplx *psicT;               // [KK, M]
    const cplx *gs;                  // [nw, KK, M]
    cplx *S;                         // [nw, ldS]
    __device__ bool active(int) const { return true; }
    __device__ bool skip_tile(int row0, int col0) const { return row0 > col0; }
    __device__ const cplx *ptrA(int b, int row, int k) const {
        return k < KK ? psicT + (long)k * M + row : gs + ((long)b * KK + (k - KK)) * M + row;
    }
    __device__ const cplx *ptrB(int b, int k, int col) const {
        return k < KK ? gs + ((long)b * KK + k) * M + col : psicT + (long)(k - KK) * M + col;
    }
    __device__ cplx loadA(int b, int row, int k) const { return *ptrA(b, row, k); }
    __device__ cplx loadB(int b, int k, int col) const { return *ptrB(b, k, col); }
    // incremental refill of the ring engine: two affine segments, [0, KK) and [KK, 2 KK) (KK is a multiple of 8:
    // the stacks are zero-padded to it)
    static constexpr bool INCR = true, INCR_SEG = true;
    __device__ int klimit(int) const { return kdim; }
    __device__ int kseg() const { return KK; }
    __device__ const cplx *baseA(int, int row) const { return psicT + row; }
    __device__ const cplx *baseB(int b, int col) const { return gs + (long)b * KK * M + col; }
    __device__ const cplx *baseA2(int b, int row) const { return gs + (long)b * KK * M + row; }
    __device__ const cplx *baseB2(int, int col) const { return psicT + col; }
    __device__ long kstepA() const { return M; }
    __device__ long kstepB(int) const { return M; }
    __device__ bool rowok(int, int) const { return true; }
    __device__ bool colok(int, int) const { return true; }
    __device__ void store(int b, int row, int col, double re, double im) const {
        if (row > col) return;
        const double f = row == col ? 0.5 : 1.0;
        S[(long)b * ldS + (long)row * M - (long)row * (row - 1) / 2 + (col - row)] = cmake(f * re, f * im);
    }
};

static double msd_fb_cost_per_det(const afq_handle *h) {
    return (double)h->ndet * (h->rchol_real ? 1.0 : 2.0) * h->K * (double)h->nt * h->M;
}
static double msd_fb_cost_gbar(const afq_handle *h) {
    // real-by-complex products; a 3-multiplication complex product counts 1.5
    return (double)h->K * h->M * (h->M + 1) / 2.0 + 1.5 * (double)h->M * h->M * h->ndet * h->nt;
}

bool k_msd_gbar_wanted(afq_handle *h) {
    if (h->ndet <= 1 || h->kind != AFQ_SYS_GENERIC || !h->hs_sym || h->no_ring || !h->msd_psicT) return false;
    if (2 * h->fb_split > FB_MAX_BATCH) return false;
    if (h->msd_fb_mode == 1) return false;
    if (h->msd_fb_mode == 2) return true;
    return h->nw > 32 && 1.2 * msd_fb_cost_gbar(h) < msd_fb_cost_per_det(h);
}

int k_force_bias_msd_gbar(afq_handle *h) {
    const int M = h->M, KK = (h->ndet * h->nt + 7) & ~7, nw = h->nw;     // (stacks zero-padded to whole chunks of 8)
    const long P = (long)M * (M + 1) / 2, per = (long)h->nt * M;
    h->dets[0].vbias_version = 0;                  // the averaged partials go where determinant 0 keeps its own
    if (!h->hs_pk) {
        AFQ_HIP(h, hipMalloc(&h->hs_pk, sizeof(double) * (size_t)P * h->ld_rc));
        AFQ_HIP(h, hipMemsetAsync(h->hs_pk, 0, sizeof(double) * (size_t)P * h->ld_rc, h->stream));
        AFQ_LAUNCH(h, transpose_f64_kernel, dim3((unsigned)((P + 31) / 32), (unsigned)((h->K + 31) / 32)), dim3(32, 8), 0,
                   h->stream, h->hs_pot, h->hs_pk, h->K, P, h->ld_hs, h->ld_rc);
        AFQ_POST(h);
    }
    if (!h->msd_gs) {
        AFQ_HIP(h, hipMalloc(&h->msd_gs, sizeof(cplx) * (size_t)nw * KK * M));
        AFQ_HIP(h, hipMemsetAsync(h->msd_gs, 0, sizeof(cplx) * (size_t)nw * KK * M, h->stream));             // (the pad rows)
    }
    if (!h->msd_S) {
        AFQ_HIP(h, hipMalloc(&h->msd_S, sizeof(cplx) * (size_t)nw * h->ld_hs));
        AFQ_HIP(h, hipMemsetAsync(h->msd_S, 0, sizeof(cplx) * (size_t)nw * h->ld_hs, h->stream));   // (the pad column)
    }
    AFQ_LAUNCH(h, msd_scale_ghalf_kernel, dim3((unsigned)((per + 255) / 256), nw, h->ndet), dim3(256), 0, h->stream,
               h->ghalf_all, h->detw, h->msd_gs, nw, h->ndet, per, (long)KK * M);
    AFQ_POST(h);
    {
        GbarSymProb p;
        p.batch = nw; p.rows = M; p.cols = M; p.kdim = 2 * KK; p.KK = KK; p.M = M; p.ldS = h->ld_hs;
        p.psicT = h->msd_psicT; p.gs = h->msd_gs; p.S = h->msd_S;
#ifdef AFQ_TUNING
        const int gcfg = AFQ_KNOB_INT("AFQ_GBAR_CFG", 0);
        if (gcfg == 1) AFQ_GEMM_AS(h, "msd_gbar_fold GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GbarSymProb, MAP_BATCH_XCD, true, 1, 3>(p, h->stream, h->zero_page)));
        else if (gcfg == 2) AFQ_GEMM_AS(h, "msd_gbar_fold GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GbarSymProb, MAP_BATCH_XCD_ROWS, true, 1, 3>(p, h->stream, h->zero_page)));
        else if (gcfg == 3) AFQ_GEMM_AS(h, "msd_gbar_fold GEMM", (launch_mfma_gemm_wg<4, 2, 2, 4, 4, GbarSymProb, MAP_BATCH_XCD, true>(p, h->stream, h->zero_page)));
        else if (gcfg == 4) AFQ_GEMM_AS(h, "msd_gbar_fold GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GbarSymProb, MAP_BATCH_XCD, true, 1, 2>(p, h->stream, h->zero_page)));
        else if (gcfg == 5) AFQ_GEMM_AS(h, "msd_gbar_fold GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GbarSymProb, MAP_BATCH_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
        else if (gcfg == 6) AFQ_GEMM_AS(h, "msd_gbar_fold GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GbarSymProb, MAP_COLS_FAST, true, 1, 2>(p, h->stream, h->zero_page)));
        else if (gcfg == 7) AFQ_GEMM_AS(h, "msd_gbar_fold GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GbarSymProb, MAP_COLS_FAST, true, 1, 3>(p, h->stream, h->zero_page)));
        else if (gcfg == 11) AFQ_GEMM_AS(h, "msd_gbar_fold GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GbarSymProb, MAP_COLS_FAST, true, 1, 5, 4>(p, h->stream, h->zero_page)));
        else if (gcfg == 9) AFQ_GEMM_AS(h, "msd_gbar_fold GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GbarSymProb, MAP_COLS_FAST, true, 1, 2, 4>(p, h->stream, h->zero_page)));
        else if (gcfg == 10) AFQ_GEMM_AS(h, "msd_gbar_fold GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GbarSymProb, MAP_COLS_FAST, true, 1, 3, 4>(p, h->stream, h->zero_page)));
        else if (gcfg == 8) AFQ_GEMM_AS(h, "msd_gbar_fold GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 8, GbarSymProb, MAP_BATCH_XCD, true, 1, 2>(p, h->stream, h->zero_page)));
        else
#endif
        if (M > 64)
            // 64 x 64 tiles, four waves of 2 x 2, 3-multiplication products, the half-chunk pipelined loop with the waves'
            // own refill (STAG = 2).  Measured at C5 sizes (400 x 400, contraction 2 x 400, 256 walkers;
            // profiles/r05_c5_gbar_variants.txt): 2.37-2.39 ms; one walker per XCD at a time 2.42-2.48; with loader waves
            // 2.63 (2.75 on the batch map); 4-multiplication products under loader waves 2.88; ring depth 8: 2.97;
            // before the incremental refill (per-fragment address arithmetic every chunk) 3.14
            // (the lean loop with two work-groups per CU, k_apply_exponential: 2.35 -> 2.31 ms)
            AFQ_GEMM_AS(h, "msd_gbar_fold GEMM", (launch_mfma_gemm_wg<2, 2, 2, 2, 4, GbarSymProb, MAP_COLS_FAST, true, 1, 5, 4>(p, h->stream, h->zero_page)));
        else
            AFQ_GEMM_AS(h, "msd_gbar_fold GEMM", (launch_mfma_gemm_wg<2, 2, 1, 1, 4, GbarSymProb, MAP_COLS_FAST, true>(p, h->stream, h->zero_page)));
    }
    {
        ForceBiasProb<false> p;
        fill_force_bias(p, h);
        const int ns2 = 2 * h->fb_split;
        const long slice = (P + ns2 - 1) / ns2;
        int kmax = 0;
        for (int b = 0; b < ns2; ++b) {
            long l = P - b * slice; if (l > slice) l = slice; if (l < 0) l = 0;
            p.q0[b] = b * slice; p.len[b] = (int)l;
            if (l > kmax) kmax = (int)l;
        }
        p.kdim = kmax; p.ghalf = h->msd_S; p.astride = h->ld_hs; p.rre = h->hs_pk; p.rim = nullptr; p.out = h->vbias_all;
        AFQ_GEMM_AS(h, "msd_gbar_contract GEMM", (launch_mfma_gemm_wg<4, 2, 1, 2, 4, ForceBiasProb<false>, MAP_BATCH_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
    }
    return AFQ_OK;
}

// --------------------------------------------------------------------- VHS
// rows = walkers, cols = (p,q) pairs, contraction over fields.
// hsT is hs_pot transposed, [K, M*M], so a B fragment is 128 contiguous bytes.
struct VhsProb {
    static constexpr bool A_CPLX = true, B_CPLX = false;
    int batch, rows, cols, kdim;     // 1, nw, M*M, K
    const cplx *xs;                  // [nw, K]
    const double *hsT;               // [K, ldb]
    long ldb;
    cplx *out;                       // [nw, M*M]
    double sqrt_dt;
    const int *alive;
    // symmetric Cholesky matrices (L_n[p,q] == L_n[q,p], the usual case): only the M(M+1)/2 columns
    // p <= q are contracted and every result is stored at (p,q) and (q,p); pair[col] = (p, q)
    const int2 *pair;
    int M;
    long mm;
    bool mirror;                     // false: only (p,q), p <= q, is stored (consumer: prop_fused_kernel)
    __device__ bool active(int) const { return true; }
    __device__ cplx loadA(int, int row, int k) const { return xs[(long)row * kdim + k]; }
    __device__ cplx loadB(int, int k, int col) const { return cmake(hsT[(long)k * ldb + col], 0.0); }
    __device__ const cplx *ptrA(int, int row, int k) const { return xs + (long)row * kdim + k; }
    __device__ const double *ptrB(int, int k, int col) const { return hsT + (long)k * ldb + col; }
    // incremental refill of the ring engine
    static constexpr bool INCR = true;
    __device__ int klimit(int) const { return kdim; }
    __device__ const cplx *baseA(int, int row) const { return xs + (long)row * kdim; }
    __device__ const double *baseB(int, int col) const { return hsT + col; }
    __device__ long kstepA() const { return 1; }
    __device__ long kstepB(int) const { return ldb; }
    __device__ bool rowok(int, int) const { return true; }
    __device__ bool colok(int, int) const { return true; }
    __device__ void store(int, int row, int col, double re, double im) const {
        // i*sqrt(dt)*(re + i im)
        const cplx v = cmake(-sqrt_dt * im, sqrt_dt * re);
        if (pair) {
            const int2 pq = pair[col];
            cplx *o = out + (long)row * mm;
            o[pq.x * M + pq.y] = v;
            if (mirror && pq.x != pq.y) o[pq.y * M + pq.x] = v;
        } else {
            out[(long)row * mm + col] = v;
        }
    }
};

#ifdef AFQ_TUNING
static void gemm_ts_dump(afq_handle *h, const char *what) {
    static unsigned long long *buf = nullptr;
    static int n = 0;
    if (!AFQ_KNOB_SET("AFQ_GEMM_TS")) return;
    if (!buf) {
        hipMalloc(&buf, (64 * 4 + 64) * 8);
        hipMemset(buf, 0, (64 * 4 + 64) * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(afq_gemm_ts), &buf, sizeof(buf));
    }
    if (++n != 40) return;
    unsigned long long t[64 * 4 + 64];
    hipStreamSynchronize(h->stream);
    hipMemcpy(t, buf, sizeof(t), hipMemcpyDeviceToHost);
    for (int w = 0; w < 64; w += 9)
        fprintf(stderr, "GEMM_TS %s wg %2d: loop %llu ticks = %.2f us of the 100 MHz clock (tick %.2f GHz; %llu chunks, %.0f ticks per chunk)  stores %llu\n",
                what, w, t[4 * w + 1], t[4 * w] * 0.01, t[4 * w] ? t[4 * w + 1] / (t[4 * w] * 10.0) : 0.0, t[4 * w + 3],
                t[4 * w + 3] ? (double)t[4 * w + 1] / t[4 * w + 3] : 0.0, t[4 * w + 2]);
}
#endif

int k_vhs_generic(afq_handle *h) {
#ifdef AFQ_TUNING
    struct Dump { afq_handle *h; ~Dump() { gemm_ts_dump(h, "after VHS"); } } dump_{h};
    // timing ablations of the ring loop (mfma_gemm_wg.h: afq_gemm_abl), for this GEMM only: set and cleared in stream order
    static const int vhs_abl = AFQ_KNOB_INT("AFQ_VHS_ABL", 0);
    static const int abl_zero = 0;
    struct Abl {
        afq_handle *h;
        Abl(afq_handle *h_) : h(h_) { if (vhs_abl) hipMemcpyToSymbolAsync(HIP_SYMBOL(afq_gemm_abl), &vhs_abl, sizeof(int), 0, hipMemcpyHostToDevice, h->stream); }
        ~Abl() { if (vhs_abl) hipMemcpyToSymbolAsync(HIP_SYMBOL(afq_gemm_abl), &abl_zero, sizeof(int), 0, hipMemcpyHostToDevice, h->stream); }
    } abl_guard_{h};
#endif
    VhsProb p;
    p.batch = 1; p.rows = h->nw; p.cols = h->hs_sym ? h->M * (h->M + 1) / 2 : h->M * h->M; p.kdim = h->K;
    p.xs = h->xs; p.hsT = h->hs_pot; p.ldb = h->ld_hs; p.out = h->vhs; p.sqrt_dt = h->sqrt_dt; p.alive = h->alive;
    p.pair = h->hs_sym ? h->hs_pair : nullptr; p.M = h->M; p.mm = (long)h->M * h->M;
    p.mirror = !h->vhs_upper;
    if (h->nw > 32 && !h->no_ring) {
        // work-group tile 64 walkers x 160 (p,q) pairs; hs_pot^T panels shared through the LDS ring
        // measured at C3 (tools/sweep_vhs_cfg.sh): packed symmetric columns 75.8 us with the 32 x 160 tile
        // (cfg 7), 116 us with the 64 x 160 tile that is best for the full M^2 columns (100 us)
#ifdef AFQ_TUNING
        static const int cfg_env = AFQ_KNOB_INT("AFQ_VHS_CFG", -1);
        const int cfg = cfg_env >= 0 ? cfg_env : (h->hs_sym ? 7 : 0);
        if (cfg == 1) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 4, 2, 2, 4, VhsProb, MAP_ROWS_FAST>(p, h->stream, h->zero_page)));
        else if (cfg == 2) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 2, 2, 4, VhsProb, MAP_ROWS_FAST>(p, h->stream, h->zero_page)));
        else if (cfg == 3) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 4, 4, VhsProb, MAP_ROWS_FAST>(p, h->stream, h->zero_page)));
        else if (cfg == 4) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 4, 4, VhsProb, MAP_ROWS_FAST>(p, h->stream, h->zero_page)));
        else if (cfg == 5) AFQ_GEMM(h, (launch_mfma_gemm_wg<1, 2, 2, 5, 4, VhsProb, MAP_ROWS_FAST>(p, h->stream, h->zero_page)));
        else if (cfg == 6) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 1, 2, 5, 4, VhsProb, MAP_ROWS_FAST>(p, h->stream, h->zero_page)));
        else if (cfg == 7) {
            static const int kc = AFQ_KNOB_INT("AFQ_GEMM_KC", 1);
            KernelTrace kt(h, AFQ_K_VHS);
            static const int xmap = AFQ_KNOB_INT("AFQ_VHS_XCD", 0);   // measured: 77.8 vs 75.8 us
            if (AFQ_KNOB_SET("AFQ_GEMM_PIPE")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 4, VhsProb, MAP_ROWS_FAST, false, 1, 2>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_VHS_LOADER")) {
                const int v = AFQ_KNOB_INT("AFQ_VHS_LOADER", 0);
                if (v == 2) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 4, VhsProb, MAP_COLPANEL_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (v == 3) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 8, VhsProb, MAP_ROWS_FAST, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (v == 4) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 8, VhsProb, MAP_COLPANEL_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (v == 5) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 1, 1, 5, 4, VhsProb, MAP_ROWS_FAST, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (v == 6) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 1, 1, 5, 8, VhsProb, MAP_COLPANEL_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (v == 9) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 1, 1, 5, 4, VhsProb, MAP_COLPANEL_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (v == 10) AFQ_GEMM(h, (launch_mfma_gemm_wg<1, 2, 1, 5, 4, VhsProb, MAP_COLPANEL_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (v == 11) AFQ_GEMM(h, (launch_mfma_gemm_wg<1, 1, 1, 5, 4, VhsProb, MAP_COLPANEL_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (v == 12) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 1, 1, 5, 4, VhsProb, MAP_ROWS_FAST, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (v == 13) AFQ_GEMM(h, (launch_mfma_gemm_wg<1, 2, 2, 5, 4, VhsProb, MAP_COLPANEL_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (v == 14) AFQ_GEMM(h, (launch_mfma_gemm_wg<1, 4, 2, 5, 4, VhsProb, MAP_COLPANEL_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
                else if (v == 7) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 4, VhsProb, MAP_COLPANEL_XCD, false, 1, 5, 4>(p, h->stream, h->zero_page)));
                else if (v == 8) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 4, VhsProb, MAP_ROWS_FAST, false, 1, 5, 4>(p, h->stream, h->zero_page)));
                else AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 4, VhsProb, MAP_ROWS_FAST, false, 1, 3>(p, h->stream, h->zero_page)));
            }
            else if (AFQ_KNOB_SET("AFQ_VHS_D8")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 8, VhsProb, MAP_ROWS_FAST>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_VHS_D8P")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 8, VhsProb, MAP_ROWS_FAST, false, 1, 2>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_VHS_RREG")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 4, VhsProb, MAP_ROWS_FAST, false, 1, 4>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_VHS_RREG8")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 8, VhsProb, MAP_ROWS_FAST, false, 1, 4>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_VHS_D8X")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 8, VhsProb, MAP_COLPANEL_XCD, false, 1, 2>(p, h->stream, h->zero_page)));
            else
            if (kc == 2) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 4, VhsProb, MAP_ROWS_FAST, false, 2>(p, h->stream, h->zero_page)));
            else if (xmap) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 4, VhsProb, MAP_COLPANEL_XCD>(p, h->stream, h->zero_page)));
            else if (AFQ_KNOB_SET("AFQ_VHS_NOLOADER")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 4, VhsProb, MAP_ROWS_FAST>(p, h->stream, h->zero_page)));
            // Round 4: four LOADER waves beside the four compute waves (STAG = 3) and the column panels pinned to XCDs.
            // Timing ablations of the plain loop (tools/vhs_ablate.sh, cycles per chunk of 8 contraction indices at C3):
            // 1938 as it was = 1385 for the 20 MFMAs alone + 790 for refill, fragment reads and barrier alone, of which
            // only 240 overlapped -- an LDS-DMA instruction holds its wave's instruction issue for 100+ cycles, and a wave
            // that is alone on its SIMD idles the matrix pipe meanwhile, wherever in the loop the refill sits (the
            // pipelined loop: 1882).  With the refill on waves of its own: 1611, 59.0 -> 52.2 us.  The XCD map makes the
            // eight row tiles of a column panel share one L2: HBM-side fetch 160 -> 38 MB per launch
            // (profiles/r04_vhs_variants.txt), no effect on the time by itself.
            else AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 4, VhsProb, MAP_COLPANEL_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
            h->issued_flops[AFQ_K_VHS] = mfma_gemm_wg_issued_flops<2, 2, 1, 5, VhsProb>(p, [&](int, int, int) -> long { return p.kdim; });
        }
        else if (cfg == 8) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 1, 1, 5, 4, VhsProb, MAP_ROWS_FAST>(p, h->stream, h->zero_page)));
        else if (cfg == 9) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 3, 4, VhsProb, MAP_ROWS_FAST>(p, h->stream, h->zero_page)));
        else if (cfg == 10) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 3, 4, VhsProb, MAP_ROWS_FAST>(p, h->stream, h->zero_page)));
        else {
            KernelTrace kt(h, AFQ_K_VHS);
            h->issued_flops[AFQ_K_VHS] = mfma_gemm_wg_issued_flops<2, 2, 2, 5, VhsProb>(p, [&](int, int, int) -> long { return p.kdim; });
            AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 5, 4, VhsProb, MAP_ROWS_FAST>(p, h->stream, h->zero_page)));
        }
#else
        // (the variants measured against these two are in tuning builds only: NEGATIVES.md, profiles/r04_vhs_variants.txt)
        if (h->hs_sym) {
            // packed symmetric columns: 32 walkers x 160 (p,q) pairs, four compute + four loader waves (STAG = 3), column
            // panels pinned to XCDs
            KernelTrace kt(h, AFQ_K_VHS);
            AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 1, 5, 4, VhsProb, MAP_COLPANEL_XCD, false, 1, 3>(p, h->stream, h->zero_page)));
            h->issued_flops[AFQ_K_VHS] = mfma_gemm_wg_issued_flops<2, 2, 1, 5, VhsProb>(p, [&](int, int, int) -> long { return p.kdim; });
        } else {
            KernelTrace kt(h, AFQ_K_VHS);
            h->issued_flops[AFQ_K_VHS] = mfma_gemm_wg_issued_flops<2, 2, 2, 5, VhsProb>(p, [&](int, int, int) -> long { return p.kdim; });
            AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 5, 4, VhsProb, MAP_ROWS_FAST>(p, h->stream, h->zero_page)));
        }
#endif
        return AFQ_OK;
    }
    static const TileChoice cand[] = {{2, 4}, {2, 2}, {1, 4}, {1, 2}};       // (2 x 5 does not fit two fragment sets)
    const TileChoice tc = pick_tiles(p.batch, p.rows, p.cols, cand, 4);
    const int tiles_m = (p.rows + 16 * tc.tm - 1) / (16 * tc.tm);
    int wpb = tiles_m <= 8 ? (tiles_m < 1 ? 1 : tiles_m) : 8;
    const long ntask = mfma_gemm_tasks(p.batch, p.rows, p.cols, tc.tm, tc.tn);
    while (wpb > 1 && ntask / wpb < 200) wpb >>= 1;      // keep >= ~1 workgroup per CU
    DISPATCH_TILES(h, p, tc, MAP_ROWS_FAST, wpb);
    return AFQ_OK;
}

// ------------------------------------------------------------ Taylor series
struct TaylorProb {
    static constexpr bool A_CPLX = true, B_CPLX = true;
    int batch, rows, cols, kdim;     // nw, M, ncols, M
    int nt, off;
    long vstride;                    // elements between walkers in vhs
    const cplx *vhs;                 // [nw, (nv,) M, M] (already offset to the spin's matrix)
    const cplx *tin;                 // [nw, M, nt]
    cplx *tout, *phi;
    double inv_n;
    const int *alive;
    const int *closed_w;             // closed-shell walkers: see OneBodyProbT
    int na_cols;
    static constexpr bool BTILE_SKIP = true;
    __device__ bool skip_tile_b(int b, int, int col0) const { return closed_w && col0 >= na_cols && closed_w[b] != 0; }
    __device__ bool active(int b) const { return alive[b] != 0; }
    __device__ cplx loadA(int b, int row, int k) const { return vhs[b * vstride + (long)row * kdim + k]; }
    __device__ cplx loadB(int b, int k, int col) const { return tin[((long)b * kdim + k) * nt + off + col]; }
    __device__ const cplx *ptrA(int b, int row, int k) const { return vhs + b * vstride + (long)row * kdim + k; }
    __device__ const cplx *ptrB(int b, int k, int col) const { return tin + ((long)b * kdim + k) * nt + off + col; }
    static constexpr bool INCR = true;       // incremental refill of the ring engine
    __device__ int klimit(int) const { return kdim; }
    __device__ const cplx *baseA(int b, int row) const { return vhs + b * vstride + (long)row * kdim; }
    __device__ const cplx *baseB(int b, int col) const { return tin + (long)b * kdim * nt + off + col; }
    __device__ long kstepA() const { return 1; }
    __device__ long kstepB(int) const { return nt; }
    __device__ bool rowok(int, int) const { return true; }
    __device__ bool colok(int, int) const { return true; }
    __device__ void store(int b, int row, int col, double re, double im) const {
        const long idx = ((long)b * rows + row) * nt + off + col;
        const cplx t = cmake(re * inv_n, im * inv_n);
        tout[idx] = t;
        const cplx o = phi[idx];
        phi[idx] = cmake(o.x + t.x, o.y + t.y);
    }
};

int k_apply_exponential(afq_handle *h, const cplx *vhs) {
    const int M = h->M;
    const long per = (long)M * h->nt;
    AFQ_HIP(h, hipMemcpyAsync(h->phi_t, h->phi, sizeof(cplx) * per * h->nw, hipMemcpyDeviceToDevice, h->stream));
    cplx *tin = h->phi_t, *tout = h->phi_t2;
    for (int n = 1; n <= h->exp_order; ++n) {
        for (int s = 0; s < h->nv; ++s) {
            TaylorProb p;
            p.batch = h->nw; p.rows = M; p.kdim = M; p.nt = h->nt;
            if (h->nv == 1) { p.cols = h->nt; p.off = 0; }
            else { p.cols = s == 0 ? h->na : h->nb; p.off = s == 0 ? 0 : h->na; }
            if (p.cols == 0) continue;
            p.vstride = (long)h->nv * M * M;
            p.vhs = vhs + (long)s * M * M;
            p.tin = tin; p.tout = tout; p.phi = h->phi; p.inv_n = 1.0 / n; p.alive = h->alive;
            p.closed_w = (h->closed_large && h->nv == 1) ? h->closed_w : nullptr;
            p.na_cols = h->na;
            if (!h->no_ring && M > 64 && M <= 128 && p.cols > 32 && p.cols <= 64 && h->nw >= 64) {
                // one work-group (8 waves, 128 x 64 tile) = one walker: VHS[w] and T[w] pass the LDS ring once
                AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 2, 2, 4, TaylorProb, MAP_COLS_FAST, true>(p, h->stream, h->zero_page)));
                continue;
            }
            if (!h->no_ring && M > 128 && p.cols > 32 && h->nw >= 64) {
                // large systems: 128 x 64 work-group tiles, 3M complex products
#ifdef AFQ_TUNING
                const int tcfg = AFQ_KNOB_INT("AFQ_TAYLOR_CFG", 0);
                bool done = true;
                if (AFQ_KNOB_SET("AFQ_BIG_PIPE")) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 2, 2, 4, TaylorProb, MAP_COLS_FAST, true, 1, 2>(p, h->stream, h->zero_page)));
                else if (tcfg == 1) AFQ_GEMM(h, (launch_mfma_gemm_wg<8, 1, 1, 7, 4, TaylorProb, MAP_COLS_FAST, true>(p, h->stream, h->zero_page)));
                else if (tcfg == 2) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 4, 4, TaylorProb, MAP_COLS_FAST, true>(p, h->stream, h->zero_page)));
                else if (tcfg == 3) AFQ_GEMM(h, (launch_mfma_gemm_wg<8, 1, 1, 7, 2, TaylorProb, MAP_COLS_FAST, true>(p, h->stream, h->zero_page)));
                else if (tcfg == 4) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 4, 4, TaylorProb, MAP_COLS_FAST, true, 1, 2>(p, h->stream, h->zero_page)));
                else if (tcfg == 5) AFQ_GEMM(h, (launch_mfma_gemm_wg<8, 1, 1, 7, 4, TaylorProb, MAP_COLS_FAST, true, 1, 2>(p, h->stream, h->zero_page)));
                else if (tcfg == 6) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 4, 4, TaylorProb, MAP_COLS_FAST, true, 1, 3>(p, h->stream, h->zero_page)));
                else if (tcfg == 7) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 1, 1, 8, 4, TaylorProb, MAP_COLS_FAST, true, 1, 3>(p, h->stream, h->zero_page)));
                else if (tcfg == 8) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, TaylorProb, MAP_COLS_FAST, true, 1, 3>(p, h->stream, h->zero_page)));
                else {
                    // 128 x 64 or 64 x 128 tiles, whichever pads the M x ncols output less; the 64 x 128 shape with the
                    // half-chunk pipelined loop (C5, 400 x 100: 813 -> 683 us per product)
                    const long padA = (long)((M + 127) / 128) * 128 * ((p.cols + 63) / 64) * 64;
                    const long padB = (long)((M + 63) / 64) * 64 * ((p.cols + 127) / 128) * 128;
                    if (AFQ_KNOB_SET("AFQ_TAYLOR_NOLOADER")) {
                        if (padB <= padA) AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 1, 4, 4, TaylorProb, MAP_COLS_FAST, true, 1, 2>(p, h->stream, h->zero_page)));
                        else AFQ_GEMM(h, (launch_mfma_gemm_wg<4, 2, 2, 2, 4, TaylorProb, MAP_COLS_FAST, true>(p, h->stream, h->zero_page)));
                    }
                    // round 4: 64 x 64 tiles, four compute + four loader waves (STAG = 3; see k_vhs_generic): C5 sizes 689 -> 627 us
                    else if (AFQ_KNOB_SET("AFQ_TAYLOR_LEAN")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, TaylorProb, MAP_COLS_FAST, true, 1, 5, 4>(p, h->stream, h->zero_page)));
                    else if (AFQ_KNOB_SET("AFQ_TAYLOR_LEAN1")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, TaylorProb, MAP_COLS_FAST, true, 1, 5, 1>(p, h->stream, h->zero_page)));
                    else if (AFQ_KNOB_SET("AFQ_TAYLOR_WPE")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, TaylorProb, MAP_COLS_FAST, true, 1, 3, 4>(p, h->stream, h->zero_page)));
                    else if (AFQ_KNOB_SET("AFQ_TAYLOR_WPE2")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, TaylorProb, MAP_COLS_FAST, true, 1, 2, 4>(p, h->stream, h->zero_page)));
                    else if (AFQ_KNOB_SET("AFQ_TAYLOR_S2")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, TaylorProb, MAP_BATCH_XCD, true, 1, 2>(p, h->stream, h->zero_page)));
                    else if (AFQ_KNOB_SET("AFQ_TAYLOR_S2C")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, TaylorProb, MAP_COLS_FAST, true, 1, 2>(p, h->stream, h->zero_page)));
                    else if (AFQ_KNOB_SET("AFQ_TAYLOR_XCD")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, TaylorProb, MAP_BATCH_XCD, true, 1, 3>(p, h->stream, h->zero_page)));
                    else if (AFQ_KNOB_SET("AFQ_TAYLOR_LOADER")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, TaylorProb, MAP_COLS_FAST, true, 1, 3>(p, h->stream, h->zero_page)));
                    // round 5: the same 64 x 64 tiles from four waves that refill the ring themselves inside the half-chunk
                    // pipelined loop (STAG = 2): 627-633 -> 612-616 us (C5 sizes).  Forcing two work-groups per CU
                    // (128 VGPRs, WPE = 4) spills 65-98 registers into the chunk loop: 2102 us
                    else if (AFQ_KNOB_SET("AFQ_TAYLOR_S2DEF")) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, TaylorProb, MAP_COLS_FAST, true, 1, 2>(p, h->stream, h->zero_page)));
                    else done = false;
                }
                if (done) continue;
#endif
                {
                    // ... and TWO work-groups per CU: the 3-multiplication kernels of this engine hold 142-160 VGPRs, i.e. one
                    // work-group of 4 + 4 waves per CU.  Forcing 128 registers on the pipelined loops spills (2102 us);
                    // the lean loop of STAG = 5 -- loader waves, compute waves that read the fragments of ONE sub-step at a
                    // time into one set of registers -- needs 122, and what its own waves no longer overlap the second
                    // work-group does: 613 -> 589 us
                    // (closed-shell walkers: with the column tile as the fast index the tiles that are left out are every
                    //  other work-group, i.e. every other XCD gets nothing but work-groups that return at once -- the column tile as the slowest index of the launch)
                    if (p.closed_w) AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, TaylorProb, MAP_COLTILE_SLOW, true, 1, 5, 4>(p, h->stream, h->zero_page)));
                    else AFQ_GEMM(h, (launch_mfma_gemm_wg<2, 2, 2, 2, 4, TaylorProb, MAP_COLS_FAST, true, 1, 5, 4>(p, h->stream, h->zero_page)));
                }
                continue;
            }
            const TileChoice tc = pick_tiles(p.batch, p.rows, p.cols, kCplxTiles, 4);
            // one workgroup = every tile of one walker (its VHS and T panels stay in one L1/L2)
            const long per = mfma_gemm_tasks(1, p.rows, p.cols, tc.tm, tc.tn);
            const int wpb = per <= 8 ? (int)per : 4;
            DISPATCH_TILES(h, p, tc, MAP_COLS_FAST, wpb);
        }
        cplx *t = tin; tin = tout; tout = t;
    }
    return AFQ_OK;
}

// ------------------------------------------------------------------ full G
struct FullGProb {
    static constexpr bool A_CPLX = true, B_CPLX = true;
    int batch, rows, cols, kdim;     // nw, M, M, ns
    int nt, off, spin, M;
    const cplx *psi;                 // [M, nt]
    long psi_stride;                 // 0 or M*nt (per-walker trial)
    const cplx *ghalf;               // [nw, nt, M]
    cplx *G;                         // [nw, 2, M, M]
    __device__ bool active(int) const { return true; }
    __device__ cplx loadA(int b, int row, int k) const { return cconj(psi[b * psi_stride + (long)row * nt + off + k]); }
    __device__ cplx loadB(int b, int k, int col) const { return ghalf[((long)b * nt + off + k) * M + col]; }
    __device__ void store(int b, int row, int col, double re, double im) const {
        G[(((long)b * 2 + spin) * M + row) * M + col] = cmake(re, im);
    }
};

int k_full_G(afq_handle *h) {
    const int M = h->M;
    for (int s = 0; s < 2; ++s) {
        const int ns = s == 0 ? h->na : h->nb;
        if (ns == 0) {
            AFQ_HIP(h, hipMemset2DAsync(h->G + (long)s * M * M, sizeof(cplx) * 2 * M * M, 0,
                                        sizeof(cplx) * M * M, h->nw, h->stream));
            continue;
        }
        FullGProb p;
        p.batch = h->nw; p.rows = M; p.cols = M; p.kdim = ns; p.nt = h->nt;
        p.off = s == 0 ? 0 : h->na; p.spin = s; p.M = M;
        p.psi = h->psi; p.psi_stride = h->psi_stride; p.ghalf = h->ghalf; p.G = h->G;
        const TileChoice tc = pick_tiles(p.batch, p.rows, p.cols, kCplxTiles, 4);
        DISPATCH_TILES(h, p, tc, MAP_COLS_FAST, 4);
    }
    return AFQ_OK;
}
