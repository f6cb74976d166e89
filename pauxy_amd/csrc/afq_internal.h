// Internal declarations shared by the translation units of libafqmc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../include/afqmc_hip.h"

typedef double2 cplx;   // (x = re, y = im), same bytes as numpy complex128

__host__ __device__ inline cplx cmake(double r, double i) { return make_double2(r, i); }
__host__ __device__ inline cplx cadd(cplx a, cplx b) { return cmake(a.x + b.x, a.y + b.y); }
__host__ __device__ inline cplx csub(cplx a, cplx b) { return cmake(a.x - b.x, a.y - b.y); }
__host__ __device__ inline cplx cmul(cplx a, cplx b) {
    return cmake(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__host__ __device__ inline cplx cconj(cplx a) { return cmake(a.x, -a.y); }
__host__ __device__ inline cplx cscale(cplx a, double s) { return cmake(a.x * s, a.y * s); }
// a += b*c
__host__ __device__ inline void cfma(cplx &a, cplx b, cplx c) {
    a.x = fma(b.x, c.x, a.x); a.x = fma(-b.y, c.y, a.x);
    a.y = fma(b.x, c.y, a.y); a.y = fma(b.y, c.x, a.y);
}
__host__ __device__ inline double cabs2(cplx a) { return a.x * a.x + a.y * a.y; }
__device__ inline cplx cdiv(cplx a, cplx b) {
    // Smith's algorithm (what C99/numpy use) to avoid overflow
    if (fabs(b.x) >= fabs(b.y)) {
        double r = b.y / b.x, d = b.x + b.y * r;
        return cmake((a.x + a.y * r) / d, (a.y - a.x * r) / d);
    } else {
        double r = b.x / b.y, d = b.x * r + b.y;
        return cmake((a.x * r + a.y) / d, (a.y * r - a.x) / d);
    }
}

#define AFQ_NSCAL 12
enum { T_GREENS = 0, T_ONEBODY, T_FB, T_VHS, T_EXP, T_OVLP, T_QR, T_ENERGY, T_COUNT };

#define AFQ_NCOUNTERS 8     // afq_counters_ext (include/afqmc_hip.h)

struct afq_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // ---- system
    int kind = 0;
    int M = 0, K = 0, na = 0, nb = 0, nt = 0;
    double ecore = 0.0;
    // generic
    double *hs_pot = nullptr;       // f64, TRANSPOSED: [K, ld_hs] with ld_hs = M*M rounded up to even (zero pad)
    bool hs_sym = false;            // L_n symmetric: hs_pot holds only the columns (p <= q), [K, ld_hs]
    double *L_full = nullptr;       // [K, M, Mp] every L_n as a padded row-major matrix (full-G energy; built on first use)
    int2 *hs_pair = nullptr;        // [M(M+1)/2] (p, q) of every packed column
    long ld_hs = 0, ld_rc = 0;      // leading dimensions of hs_pot^T and of rchol_re/im (K rounded up to even)
    bool rchol_real = true;
    double *rchol_re = nullptr;     // f64 [nt*M, ld_rc]
    double *rchol_im = nullptr;     // f64 [nt*M, ld_rc] or null when real
    double *rchol_frag[2] = {nullptr, nullptr};   // energy-kernel A operand, fragment order, per spin
    double *rchol_frag_im[2] = {nullptr, nullptr};
    // quadratic-form exchange (k_energy.hip): Atil_s [(N_s M), ldq] f64 (c128 when rchol is complex), built on first
    // use; atil[1] == atil[0] when both spins have the same half-rotated vectors (closed-shell trial)
    void *atil[2] = {nullptr, nullptr};
    bool atil_unavailable = false;  // automatic mode: Atil did not fit when it was to be built -> T-intermediate kernel
    bool rchol_same = false;        // alpha and beta blocks of rchol are bitwise equal
    int exx_mode = 0;               // afq_set_exchange_algorithm: 0 auto, 1 T-intermediate (exx_kernel), 2 quadratic form
    cplx *exq_y = nullptr;          // [2 * slices, nw, ceil(N M / 16)] per-tile partial sums of the quadratic form
    size_t exq_y_len = 0;
    cplx *H1 = nullptr;             // [2, M, M]
    cplx *rH1 = nullptr;            // half-rotated H1: [nt, M]; rH1[i][q] = sum_p conj(psi[p,i]) H1_s[p,q]
    // hubbard
    double U = 0.0;
    // ueg
    int nq = 0;
    int64_t nnzA = 0, nnzB = 0;
    int64_t *iA_colptr = nullptr, *iA_row = nullptr; cplx *iA_val = nullptr;
    int64_t *iB_colptr = nullptr, *iB_row = nullptr; cplx *iB_val = nullptr;
    // row-major (CSR over M*M rows) copies for the VHS gather
    int64_t *iA_rowptr = nullptr, *iA_col = nullptr; cplx *iA_rval = nullptr;
    int64_t *iB_rowptr = nullptr, *iB_col = nullptr; cplx *iB_rval = nullptr;
    // column-ELL copy of [iA | iB] for the LDS force-bias kernel: k-th non-zero of column c at [k * 2nq + c]
    int ell_len = 0; int *ell_row = nullptr; cplx *ell_val = nullptr;
    // rows of G the UEG energy gathers touch (the occupied orbitals of the trial): compact index per row or -1
    int ueg_nrows = 0; int *ueg_rmap = nullptr; int *ueg_rows = nullptr;
    // the energy's index lists packed for the thread-per-q kernel: (row of the staged G << 16) | column, int32 offsets
    int *ueg_kp = nullptr, *ueg_pm = nullptr, *ueg_koff = nullptr, *ueg_poff = nullptr;
    int64_t *kpq_off = nullptr, *kpq_i = nullptr, *kpq_kpq = nullptr;
    int64_t *pmq_off = nullptr, *pmq_i = nullptr, *pmq_pmq = nullptr;
    double *vqvec = nullptr; double vol = 1.0; double *H1diag = nullptr;
    void *ueg_fast = nullptr;       // k_ueg.hip: tables of the plane-wave step that needs no M x M intermediate

    // ---- trial
    bool have_trial = false;
    cplx *psi = nullptr;            // [M, nt]
    cplx *psic = nullptr;           // conj(psi) [M, nt] (B operand of the overlap GEMM)
    cplx *psicT = nullptr;          // conj(psi)^T [nt, M]: coalesced reads of the Hubbard force bias (single-determinant upload only)
    long psi_stride = 0;            // elements between per-walker 'trials' (back-propagation only; 0 = shared psi)
    bool psi_real = false;          // every imaginary part of the uploaded trial is exactly zero
    bool psi_closed = false;        // na == nb and the alpha and beta blocks of the (single, shared) trial are bitwise equal
    // Closed-shell populations (round 5).  The Green's function kernel checks every walker's spin blocks for bitwise equality
    // (greens_small_kernel); a walker that fails raises closed_bad to the epoch of that launch.  closed_checked_version is
    // the ghalf_version whose Ghalf comes from such a checked launch (carried through the comb's clones, which copy whole
    // walkers): while it equals ghalf_version, "*closed_bad < closed_epoch" ON THE DEVICE means Ghalf_b == Ghalf_a for every
    // walker, and the exchange energy evaluates one spin (k_energy.hip).  The host never reads the flag.
    unsigned long long *closed_bad = nullptr;
    unsigned long long closed_epoch = 0, closed_checked_version = 0;
    // host-side HINT read with every block's sums (est_publish_kernel): the population held an open-shell walker at the last
    // block end.  launch_exx_quadratic then skips the one-spin-first form (two launches when the population is open: 181
    // against 135 us at C3) for the one-launch two-spin form; results are the same either way, a stale hint costs time only
    unsigned long long closed_epoch_pub = 0;
    bool exx_open_hint = false;
    // Large-system propagation (the GEMM chain of k_onebody / k_apply_exponential: M > 128), round 6: closed_w[w] = 1 when walker
    // w's spin blocks are bitwise equal (closed_flags_kernel, every step, on the walkers themselves: no state to keep valid).
    // While closed_large is set the one-body and Taylor GEMMs leave out the work-group tiles that lie wholly in the beta columns
    // of such a walker; closed_copy_beta_kernel copies the propagated alpha block over the beta block at the end of the step.
    int *closed_w = nullptr;
    int closed_w_n = 0;
    bool closed_large = false;

    // multi-determinant trial: the trial-dependent operands of every determinant;
    // psi / psic / rchol_* / rchol_frag* / rH1 above and ghalf / vbias below are VIEWS of the selected one
    struct DetOps {
        cplx *psi = nullptr, *psic = nullptr, *rH1 = nullptr;
        double *rchol_re = nullptr, *rchol_im = nullptr;
        double *rchol_frag[2] = {nullptr, nullptr}, *rchol_frag_im[2] = {nullptr, nullptr};
        void *atil[2] = {nullptr, nullptr};
        bool rchol_same = false;
        unsigned long long vbias_version = 0;   // ghalf_version this determinant's force-bias partials were contracted from
    };
    int ndet = 1, cur_det = 0;
    std::vector<DetOps> dets;       // size ndet when ndet > 1
    cplx *coeffs = nullptr;         // [ndet] device copy of the CI coefficients
    cplx *detd = nullptr;           // [ndet, nw] per-determinant overlaps <D_d|phi_w>
    cplx *detd_a = nullptr;         // [ndet, nw] their alpha factors det(phi_a^T conj(D_d,a)) (multi_det.py:209 tests it first)
    cplx *det_a_out = nullptr;      // where the Green's function launch in flight leaves the alpha determinants (or null)
    cplx *detw = nullptr;           // [nw, ndet] weights conj(c_d) <D_d|phi_w> of the last evaluation
    cplx *ghalf_all = nullptr;      // owning pointers of the per-determinant slices
    cplx *vbias_all = nullptr;
    cplx *energy_all = nullptr;     // [ndet, nw, 3] per-determinant local energies
    // force bias of a multi-determinant trial through the determinant-averaged Green's function (the reference's own
    // formulation, propagation/generic.py:154-157 + walkers/multi_det.py:283-290; k_gemm.hip: k_force_bias_msd_gbar)
    int msd_fb_mode = 0;            // afq_set_msd_force_bias: 0 auto, 1 one contraction per determinant, 2 averaged G
    bool msd_fb_gbar = false;       // the last force_bias() left ONE already-averaged set of partials in vbias_all
    cplx *msd_psicT = nullptr;      // [ndet nt, M] conj(psi_d)^T of every determinant, stacked
    cplx *msd_gs = nullptr;         // [nw, ndet nt, M] Ghalf_d scaled by w_d / sum_d w_d
    cplx *msd_S = nullptr;          // [nw, ld_hs] (Gbar + Gbar^T) on the packed columns p <= q (the diagonal: Gbar[p,p])
    double *hs_pk = nullptr;        // [M (M + 1) / 2, ld_rc] packed hs_pot with the field index contiguous

    // ---- back-propagation (estimators/back_propagation.py, walkers/stack.py FieldConfig)
    int nbp = 0;                    // field configurations kept per walker (0 = off)
    cplx *bp_hist = nullptr;        // [nw, nbp, K] shifted fields x - xbar of the last steps
    int *bp_n = nullptr;            // [nw] FieldConfig.step
    int *bp_flag = nullptr;         // [nw] set by the weight kernel when this step's fields are recorded
    double *bp_cos = nullptr;       // [nw] running product of the cosine factors
    cplx *bp_ph = nullptr;          // [nw] running product of I / |I|
    cplx *phi_old = nullptr;        // [nw, M, nt] walker at the start of the back-propagation window
    cplx *phi_bp = nullptr;         // [nw, M, nt] back-propagated trial (+ its conjugate behind it)
    cplx *BH1dag = nullptr;         // [2, M, M] BH1^H
    cplx *bp_xs = nullptr;          // [nw, K]
    cplx *bp_est = nullptr;         // [4 + 2 M M]

    // ---- discrete Hirsch propagator (propagation/hubbard.py:12-343)
    bool hirsch = false;
    cplx hs_delta[2][2];            // auxf - 1, [field][spin]
    cplx hs_wfac[2];                // aux_wfac
    cplx hs_auxf[2][2];             // auxf exp(-dt U / 2), [field][spin]
    cplx hs_gamma;                  // arccosh(exp(+-dt U / 2)) (complex for the charge decomposition)
    bool hs_charge = false;
    bool hs_direct = false;         // single_site_update: False -- two_body_direct (hubbard.py:222-275)
    double *hs_fbfac = nullptr;     // [nw] force-bias factor of the direct update
    cplx *hs_oinv = nullptr;        // [nw, 2, nmax, nmax] inverse overlaps O^-1 (= inv_ovlp^T of the reference)
    double *hs_u = nullptr;         // [nw, M] uniforms of the site updates
    int *hs_fields = nullptr;       // [nw, M] chosen fields (-1: not visited)
    int *hs_used = nullptr;         // [nw] uniforms consumed
    int *hs_alive0 = nullptr;       // [nw] walkers the driver propagates this step (|w| > 1e-8)

    // ---- propagator
    bool have_prop = false;
    cplx *BH1 = nullptr;            // [2, M, M]
    double cap_frac = 0.0, cap_total = -1.0;   // afq_set_weight_cap: cap applied inside the weight-update kernel
    bool vhs_upper = false;         // set around build_vhs + k_prop_fused: VHS[w] holds only its upper triangle
    bool bh1_same = false;          // BH1[0] and BH1[1] are bitwise equal
    bool bh1_real = false;          // every imaginary part of BH1 is exactly zero (real trial, real L_n)
    cplx *mf_shift = nullptr;       // [K]
    double dt = 0.0, sqrt_dt = 0.0;
    int exp_order = 6;
    int flags = AFQ_PROP_HYBRID | AFQ_PROP_FORCE_BIAS;
    int nv = 1;                     // VHS matrices per walker (2 for Hubbard spin HS)
    bool vhs_diag = false;          // Hubbard: VHS stored as diagonals [nw, nv, M]

    // ---- walkers
    int nw = 0;
    cplx *phi = nullptr;            // [nw, M, nt]
    cplx *phi_t = nullptr;          // scratch [nw, M, nt] (Taylor term ping)
    cplx *phi_t2 = nullptr;         // scratch [nw, M, nt] (Taylor term pong)
    double *weight = nullptr, *unscaled = nullptr, *detR = nullptr;
    // use_log_shift (walkers/handler.py:45,456-475): the shifts are the same for every walker, log_detR is per walker
    double *log_detR = nullptr;     // [nw]
    bool log_shift_on = false;
    double log_shift = 0.0, detR_shift = 0.0;
    cplx *ot = nullptr, *ehyb = nullptr, *phase = nullptr, *eloc = nullptr;
    cplx *ghalf = nullptr;          // [nw, nt, M]
    cplx *G = nullptr;              // [nw, 2, M, M] (allocated on demand)
    cplx *ovlp_old = nullptr, *ovlp_new = nullptr;   // [nw]
    double *xi = nullptr;           // [nw, K]
    int fb_split = 1;               // contraction slices of the force-bias GEMM
    cplx *vbias = nullptr;          // [fb_split*2, nw, K] partial per-spin Coulomb vectors X_a, X_b
    cplx *ghalf_sum = nullptr;      // [nw, na*M] Ghalf_a + Ghalf_b when both spins share rchol (force bias on half the contraction)
    // every writer of ghalf bumps ghalf_version; ghalf_sum is current when gsum_version equals it (the small Green's
    // function kernel writes the sum itself, otherwise k_force_bias_generic runs ghalf_sum_kernel first)
    unsigned long long ghalf_version = 1, gsum_version = 0;
    unsigned long long vbias_version = 0;       // ghalf_version the force-bias partials in vbias were contracted from
    // Hubbard: diag(G_s) as partial sums over row blocks of Ghalf, written by the Ghalf GEMM itself (k_bigdet.hip);
    // current when gdiag_version == ghalf_version, cloned along by the comb
    cplx *gdiag = nullptr;          // [2 nw, gdiag_parts, M]
    int gdiag_parts = 0;
    unsigned long long gdiag_version = 0;
    cplx *xbar = nullptr, *xs = nullptr;              // [nw, K]
    cplx *cmf = nullptr, *cfb = nullptr;              // [nw]
    cplx *vhs = nullptr;            // [nw, nv, M, M] or [nw, nv, M] when vhs_diag
    cplx *lu_ws = nullptr;          // [nw, N, N] workspace of the generic Green's kernel (N > 128)
    cplx *big_ws = nullptr;         // [2 nw, N, N] overlap matrices / inverses (k_bigdet.hip)
    cplx *big_ws2 = nullptr;        // [2 nw, N, N] second workspace (Cholesky-QR)
    cplx *detm = nullptr;           // [2 nw] determinant mantissas
    int *dete = nullptr;            // [2 nw] determinant exponents
    double *qr_logd = nullptr;      // [2, 2 nw] log det R per Cholesky-QR pass
    int *qr_fail = nullptr;         // [nw] Cholesky breakdown -> Gram-Schmidt fallback
    cplx *energy = nullptr;         // [nw, 3]
    cplx *exx_part = nullptr;       // exchange partial sums
    int64_t exx_part_len = 0;
    double *gfrag = nullptr;        // Ghalf in MFMA fragment order (energy kernel B operand)
    size_t gfrag_bytes = 0;
    bool prop_pending = false;                      // afq_propagate_begin done, afq_propagate_finish outstanding
    double *est_stage = nullptr;                    // mapped host memory: estimator sums + scal[4] + sequence number
    unsigned long long est_seq = 0;
    double scal_cache[AFQ_NSCAL] = {0};   // scal[] as of the last afq_estimates_get_end; stale after a population control
    bool scal_cache_valid = false;
    bool est_pending = false;
    unsigned ktrace_mask = 0;          // bit k: event pairs around the launches of kernel kind k
    std::vector<hipEvent_t> ktrace_ev[AFQ_K_COUNT];   // start/stop pairs
    double issued_flops[AFQ_K_COUNT] = {0, 0, 0, 0, 0};   // matrix-pipe flops of the last launch of each kind (afq_kernel_issued_flops)
    double prop_issued_open = 0.0, prop_issued_closed = 0.0;   // per walker, last fused-propagator launch (afq_propagator_issued_flops)
    int ktrace_used[AFQ_K_COUNT] = {0, 0, 0, 0, 0};
    int ktrace_stride[AFQ_K_COUNT] = {1, 1, 1, 1, 1}, ktrace_seen[AFQ_K_COUNT] = {0, 0, 0, 0, 0};   // every n-th launch of a kind is timed
    // afq_launch_trace: an event pair around EVERY launch, keyed by the launch's breadcrumb name (a profile pass, not
    // something to leave on inside a timed region)
    bool ltrace_on = false, ltrace_open = false;
    std::vector<hipEvent_t> ltrace_ev;
    std::vector<const char *> ltrace_name;
    int *gj_flag = nullptr;         // [2 nw] blocked Gauss-Jordan: 1 = a pivot block was poorly conditioned (k_bigdet.hip)
    cplx *estimates = nullptr;      // [10]
    // Mixed estimator with one_rdm: True (estimators/mixed.py:226-229): G then is per-walker STATE (walker.G: the
    // Green's function the walker last evaluated -- before the step's propagation, or at an energy evaluation),
    // cloned by the comb, and rdm_acc += sum_w weight_w Re G_w with every afq_estimates_update
    bool rdm_on = false;
    double *rdm_acc = nullptr;      // [2, M, M]
    unsigned long long *counters = nullptr;   // [AFQ_NCOUNTERS]
    int *alive = nullptr;           // [nw]
    int *parent_ix = nullptr;       // [nw]
    // [AFQ_NSCAL] device scalars: 0 total weight of the last comb, 1 local pairs (< 0: collapsed), 2 collapse flag (sticky),
    // 3 exchange overflow (sticky), 4 largest transfer between two ranks, 5 comb events, 6 communication error (sticky:
    // a peer's flag never arrived), 7 walkers this rank has sent, 8 bytes this rank has sent (window transport)
    double *scal = nullptr;
    void *pack_tmp = nullptr;
    void *zero_page = nullptr;      // 256 zero bytes: source of out-of-range LDS-DMA loads
    // Ghalf / ovlp_new describe the CURRENT phi of every walker (set by the end-of-step Green's
    // function of afq_propagate, cleared by everything that writes phi, psi or Ghalf)
    bool greens_valid = false;
    bool greens_cache = true;       // AFQ_NO_GREENS_CACHE=1 or a handed-out device pointer turns it off
    bool no_fused = false;          // AFQ_NO_FUSED=1: separate one-body / Taylor launches (A/B runs)
    bool no_vhs_upper = false;      // AFQ_VHS_MIRROR=1: always store both triangles of the HS potential (A/B runs)
    bool no_ring = false;           // AFQ_NO_RING=1: register-prefetch GEMM engine only (A/B runs)

    // afq_propagate -> k_greens: run the step's weight update behind the determinant (greens_small_kernel)
    bool fuse_weight_req = false, fuse_weight_done = false;
    // afq_estimates_fuse_next: the weight update of the next step adds every walker's estimator terms to est_acc[w][6]
    // (per walker: no cross-walker sum, no atomics); the next estimates_kernel launch, or the next fetch, folds them in
    double *est_acc = nullptr;
    bool fuse_est_req = false, est_acc_pending = false;
    // the Green's function cached at the end of a step as overlap + spin sum of Ghalf only (afq_propagate_finish)
    bool gsum_only = false, ghalf_skip_store = false, ghalf_skipped = false;
    cplx fuse_eshift;

    // rng
    uint64_t rng_seed = 0, rng_stream = 0, rng_counter = 0;
    bool rng_inline = false;            // this step's fields are drawn inside fields_kernel (no rng launch)
    uint64_t rng_inline_counter = 0;

    // diagnostics: host-side breadcrumbs of the launches queued on the stream (afq_last_launch)
    const char *crumb_name[64] = {nullptr};     // ring of the last 64 kernel names (string literals)
    const char *crumb_api = "";                 // C-ABI entry point that queued them
    volatile unsigned long long n_launch = 0;   // launches queued so far
    bool debug_sync = false;                    // AFQ_DEBUG_SYNC=1: synchronise + check after every launch
    bool debug_markers = false;                 // AFQ_DEBUG_MARKERS=1: a 1-thread marker kernel behind every launch
    volatile unsigned long long *retired = nullptr;   // pinned, device-mapped: index of the last RETIRED launch

    // library-owned RCCL communicator (afq_comm_init; walkers/handler.py:232,291,313,322, mixed.py:261,273)
    void *comm = nullptr;                       // afq_comm_state (k_comm.hip)

    // timers
    bool timers_on = false;
    double t_ms[T_COUNT] = {0};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_energy_ms = 0.0;
};

#define AFQ_HIP(h, call)                                                        \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) {                                                 \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);       \
            return AFQ_EHIP;                                                    \
        }                                                                       \
    } while (0)

// tuning / A-B switches read from the environment exist only in builds made with -DAFQ_TUNING
// (make TUNING=1); the product library has none of them
// AFQ_KNOB_SET("X"): is switch X set?  AFQ_KNOB_INT("X", d): its integer value, d when unset.  In the product build both
// are constants (false / d), no environment is read, and every kernel variant a switch selects sits inside an
// `#ifdef AFQ_TUNING` region, so the product library does not even contain it.
#ifdef AFQ_TUNING
inline const char *afq_knob(const char *name) { return getenv(name); }
#define AFQ_KNOB_SET(name) (afq_knob(name) != nullptr)
#define AFQ_KNOB_INT(name, dflt) (afq_knob(name) ? atoi(afq_knob(name)) : (dflt))
#else
#define AFQ_KNOB_SET(name) false
#define AFQ_KNOB_INT(name, dflt) (dflt)
#endif

// every kernel launch goes through AFQ_LAUNCH / AFQ_GEMM + AFQ_POST: the name of the kernel is left in the
// handle's breadcrumb ring before the launch, and AFQ_POST checks the launch (and, in the debug modes,
// queues a marker or synchronises so that a failing or hanging kernel is identified by name)
void afq_launch_trace_mark(afq_handle *h, const char *name);      // afq_api.hip
inline void afq_note_launch(afq_handle *h, const char *name) {
    h->crumb_name[h->n_launch & 63] = name;
    h->n_launch = h->n_launch + 1;
    if (h->ltrace_on) afq_launch_trace_mark(h, name);
}
hipError_t afq_post_launch(afq_handle *h);      // afq_api.hip
#define AFQ_LAUNCH(h, kern, ...) do { afq_note_launch((h), #kern); hipLaunchKernelGGL(kern, __VA_ARGS__); } while (0)
#define AFQ_POST(h)                                                              \
    do {                                                                        \
        hipError_t e_ = afq_post_launch(h);                                     \
        if (e_ != hipSuccess) {                                                 \
            (h)->err = std::string("kernel ") + (h)->crumb_name[((h)->n_launch - 1) & 63] + ": " + hipGetErrorString(e_); \
            return AFQ_EHIP;                                                    \
        }                                                                       \
    } while (0)
#define AFQ_GEMM(h, call) AFQ_GEMM_AS(h, __func__, call)
// the same with the breadcrumb / launch-trace name chosen by the caller (functions that launch several different GEMMs)
#define AFQ_GEMM_AS(h, name, call)                                              \
    do {                                                                        \
        afq_note_launch((h), name);                                             \
        hipError_t e_ = (call);                                                 \
        if (e_ == hipSuccess) e_ = afq_post_launch(h);                          \
        if (e_ != hipSuccess) {                                                 \
            (h)->err = std::string("GEMM of ") + __func__ + ": " + hipGetErrorString(e_); \
            return AFQ_EHIP;                                                    \
        }                                                                       \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is per (kernel, device): raise it once per device
#define AFQ_MAX_DEVICES 16
inline hipError_t afq_raise_lds(const void *kern, size_t lds, size_t (&set)[AFQ_MAX_DEVICES]) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= AFQ_MAX_DEVICES) d = -1;
    if (d >= 0 && lds <= set[d]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && d >= 0) set[d] = lds;
    return e;
}

#define AFQ_FAIL(h, code, msg) do { (h)->err = (msg); return (code); } while (0)

struct PhaseTimer {
    afq_handle *h; int slot;
    PhaseTimer(afq_handle *h_, int s) : h(h_), slot(s) {
        if (h->timers_on) hipEventRecord(h->ev0, h->stream);
    }
    ~PhaseTimer() {
        if (h->timers_on) {
            hipEventRecord(h->ev1, h->stream);
            hipEventSynchronize(h->ev1);
            float ms = 0; hipEventElapsedTime(&ms, h->ev0, h->ev1);
            h->t_ms[slot] += ms;
        }
    }
};

// Brackets ONE kernel launch with a start/stop event pair on the handle's stream when
// afq_kernel_trace is on; nothing is synchronised here (afq_kernel_trace_get reads the pairs).
struct KernelTrace {
    afq_handle *h; int kind, idx;
    KernelTrace(afq_handle *h_, int k) : h(h_), kind(k), idx(-1) {
        if (!(h->ktrace_mask >> k & 1) || h->ktrace_used[k] >= 4096) return;
        if (h->ktrace_seen[k]++ % h->ktrace_stride[k]) return;
        idx = h->ktrace_used[k];
        std::vector<hipEvent_t> &ev = h->ktrace_ev[k];
        while ((int)ev.size() < 2 * idx + 2) { hipEvent_t e; hipEventCreate(&e); ev.push_back(e); }
        hipEventRecord(ev[2 * idx], h->stream);
    }
    ~KernelTrace() {
        if (idx < 0) return;
        hipEventRecord(h->ktrace_ev[kind][2 * idx + 1], h->stream);
        h->ktrace_used[kind] = idx + 1;
    }
};

// ---- launchers implemented in the kernel translation units -----------------
// k_gemm.hip
int k_onebody(afq_handle *h, const cplx *rowscale = nullptr);   // phi <- [diag(rowscale_w)] BH1 phi (all live walkers)
int k_force_bias_generic(afq_handle *h);                    // ghalf -> vbias[2,nw,K]
bool k_msd_vbias_current(afq_handle *h);                    // every determinant's force-bias partials match ghalf_version
bool k_msd_gbar_wanted(afq_handle *h);                      // multi-determinant force bias through the averaged G
int k_force_bias_msd_gbar(afq_handle *h);                   // ghalf_all, detw -> averaged partials in vbias_all
bool k_fb_use_sum(afq_handle *h);                           // force bias runs once over Ghalf_a + Ghalf_b
int k_vhs_generic(afq_handle *h);                           // xs -> vhs
int k_apply_exponential(afq_handle *h, const cplx *vhs);    // phi <- sum_n vhs^n/n! phi
int k_full_G(afq_handle *h);                                // G = conj(psi) ghalf
// k_fused.hip
int k_prop_fused_supported(afq_handle *h);
int k_prop_fused(afq_handle *h);                           // phi <- B exp(V) B phi for live walkers, in place
// k_hirsch.hip
int k_hirsch_alive(afq_handle *h, int mode);
int k_hirsch_kinetic(afq_handle *h);
int k_hirsch_two_body(afq_handle *h);
int k_hirsch_eshift(afq_handle *h, double fac);
int k_hirsch_free(afq_handle *h, double eshift);           // free-projection step from the uniforms in hs_u
int k_rng_uniform(afq_handle *h, double *u, long n);
// k_fullg.hip
int k_energy_full_g(afq_handle *h, const cplx *G_dev, int ng, cplx *E_dev);   // estimators/generic.py:398-434
// k_bigdet.hip
int k_greens_big_supported(afq_handle *h);
struct WeightArgs;
int k_greens_big(afq_handle *h, cplx *ghalf, cplx *det, cplx *oinv = nullptr, const WeightArgs *wa = nullptr);   // ghalf may be null
                                // (overlap only); wa: the step's weight update rides on the determinant kernel
int k_reortho_big(afq_handle *h);                           // Cholesky-QR2; sets qr_fail for breakdowns
// k_small.hip
int k_alive(afq_handle *h);
int k_greens(afq_handle *h, cplx *det_out);                 // ghalf + det
int k_overlap(afq_handle *h, cplx *det_out);                // det(psi^H phi)
int k_inverse_overlap(afq_handle *h, cplx *oinv, cplx *det_out);   // O^-1 [nw,2,nmax,nmax] + det, live walkers
int k_fields(afq_handle *h);                                // vbias -> xbar(clipped), xs, cmf, cfb
int k_fields_explicit(afq_handle *h, const double *xi_d, const cplx *xbar_d, cplx *xs_d, cplx *cmf_d, cplx *cfb_d);
int k_xbar(afq_handle *h);
int k_bp_push(afq_handle *h);
int k_bp_fields(afq_handle *h, int i);
int k_bp_init(afq_handle *h, const cplx *phi0_dev);
int k_conj_copy(afq_handle *h, const cplx *src, cplx *dst, long n);
int k_conj_transpose(afq_handle *h, const cplx *A, cplx *At);
int k_bp_accumulate(afq_handle *h, int restore, int with_energy);
int k_bp_reset(afq_handle *h, bool first);
int k_bp_hirsch_step(afq_handle *h, int i);                 // B(x)^H of the i-th most recent discrete configuration
int k_xbar_fields(afq_handle *h, cplx *hubbard_factors = nullptr);   // + the Hubbard row-scaling factors (continuous fields)
int k_msd_combine(afq_handle *h, cplx *det_out, bool skip_small);          // detd -> detw, det_out = sum_d detw
int k_msd_energy_combine(afq_handle *h);                   // energy_all, detw -> energy                                  // vbias / G -> xbar (unclipped), system dispatch
int k_update_weight(afq_handle *h, cplx eshift);
int k_reortho(afq_handle *h, cplx *keep = nullptr, bool *keep_done = nullptr);
int k_cap_weights(afq_handle *h, double frac, double total_weight);
int k_comb(afq_handle *h, double r, double target, bool with_greens = false);
int k_clone_pairs(afq_handle *h, bool with_greens, bool reset_weights = false);
int k_closed_flags(afq_handle *h);          // closed_w[w] for every walker
int k_closed_copy_beta(afq_handle *h);      // phi[w][:, na:] = phi[w][:, :na] where closed_w[w]
int k_scale_by_inverse(afq_handle *h, cplx *x, const double *d);   // x[w] /= d[w]
int k_log_shift_reortho(afq_handle *h);                            // detR -> exp(log det R - detR_shift), log_detR += log
int k_log_ovlp_sums(afq_handle *h, double *out3);                  // sums of |ot|, |detR|, |log_detR| (device -> host)
int k_scale_weights(afq_handle *h, double scale);
int k_reset_weights(afq_handle *h, bool after_comb = false);
// the hand-over of a block's sums to the host (afq_estimates_get_begin): written into mapped host memory, zeroed, sequence
// number published -- by est_publish_kernel, or by the estimates_kernel launch that folds the last accumulators in (host_out
// null: no hand-over)
struct EstPublish {
    double *host_out = nullptr;
    unsigned long long *host_seq = nullptr;
    unsigned long long seq = 0;
    const double *scal = nullptr;
    const unsigned long long *closed_bad = nullptr;
    int nest = 0, zero = 0;
};
int k_estimates(afq_handle *h, int have_energy, bool fold_only = false, const EstPublish *pub = nullptr);
int k_rdm_accumulate(afq_handle *h);
int k_rng_normal(afq_handle *h);
int k_rng_normal_into(afq_handle *h, double *out_d, long n);
int k_philox_raw(afq_handle *h, const unsigned int *in_d, unsigned int *out_d, int n);
// k_ueg.hip (plane-wave step from Ghalf and per-walker HS coefficients: no G, no dense HS potential)
int k_ueg_fast_system(afq_handle *h, int M, int nq, const int64_t *Acp, const int64_t *Arow, const double *Aval,
                      const int64_t *Bcp, const int64_t *Brow, const double *Bval);
int k_ueg_fast_trial(afq_handle *h, const double *psi);
int k_ueg_fast_propagator(afq_handle *h, const double *BH1);
int k_ueg_fast_supported(afq_handle *h);
int k_ueg_fields(afq_handle *h);                            // force bias + fields + HS coefficients
int k_prop_ueg(afq_handle *h);                              // phi <- B exp(V) B phi
int k_ueg_step(afq_handle *h);                              // the two above in one launch when its LDS plan fits
void k_ueg_fast_free(afq_handle *h);
// k_comm.hip
int k_comm_size(afq_handle *h);                             // ranks of the library-owned communicator (1 without)
void k_comm_destroy(afq_handle *h);
int k_comm_popcontrol(afq_handle *h, double r, double target, bool with_greens);
// k_energy.hip
int k_energy_generic(afq_handle *h);
int k_prepare_energy_operands(afq_handle *h, const double *rchol_host);
int k_exchange_uses_quadratic(afq_handle *h);            // which algorithm k_energy_generic takes for the current trial
void k_free_atil(void *(&atil)[2]);
// k_models.hip (Hubbard / UEG)
int k_vhs_hubbard(afq_handle *h);
int k_apply_exponential_diag(afq_handle *h, const cplx *vhs_diag);
int k_exp_diag_factors(afq_handle *h, const cplx *vhs_diag, cplx *out);   // out[w, v, p] = sum_{n <= order} d^n / n!
int k_energy_hubbard(afq_handle *h);
int k_vbias_ueg(afq_handle *h);
int k_vhs_ueg(afq_handle *h);
int k_energy_ueg(afq_handle *h);
