/*
 * afqmc_hip.h -- C ABI of libafqmc_hip.so: MI355X (gfx950) phaseless-AFQMC
 * walker propagation, Green's functions and local energies, walker-batched.
 *
 * The reference (pauxy-qmc/pauxy) has no FFI on this path: the hot path sits
 * behind the Python plug-in objects Propagator / Walkers / Estimators that
 * pauxy/qmc/afqmc.py drives.  Each entry point below therefore cites the
 * reference *Python* function it replaces (paths relative to the reference's
 * pauxy/ package); pauxy_amd/ holds the Python classes that bind these entry
 * points with ctypes behind the reference's own class/method names
 * (INTEGRATION.md shows the binding).
 *
 * Conventions
 *  - plain C: opaque handle, pointers, ints, doubles.  Every function returns
 *    0 (AFQ_OK) or a negative AFQ_E* code; afq_last_error() gives the text.
 *  - "c128" = complex128 as numpy stores it: interleaved (re, im) doubles,
 *    C order.  Host arrays are only read unless documented "out".
 *  - one handle per GPU, owned by one host thread; calls on one handle are
 *    serialised by the caller.  All work is queued on the handle's HIP stream;
 *    functions that return data to the host synchronise that stream first.
 *  - the handle owns every device allocation.
 */
#ifndef AFQMC_HIP_H
#define AFQMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct afq_handle afq_handle;

#define AFQ_OK            0
#define AFQ_EINVAL       (-1)   /* bad argument                                  */
#define AFQ_ESTATE       (-2)   /* call order: system/trial/propagator/walkers   */
#define AFQ_EHIP         (-3)   /* HIP runtime error (text in afq_last_error)    */
#define AFQ_ENOMEM       (-4)
#define AFQ_EUNSUPPORTED (-5)
#define AFQ_EWEIGHT      (-6)   /* total weight < 1e-8 (walkers/handler.py:236)  */
#define AFQ_EOVERFLOW    (-7)   /* more walkers moved between two ranks in one population control than the
                                   exchange slots of the communicator hold (afq_comm_set_capacity)   */
#define AFQ_ECOMM        (-8)   /* communicator: a peer never signalled, probe mismatch, bootstrap failure */

/* system kinds */
#define AFQ_SYS_GENERIC 1
#define AFQ_SYS_HUBBARD 2
#define AFQ_SYS_UEG     3

/* propagator flags (propagation/continuous.py:18-37) */
#define AFQ_PROP_HYBRID          1   /* hybrid weight update (default)            */
#define AFQ_PROP_FORCE_BIAS      2
#define AFQ_PROP_FREE_PROJECTION 4
#define AFQ_PROP_HUBBARD_SPIN    8   /* charge_decomposition: false               */

/* walker fields for afq_walkers_set / afq_walkers_get.
 * per-walker shapes: PHI c128[M, na+nb]; WEIGHT, UNSCALED_WEIGHT, DETR f64;
 * OT, HYBRID_ENERGY, PHASE, ELOC c128; GHALF c128[na+nb, M] (alpha rows first);
 * G c128[2, M, M]; XBAR, XSHIFTED c128[K]; ENERGY c128[3]; LOG_DETR f64         */
enum afq_field {
    AFQ_F_PHI = 0,
    AFQ_F_WEIGHT = 1,
    AFQ_F_UNSCALED_WEIGHT = 2,
    AFQ_F_OT = 3,
    AFQ_F_HYBRID_ENERGY = 4,
    AFQ_F_PHASE = 5,
    AFQ_F_DETR = 6,
    AFQ_F_ELOC = 7,
    AFQ_F_GHALF = 8,
    AFQ_F_G = 9,
    AFQ_F_XBAR = 10,
    AFQ_F_XSHIFTED = 11,
    AFQ_F_ENERGY = 12,
    AFQ_F_LOG_DETR = 13,
    AFQ_F_COUNT_
};

/* index of each accumulator in the estimates vector (estimators/mixed.py:460-469) */
enum afq_estimate {
    AFQ_EST_UWEIGHT = 0, AFQ_EST_WEIGHT = 1, AFQ_EST_ENUMER = 2, AFQ_EST_EDENOM = 3,
    AFQ_EST_EPROJ = 4, AFQ_EST_E1B = 5, AFQ_EST_E2B = 6, AFQ_EST_EHYB = 7,
    AFQ_EST_OVLP = 8, AFQ_EST_TIME = 9, AFQ_EST_COUNT_ = 10
};

/* ---- lifetime ----------------------------------------------------------- */
int afq_version(void);
int afq_create(int device_id, afq_handle **out);
int afq_destroy(afq_handle *h);
const char *afq_last_error(afq_handle *h);
int afq_sync(afq_handle *h);

/* ---- read-only inputs of the path --------------------------------------- */
/* systems/generic.py:74-166 arrays.  hs_pot f64[M*M, K] (= chol_vecs);
 * rchol c128[(na+nb)*M, K], row i*M+p, alpha block first
 * (trial_wavefunction/multi_slater.py:402-409); H1 c128[2,M,M].              */
int afq_set_system_generic(afq_handle *h, int M, int K, int na, int nb,
                           const double *hs_pot, const double *rchol,
                           const double *H1, double ecore);
/* systems/hubbard.py:46-104.  T c128[2,M,M]; fields K = M.                    */
int afq_set_system_hubbard(afq_handle *h, int M, int na, int nb, double U,
                           const double *T);
/* systems/ueg.py:43-191,336-428.  iA, iB: CSC [M*M, nq] (colptr int64[nq+1],
 * rowidx int64[nnz], val c128[nnz]); the four ragged index lists of
 * systems/ueg.py:139-176 as offsets int64[nq+1] + flat int64 indices;
 * vqvec f64[nq]; H1diag f64[2,M]; fields K = 2*nq.                            */
int afq_set_system_ueg(afq_handle *h, int M, int nq, int na, int nb,
                       const int64_t *iA_colptr, const int64_t *iA_row, const double *iA_val,
                       const int64_t *iB_colptr, const int64_t *iB_row, const double *iB_val,
                       const int64_t *kpq_off, const int64_t *kpq_i, const int64_t *kpq_kpq,
                       const int64_t *pmq_off, const int64_t *pmq_i, const int64_t *pmq_pmq,
                       const double *vqvec, double vol, const double *H1diag, double ecore);
/* trial determinant psi c128[M, na+nb] (trial.psi after walkers/handler.py:61) */
int afq_set_trial(afq_handle *h, const double *psi);
/* ---- discrete Hirsch Hubbard-Stratonovich propagator -----------
 * propagation/hubbard.py:12-343 (Hirsch, single-site updates, constrained path) for a Hubbard system and a
 * single-determinant trial with N <= 128 electrons per spin (the inverse overlaps live in LDS up to N = 68 and are
 * updated in place in global memory beyond).  bt2 c128[2, M, M] = expm(-dt/2 T) (:36-37);
 * charge_decomposition selects the charge (complex gamma) or spin HS (:66-82).
 *   afq_propagate_hirsch   one step for every walker with |weight| > 1e-8, site uniforms from the device
 *                          Philox stream: kinetic + importance sampling (:148-172), M single-site
 *                          updates (:174-225), kinetic again, weight *= exp(dt eshift) (:285-312)
 *   afq_hirsch_kinetic / afq_hirsch_two_body / afq_hirsch_finish   the same step in three calls so that a
 *                          host driver can hand in numpy's uniforms: u f64[nw, M] (row w is read only if
 *                          walker w survived the first kinetic step, i.e. weight != 0 after it);
 *                          fields_out int32[nw, M] (chosen field 0/1, -1 not visited), used_out int32[nw]
 *                          (uniforms consumed; < M only when both field probabilities vanish) -- both
 *                          may be NULL.
 *   afq_hirsch_free_projection / afq_propagate_hirsch_free   propagate_walker_free (:303-343): no importance
 *                          sampling -- kinetic half step, every site takes field 0 (uniform < 0.5) or 1 and scales its row
 *                          of phi, kinetic half step, weight *= exp(dt eshift) |wfac|, phase *= exp(i arg wfac),
 *                          ot = <psi_T|phi>.  The first call switches the mode (the estimators then accumulate
 *                          weight * ot * phase, mixed.py:151-175, and the re-orthogonalisation folds det R into weight and
 *                          phase); u f64[nw, M] numpy's uniforms or NULL for the device stream, fields_out may be NULL.
 *   afq_hirsch_single_site    on = 0 replaces the M single-site updates of the three calls above (and of
 *                          afq_propagate_hirsch) by two_body_direct (:222-275, `single_site_update: False`): all M fields
 *                          drawn from the dynamic force bias of the current Green's function, phi scaled once, one overlap;
 *                          M uniforms per walker with a non-zero weight, no field history (back propagation: AFQ_EUNSUPPORTED). */
int afq_set_propagator_hirsch(afq_handle *h, const double *bt2, double dt, int charge_decomposition);
int afq_hirsch_free_projection(afq_handle *h, int on);
int afq_hirsch_single_site(afq_handle *h, int on);
int afq_propagate_hirsch_free(afq_handle *h, const double *u, int32_t *fields_out, double eshift);
int afq_propagate_hirsch(afq_handle *h, double eshift);
int afq_hirsch_kinetic(afq_handle *h);
int afq_hirsch_two_body(afq_handle *h, const double *u, int32_t *fields_out, int32_t *used_out);
int afq_hirsch_finish(afq_handle *h, double eshift);

/* Local energy of a generic Cholesky Hamiltonian from FULL Green's functions
 * (estimators/generic.py:398-434, local_energy_generic_cholesky; what estimators/mixed.py:383-437
 * dispatches to when no half-rotated Ghalf is given): G c128[n, 2, M, M] -> E c128[n, 3] = (E, E1b, E2b).
 * O(M^3 K) per Green's function; not on the per-step path.                                       */
int afq_local_energy_full_g(afq_handle *h, const double *G, int n, double *E_out);

/* ---- back-propagated estimator --------------------------------
 * estimators/back_propagation.py:63-226, walkers/stack.py:5-127 (FieldConfig),
 * propagation/generic.py:181-211,253-290.  After afq_bp_configure every
 * afq_propagate records the shifted fields x - xbar of each walker together with
 * the phase / cosine factors of the weight update (up to nbp steps; the history,
 * phi_old and the factors travel with the walker through the comb, afq_walkers_copy
 * and afq_walker_pack / unpack).  Call after afq_walkers_alloc and afq_set_propagator;
 * phi_old starts as the current phi.
 * Hubbard systems: with the discrete Hirsch propagator the kernel of afq_hirsch_two_body / afq_propagate_hirsch
 * records the chosen field of every site as it goes (FieldConfig.push, walkers/stack.py:35-49, called at
 * propagation/hubbard.py:215-216; no weight factors on this path), and afq_bp_update applies
 * B(x)^H = (BT2 diag(auxf[x, spin]) BT2)^H of propagation/hubbard.py:568-600,634-672; the continuous Hubbard
 * propagator is refused (the reference would read its fields as 0 / 1 indices, estimators/back_propagation.py:125). */
int afq_bp_configure(afq_handle *h, int nbp);
/* FieldConfig.step of every walker: int32[nw]                                   */
int afq_bp_steps(afq_handle *h, int32_t *steps_out);
/* BackPropagation.update_uhf for the whole population: back-propagates phi_bp0
 * (trial.psi or trial.init, c128[M, na+nb]) through every walker's recorded fields
 * (most recent first, re-orthogonalised every nstblz steps), forms
 * G_bp[w] = gab(phi_bp, phi_old)^T and returns est_out c128[4 + 2 M M] =
 * [0, 0, 0, sum_w wt_w, sum_w wt_w G_bp[w]] with wt = weight (restore_weights 0),
 * weight * prod(I/|I|) (1, "partial") or weight * prod(I/|I|) / prod(cos) (2, "full");
 * then, when `reset` is set (the last of the nsplit path lengths, back_propagation.py:68-69,219-222), resets the
 * histories and copies phi -> phi_old.  Generic systems (propagation/generic.py:253-290) and the UEG
 * (propagation/planewave.py:114-178; B(x)^H = B(-conj x) for both).  With eval_energy the entries 0-2 are
 * sum_w wt_w (E, E1b, E2b)[G_bp[w]] from the full-G Cholesky energy (generic systems).          */
int afq_bp_update(afq_handle *h, const double *phi_bp0, int nstblz, int restore_weights, int eval_energy,
                  int reset, double *est_out);

/* Multi-determinant (NOMSD / PHMSD) trial |psi_T> = sum_d c_d |D_d> for a generic system, replacing
 * the single-determinant operands of afq_set_system_generic / afq_set_trial.  Call after the
 * system and before afq_walkers_alloc.
 *   psi    [ndet, M, na+nb] c128     pauxy/trial_wavefunction/multi_slater.py:36-38 (trial.psi)
 *   coeffs [ndet] c128               trial.coeffs (enter as conj(c_d), walkers/multi_det.py:226)
 *   rchol  [ndet, (na+nb) M, K] c128 per-determinant half rotation, multi_slater.py:370-409
 * Green's function / overlap, force bias and local energy then follow walkers/multi_det.py:194-257,
 * 135-162, propagation/generic.py:154-157 and estimators/mixed.py:439-448 (the per-determinant
 * energy is evaluated in its half-rotated form, algebraically equal to the reference's full-G form). */
int afq_set_trial_multi(afq_handle *h, int ndet, const double *psi, const double *coeffs, const double *rchol);
/* weights conj(c_d) <D_d|phi_w> of the last Green's function / overlap evaluation: [nw, ndet] c128
 * (MultiDetWalker.weights, walkers/multi_det.py:226)                                                  */
int afq_walkers_det_weights(afq_handle *h, double *weights_out);

/* propagation/continuous.py:13-80 + <system>.construct_one_body_propagator:
 * BH1 c128[2,M,M], mf_shift c128[K], dt, expansion_order, AFQ_PROP_* flags.    */
int afq_set_propagator(afq_handle *h, const double *BH1, const double *mf_shift,
                       double dt, int exp_order, int flags);

/* ---- walker population (walkers/handler.py:36-164, walkers/walker.py:24-61) */
int afq_walkers_alloc(afq_handle *h, int nw);
int afq_walkers_set(afq_handle *h, int field, const void *host, int first, int count);
int afq_walkers_get(afq_handle *h, int field, void *host, int first, int count);
/* device pointer of a field (row `first`), for zero-copy plumbing */
int afq_walkers_device_ptr(afq_handle *h, int field, void **dev_ptr, int64_t *bytes_per_walker);

/* ---- the hot path -------------------------------------------------------- */
/* walkers/single_det.py:295-321 for every walker: Ghalf (+ full G when
 * want_G or the system needs it) and det(phi^T psi*) -> ovlp_out c128[nw]
 * (may be NULL).                                                            */
int afq_greens(afq_handle *h, int want_G, double *ovlp_out);
/* walkers/single_det.py:170-199 for every walker -> ovlp_out c128[nw]         */
int afq_calc_overlap(afq_handle *h, double *ovlp_out);
/* O^-1 of every walker and spin, O = phi_s^T conj(psi_s) (so O^-1 = inv_O of estimators/greens_function.py:82-115
 * and the transpose of SingleDetWalker.inv_ovlp, walkers/single_det.py:95-115): c128[nw, 2, nmax, nmax] row-major with
 * leading dimension nmax = max(na, nb), zero padded; ovlp_out c128[nw] may be NULL.  N <= 128.                         */
int afq_inverse_overlap(afq_handle *h, double *oinv_out, double *ovlp_out);
/* propagation/continuous.py:232-262 (phaseless) or :175-200 (free projection)
 * for every live walker (|weight| > 1e-8, qmc/afqmc.py:232).
 * xi: host f64[nw, K] normal fields (row iw used only if walker iw is live),
 * or NULL to draw them on the device (afq_rng_seed).                         */
int afq_propagate(afq_handle *h, const double *xi, double eshift_re, double eshift_im);
/* The same step in two calls.  Only the weight update at the very end of a step reads the energy shift
 * (propagation/continuous.py:194-200, :206-230): afq_propagate_begin enqueues everything up to and including the
 * propagation of the Slater matrices, afq_propagate_finish the overlap / Green's function of the propagated walkers and
 * the weight update.  A driver that derives the shift of the next block from the estimators of the block just ended
 * (qmc/afqmc.py:247-250 after estimators/mixed.py:261-273) can therefore enqueue the first part of the next step BEFORE
 * it waits for those estimators (afq_estimates_get_begin / _end below) and keep the device busy across the block
 * boundary.  afq_propagate(h, xi, e) == afq_propagate_begin(h, xi) + afq_propagate_finish(h, e); between the two calls
 * only afq_estimates_get_end, afq_set_weight_cap and afq_last_error may be called (AFQ_ESTATE otherwise).           */
int afq_propagate_begin(afq_handle *h, const double *xi);
int afq_propagate_finish(afq_handle *h, double eshift_re, double eshift_im);
/* walkers/handler.py:166-181 -> walkers/single_det.py:215-255; detR f64[nw] out (may be NULL) */
int afq_reortho(afq_handle *h, double *detR_out);
/* walkers: {use_log_shift: true} (walkers/handler.py:45,228,456-475, walkers/single_det.py:159,192,250-253,320).
 * Every walker carries the same log_shift / detR_shift (walker.py:49-52, handler.py:471-474), so they are handle
 * state: walker.ot = overlap * exp(-log_shift) on the continuous path (exp(+log_shift) behind calc_otrial on the
 * discrete path, :159), detR = exp(log det R - detR_shift), and AFQ_F_LOG_DETR accumulates log(detR) per walker.
 * on == 0 switches the option off.  afq_calc_overlap / afq_greens keep returning the unshifted determinant.
 * afq_log_ovlp_sums: f64[3] = sums over this rank's walkers of |ot|, |detR|, |log_detR| (handler.py:457-462);
 * the caller reduces them over ranks and feeds the running averages back through afq_set_log_shift.          */
int afq_set_log_shift(afq_handle *h, int on, double log_shift, double detR_shift);
int afq_log_ovlp_sums(afq_handle *h, double *sums3);
/* estimators/mixed.py:383-437 dispatch for every walker, from the current
 * Ghalf/G (call afq_greens first): E c128[nw,3] = (E, E1, E2); may be NULL.   */
int afq_local_energy(afq_handle *h, double *E_out);

/* The exchange part of estimators/generic.py:198-216 has two device algorithms with the same result:
 *   1  T-intermediate: T[x,i,j] = sum_p rchol[(i,p),x] Ghalf[j,p] on MFMA tiles, traced in registers
 *      (4 K M N^2 flops per spin and walker)
 *   2  quadratic form: exx = g^T Atil g with Atil[(j,p),(i,q)] = sum_x rchol[(i,p),x] rchol[(j,q),x] built once per
 *      trial ((N M)^2 entries per spin in HBM) -- one [nw x NM] x [NM x NM] GEMM per evaluation, K / M times
 *      fewer flops
 * mode 0 (default) picks 2 when K >= M and the operands of all determinants fit 72 GB, else 1.            */
int afq_set_exchange_algorithm(afq_handle *h, int mode);
int afq_exchange_algorithm(afq_handle *h, int *mode_out);

/* Force bias of a multi-determinant trial (propagation/generic.py:154-157, walkers/multi_det.py:283-290) has two
 * device algorithms with the same result:
 *   1  one half-rotated contraction per determinant, rchol_d^T vec(Ghalf_d), averaged with the weights
 *      conj(c_d) <D_d|phi> afterwards (ndet, or 2 ndet for complex vectors, times K (Na + Nb) M products)
 *   2  the reference's own formulation: the weights go into ONE determinant-averaged Green's function
 *      Gbar = sum_d w_d conj(psi_d) Ghalf_d / sum_d w_d, contracted once with the symmetric-packed hs_pot
 *      (K M (M + 1) / 2 products + the build of Gbar); needs symmetric L_n
 * mode 0 (default) picks 2 when it is the cheaper one for more than 32 walkers.  afq_msd_force_bias: what the next
 * force bias will use (0 for a single-determinant trial).                                                          */
int afq_set_msd_force_bias(afq_handle *h, int mode);
int afq_msd_force_bias(afq_handle *h, int *mode_out);

/* unit-test hooks (reference: <system propagator>.construct_force_bias /
 * construct_VHS, Continuous.apply_exponential, operations.kinetic_real)       */
int afq_force_bias(afq_handle *h, double *xbar_out);                 /* c128[nw,K] */
int afq_shift_fields(afq_handle *h, const double *xi, const double *xbar,
                     double *xs_out, double *cmf_out, double *cfb_out); /* continuous.py:140-158 */
int afq_vhs(afq_handle *h, const double *xs, double *vhs_out);       /* c128[nw,nv,M,M] */
int afq_vhs_count(afq_handle *h, int *nv);                           /* 1, or 2 for spin HS */
int afq_apply_exponential(afq_handle *h, const double *vhs);         /* phi <- Taylor(vhs) phi */
int afq_kinetic(afq_handle *h);                                      /* phi <- BH1 phi */

/* ---- driver glue that must stay on the device between steps --------------- */
/* qmc/afqmc.py:235-236: weight > frac*total_weight -> frac*total_weight.  A negative
 * total_weight means "the total weight the last afq_popcontrol_comb measured" (kept on the
 * device; the population size before the first comb), which needs no host round trip.     */
int afq_cap_weights(afq_handle *h, double frac, double total_weight);
/* The same cap applied by afq_propagate itself, at the end of its weight-update kernel (one launch
 * less per step).  frac <= 0 switches it off (the default); total_weight as in afq_cap_weights.   */
int afq_set_weight_cap(afq_handle *h, double frac, double total_weight);
/* walkers/handler.py:225-338 for a single rank: rescale, comb with the uniform
 * r, clone/kill, weights reset to 1.  parent_ix int32[nw] out (may be NULL);
 * total_weight_out f64 (may be NULL).  With BOTH outputs NULL the call only enqueues
 * work (no host synchronisation, no AFQ_EWEIGHT check).                         */
int afq_popcontrol_comb(afq_handle *h, double r, double target_weight,
                        int32_t *parent_ix, double *total_weight_out);

/* ---- library-owned communicator ----------------------------------------------------
 * walkers/handler.py:225-338 across ranks -- the Allgather of the weights (:232), rank 0's comb uniform
 * (:276,291) and the point-to-point walker copies (:303-331) -- and the estimator reduction of
 * estimators/mixed.py:261,273, on the device over RCCL (xGMI on one node), one process per GPU.
 * After afq_comm_init, afq_popcontrol_comb is a collective: every rank calls it with the same
 * target_weight (= total walkers) and rank 0's r; the weights are all-gathered on the device, every rank
 * decides the identical global comb, cloned walkers that change rank are written by the sending GPU straight
 * into the receiving GPU's mapped window (live slots only), nothing is read back by the host unless outputs are requested
 * (parent_ix is then the GLOBAL int32[nranks * nw] comb, total_weight_out the global weight).  The cap of
 * afq_cap_weights / afq_set_weight_cap with total_weight < 0 uses the global weight of the last comb.
 *   afq_comm_unique_id     128-byte ncclUniqueId, made by rank 0 and handed to every rank by the caller
 *   afq_comm_available     1 if librccl can be loaded in this process (no collective: lets the ranks AGREE on calling
 *                          afq_comm_init before any of them blocks inside ncclCommInitRank)
 *   afq_comm_init          ncclCommInitRank on the handle's GPU (librccl is loaded here, not before)
 *   afq_comm_init_ipc      the same communicator without RCCL: every rank maps every peer's window
 *                          (hipIpcGetMemHandle / hipIpcOpenMemHandle, xGMI peer access on one node) and all three
 *                          exchanges -- weights all-gather, walker slots, estimator reduction -- are kernels writing
 *                          into the peers' windows, ordered by system-scope flags.  `allgather(send, recv, bytes, user)`
 *                          is the caller's all-gather of `bytes` per rank over its own communicator (MPI /
 *                          torch.distributed; returns 0): the bootstrap for the window handles, called on the host at
 *                          afq_comm_init_ipc time and whenever the windows are re-made (first population control,
 *                          capacity change) -- at the same call on every rank
 *   afq_comm_set_transport 1 (default): walkers that change rank are written by the pack kernel straight into the
 *                          destination rank's mapped window, LIVE SLOTS ONLY, no host knowledge of the counts, nothing
 *                          posted by the host; 0 (RCCL communicator only; also the automatic fall-back when a rank cannot
 *                          export / map windows): fixed-capacity slots to and from every peer in one ncclSend / ncclRecv group
 *   afq_comm_set_capacity  walkers one rank can send to one peer per event (default nw up to 512 walkers per rank:
 *                          what a rank owns, cannot overflow; max(512, nw / 4) above, so that the windows stay a few
 *                          hundred MB); exceeding it raises AFQ_EOVERFLOW at the next afq_estimates_get / fetching comb.
 *                          Only the ncclSend / ncclRecv transport pays link time for unused capacity
 *   afq_comm_set_timeout   how long (seconds, > 0) a kernel waits for a peer before it gives up; default 300 s, or
 *                          AFQ_COMM_TIMEOUT_S from the environment when a communicator is created.  The budget covers
 *                          whatever may delay one rank between two collectives (rank 0 writing output or restart files,
 *                          a file-system stall, first-use code loading); it is per process and GPU, not per handle
 *   afq_comm_probe         collective known-answer round through the all-gather, one full slot to and from every peer on
 *                          the configured transport, and the all-reduce; AFQ_ECOMM on any mismatch
 *                          (mismatch_out int64[3], may be NULL: collective values, slot elements, flags that never came)
 *   afq_comm_stats         int64[AFQ_COMM_NSTATS]: largest per-peer transfer seen, events, capacity, overflow flag, rank,
 *                          size, walkers sent, bytes sent, transport (1 windows / 0 send-recv), communication error flag,
 *                          communicator kind (0 in-process, 1 RCCL, 2 IPC), window memory kind (1 uncached,
 *                          2 fine-grained, 3 plain)
 *   afq_estimates_allreduce  sum over ranks of buf c128[nest] (in/out, host), or with buf == NULL of the
 *                          device accumulators of afq_estimates_update in place (no host round trip; a
 *                          following afq_estimates_get returns the global sums on every rank)
 * A kernel that waits for a peer gives up after the wait budget and raises a sticky error, reported as AFQ_ECOMM by
 * the next synchronising call (afq_estimates_get / _end, a fetching afq_popcontrol_comb) -- never a hung device.  The
 * sticky flags and statistics belong to the communicator: creating or destroying one clears them.
 * In-process variant: afq_comm_init_local makes handles[0..n) (one host thread driving several GPUs, or
 * several handles on one GPU) the ranks 0..n-1 of one communicator; the collectives are then the group calls
 * afq_popcontrol_comb_local / afq_estimates_allreduce_local (the kernels and windows of the IPC communicator, the
 * windows being plain device pointers of the one process).                                               */
#define AFQ_COMM_ID_BYTES 128
#define AFQ_COMM_NSTATS 12
typedef int (*afq_allgather_fn)(const void *send, void *recv, int bytes, void *user);
int afq_comm_unique_id(void *id_out);
int afq_comm_available(void);
int afq_comm_init(afq_handle *h, const void *unique_id, int rank, int nranks);
int afq_comm_init_ipc(afq_handle *h, int rank, int nranks, afq_allgather_fn allgather, void *user);
int afq_comm_destroy(afq_handle *h);
int afq_comm_set_transport(afq_handle *h, int window);
int afq_comm_set_capacity(afq_handle *h, int max_walkers_per_peer);
int afq_comm_set_timeout(afq_handle *h, double seconds);
int afq_comm_probe(afq_handle *h, int64_t *mismatch_out);
int afq_comm_stats(afq_handle *h, int64_t *out);
int afq_comm_parent_ix(afq_handle *h, int32_t *parent_ix_global);
int afq_estimates_allreduce(afq_handle *h, double *buf, int nest);
int afq_comm_init_local(afq_handle **handles, int n);
int afq_popcontrol_comb_local(afq_handle **handles, int n, double r, double target_weight,
                              int32_t *parent_ix_global, double *total_weight_out);
int afq_estimates_allreduce_local(afq_handle **handles, int n);
/* multi-rank building blocks: scale weights by 1/scale saving unscaled_weight
 * (handler.py:244-246); copy walker src -> dst inside this GPU; pack / unpack
 * the minimal walker state (phi + scalars) to a device buffer for transport;
 * reset all weights to 1 (handler.py:337-338).                                 */
int afq_walkers_scale_weights(afq_handle *h, double scale);
int afq_walkers_copy(afq_handle *h, int src, int dst);
int afq_walker_pack_bytes(afq_handle *h, int64_t *bytes);
int afq_walker_pack(afq_handle *h, int iw, void *dev_buf);
int afq_walker_unpack(afq_handle *h, int iw, const void *dev_buf);
int afq_walkers_reset_weights(afq_handle *h);
/* estimators/mixed.py:180-225: accumulate the 10 mixed estimators over all
 * walkers on the device; eval_energy != 0 runs afq_greens + afq_local_energy. */
int afq_estimates_update(afq_handle *h, int eval_energy);
/* Mixed estimator with one_rdm: True (estimators/mixed.py:226-233,279-283): after afq_estimates_rdm(h, 1) every
 * afq_estimates_update also adds sum_w weight_w Re(G_w) to a device accumulator f64[2, M, M], where G_w is
 * walker.G as the reference leaves it: the Green's function evaluated before the step's propagation
 * (propagation/continuous.py:245), refreshed on energy steps, cloned with the walker by the comb.
 * afq_estimates_rdm_get returns (and optionally zeroes) the accumulator.  Single determinant.  With a communicator
 * walker.G travels with a walker cloned to another rank (exchange slot, afq_walker_pack) and afq_estimates_allreduce
 * reduces the accumulator along with the ten estimators (the reference keeps them in one vector, mixed.py:261).  */
int afq_estimates_rdm(afq_handle *h, int on);
int afq_estimates_rdm_get(afq_handle *h, double *rdm_out /* f64[2, M, M] */, int zero);
/* Synchronises the stream; also the place where a population that collapsed in an asynchronous comb is
 * reported (AFQ_EWEIGHT, walkers/handler.py:236-241) and an exchange overflow (AFQ_EOVERFLOW).           */
int afq_estimates_get(afq_handle *h, double *est_out /* c128[10] */, int zero);
/* The same without blocking at enqueue time: _begin enqueues the copy of the sums (and their zeroing) behind the work
 * already in the stream, _end waits for that copy only -- work enqueued after _begin keeps running.  One fetch in flight
 * at a time (AFQ_ESTATE).                                                                                            */
/* The next afq_propagate (or afq_propagate_finish) also takes the estimator terms of its step along, as an
 * afq_estimates_update(h, 0) called right behind it would have added them: the weight update adds every walker's terms
 * to per-walker accumulators, and the next afq_estimates_update, or the next fetch / all-reduce, folds those into the
 * sums.  For a step that neither combs nor evaluates the energy this saves the launch of the summation kernel.
 * The call also tells the library that nothing but the next step's force bias will read the Green's function this step
 * leaves behind: where that force bias contracts Ghalf_a + Ghalf_b (generic system, both spins sharing the half-rotated
 * Cholesky block, hybrid weights) the per-spin Ghalf is not stored; afq_local_energy / afq_walkers_get(AFQ_F_GHALF)
 * called afterwards evaluate it first (same numbers, one more kernel).
 * Continuous propagator, not with afq_estimates_rdm on.                                                            */
int afq_estimates_fuse_next(afq_handle *h);
int afq_estimates_get_begin(afq_handle *h, int zero);
/* afq_estimates_update(h, eval_energy) followed by afq_estimates_get_begin(h, zero) as ONE call: the summation launch of the
 * update writes the block's sums into the mapped host buffer itself (estimators/mixed.py:211-225 + :261-273 of the step
 * that ends a block) -- one launch less at every block boundary; afq_estimates_get_end collects them as after _get_begin. */
int afq_estimates_update_publish(afq_handle *h, int eval_energy, int zero);
int afq_estimates_get_end(afq_handle *h, double *est_out /* c128[10] */);

/* ---- misc ------------------------------------------------------------------ */
/* Device stream of auxiliary fields (used by afq_propagate with xi == NULL; replaces numpy.random.normal of
 * propagation/continuous.py:133 when parity with the host stream is not required): Philox4x32-10 (Salmon et
 * al., SC'11), counter = (element pair index, launch counter ^ stream hash), key = seed, Box-Muller on two
 * 53-bit uniforms.  `stream` must differ between ranks (qmc/utils.py:14 seeds seed + rank).
 *   afq_rng_normal      the next n normals of the stream (what the next afq_propagate would have used)
 *   afq_rng_philox4x32  the bare block function for known-answer tests: ctr_key uint32[n][6] =
 *                       (counter[4], key[2]) -> out uint32[n][4]                                          */
int afq_rng_seed(afq_handle *h, uint64_t seed, uint64_t stream);
int afq_rng_normal(afq_handle *h, double *out, int64_t n);
int afq_rng_philox4x32(afq_handle *h, const uint32_t *ctr_key, uint32_t *out, int n);
/* Diagnostics.  Every kernel launch leaves its name in a host-side ring of the handle; afq_last_launch formats
 * "api=<entry point> queued=<n> [retired=<m> first-unretired=<kernel>] last: <names>" into buf and may be
 * called from a watchdog thread while the owning thread is blocked in a synchronising call (host memory only).
 * afq_debug(h, sync, markers): sync != 0 synchronises and checks after every launch (a failing kernel is named
 * in afq_last_error); markers != 0 queues a one-thread marker behind every launch so that `retired` counts the
 * launches that completed.  Environment: AFQ_DEBUG_SYNC=1, AFQ_DEBUG_MARKERS=1 set them at afq_create.     */
int afq_debug(afq_handle *h, int sync_every_launch, int markers);
int afq_last_launch(afq_handle *h, char *buf, int len, uint64_t *queued, uint64_t *retired);
/* counters: [0]=nfb_trig (continuous.py:150), [1]=nhe_trig (:210,213), [2]=overlap matrices the blocked
 * Gauss-Jordan flagged as poorly conditioned block-wise and handed to the step-by-step kernel (diagnostic),
 * [3]=walker steps the fused propagator took through its closed-shell deal (spin blocks bitwise equal)       */
int afq_counters(afq_handle *h, int64_t *out, int reset);
/* the same, first n counters (n <= 8): [4]=walker energy evaluations whose exchange energy was evaluated for one spin and
 * counted twice (closed-shell population, decided on the device: energy_finish_kernel), [5]=walker Green's functions
 * computed for one spin (closed-shell walker, greens_small_kernel), [6]=per-determinant overlaps of a multi-determinant
 * trial that came out as NaN (0 x inf behind a zero pivot of a singular overlap matrix) and were taken as zero overlaps,
 * [7]=walker steps that went through the large-system GEMM chain (M > 128) as closed-shell walkers: alpha columns only in the
 * one-body and Taylor products, copied over the beta block behind the chain (closed_flags_kernel)                      */
int afq_counters_ext(afq_handle *h, int64_t *out, int n, int reset);
/* accumulated device ms per phase: [0] greens [1] one-body [2] force bias+fields
 * [3] vhs [4] exponential [5] overlap+weight [6] reortho [7] energy            */
int afq_timers(afq_handle *h, double *out_ms, int reset);
int afq_enable_timers(afq_handle *h, int on);
/* HIP stream of the handle (hipStream_t as void*) */
int afq_stream(afq_handle *h, void **stream);
/* Per-launch durations of the hot kernels, HIP events recorded on the handle's
 * stream around the launch (no host synchronisation until _get): bench.py's
 * live roofline measurement over its timed region.  afq_kernel_trace(h, 1)
 * clears and starts recording every kind (up to 4096 launches per kind),
 * (h, 2 << kind | ...) only the selected kinds (an event pair costs a few
 * microseconds of pipeline bubble per launch), (h, 0) stops.
 * afq_kernel_trace_stride(h, kind, n): only every n-th launch of that kind is timed (n >= 1, default 1), so that
 * sampling inside a timed region costs 1 / n of those bubbles.                                                  */
#define AFQ_K_PROPAGATOR 0   /* fused B exp(V) B kernel (k_fused.hip)          */
#define AFQ_K_EXCHANGE 1     /* Cholesky exchange-energy kernel (k_energy.hip) */
#define AFQ_K_VHS 2          /* HS potential GEMM                              */
#define AFQ_K_FORCE_BIAS 3   /* force-bias GEMM                                */
#define AFQ_K_GREENS 4       /* Green's function kernel (N <= 45 path)         */
#define AFQ_K_COUNT 5
int afq_kernel_trace(afq_handle *h, int on);
int afq_kernel_trace_stride(afq_handle *h, int kind, int stride);
int afq_kernel_trace_get(afq_handle *h, int kind, double *ms_out, int max_n, int *n_out);
/* Profile pass over EVERY launch of the library, whatever the configuration dispatches to: afq_launch_trace(h, 1)
 * clears and starts recording an event pair around each launch (up to 65536), keyed by the name the launch leaves in
 * the breadcrumb ring (the kernel for plain launches, the launching function for the GEMM engines: k_onebody,
 * k_vhs_generic ...); (h, 0) stops.  afq_launch_trace_get synchronises and returns, per distinct name, the summed
 * duration and the number of launches: names_out receives the names back to back, NUL-terminated.  The pairs cost a few
 * microseconds of bubble per launch: for an extra pass behind a timed region, not for the region itself.
 * (measurement hook, no reference counterpart)                                                                   */
int afq_launch_trace(afq_handle *h, int on);
int afq_launch_trace_get(afq_handle *h, char *names_out, int names_len, double *total_ms, int64_t *launches,
                         int max_names, int *n_out);
/* Flops the matrix pipe executed in the LAST launch of a kind: MFMA instructions x their flop count (2048 per
 * v_mfma_f64_16x16x4, 512 per 4x4x4), tile and contraction padding included, a 3-multiplication complex product
 * counted as 3.  issued / time / peak is the utilisation of the pipe; the algorithmic count of SURVEY 8d
 * (4 multiplications, no padding) over the same time is the figure comparable across implementations.
 * 0 for kinds that have not been launched or do not run on the matrix pipe.  (measurement hook, no reference
 * counterpart)                                                                                                    */
int afq_kernel_issued_flops(afq_handle *h, int kind, double *flops_out);
/* fused propagator (AFQ_K_PROPAGATOR): matrix-pipe flops ONE walker issues in the last launch's shape through the open-shell
 * deal and through the closed-shell deal (alpha slots only); afq_kernel_issued_flops reports nw x the open-shell count, the
 * count of a launch is (nw_live - n_closed) x open + n_closed x closed with n_closed from afq_counters [3]            */
int afq_propagator_issued_flops(afq_handle *h, double *open_per_walker, double *closed_per_walker);

#ifdef __cplusplus
}
#endif
#endif /* AFQMC_HIP_H */
