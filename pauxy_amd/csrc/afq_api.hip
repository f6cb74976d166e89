// C ABI of libafqmc_hip.so (see include/afqmc_hip.h): handle lifetime, uploads of
// the read-only inputs, walker storage, and the sequencing of the kernels that
// make up one propagation step (propagation/continuous.py:232-262).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "afq_internal.h"

namespace {

template <class T> int dev_alloc(afq_handle *h, T **p, size_t n) {
    if (*p) { hipFree(*p); *p = nullptr; }
    if (n == 0) return AFQ_OK;
    hipError_t e = hipMalloc((void **)p, n * sizeof(T));
    if (e != hipSuccess) { h->err = std::string("hipMalloc: ") + hipGetErrorString(e); return AFQ_ENOMEM; }
    return AFQ_OK;
}

template <class T> int dev_upload(afq_handle *h, T **p, const void *src, size_t n) {
    int rc = dev_alloc(h, p, n);
    if (rc) return rc;
    if (n) AFQ_HIP(h, hipMemcpy(*p, src, n * sizeof(T), hipMemcpyHostToDevice));
    return AFQ_OK;
}

template <class T> void dev_free(T *&p) { if (p) { hipFree(p); p = nullptr; } }

// ---- multi-determinant operand sets: psi / psic / rH1 / rchol_* of the handle are views of dets[cur_det]
void stash_det(afq_handle *h) {
    if (h->ndet <= 1) return;
    afq_handle::DetOps &o = h->dets[h->cur_det];
    o.psi = h->psi; o.psic = h->psic; o.rH1 = h->rH1; o.rchol_re = h->rchol_re; o.rchol_im = h->rchol_im;
    for (int s = 0; s < 2; ++s) { o.rchol_frag[s] = h->rchol_frag[s]; o.rchol_frag_im[s] = h->rchol_frag_im[s]; o.atil[s] = h->atil[s]; }
    o.rchol_same = h->rchol_same;
}

void select_det(afq_handle *h, int d) {
    if (h->ndet <= 1 || d == h->cur_det) return;
    stash_det(h);
    const afq_handle::DetOps &o = h->dets[d];
    h->psi = o.psi; h->psic = o.psic; h->rH1 = o.rH1; h->rchol_re = o.rchol_re; h->rchol_im = o.rchol_im;
    for (int s = 0; s < 2; ++s) { h->rchol_frag[s] = o.rchol_frag[s]; h->rchol_frag_im[s] = o.rchol_frag_im[s]; h->atil[s] = o.atil[s]; }
    h->rchol_same = o.rchol_same;
    h->cur_det = d;
    if (h->nw) {
        h->ghalf = h->ghalf_all + (size_t)d * h->nw * h->M * h->nt;
        if (h->vbias_all) h->vbias = h->vbias_all + (size_t)d * 2 * h->fb_split * h->nw * h->K;
    }
}

void free_dets(afq_handle *h) {
    if (h->ndet > 1) {
        stash_det(h);
        for (afq_handle::DetOps &o : h->dets) {
            if (o.psi) hipFree(o.psi);
            if (o.psic) hipFree(o.psic);
            if (o.rH1) hipFree(o.rH1);
            if (o.rchol_re) hipFree(o.rchol_re);
            if (o.rchol_im) hipFree(o.rchol_im);
            for (int s = 0; s < 2; ++s) {
                if (o.rchol_frag[s]) hipFree(o.rchol_frag[s]);
                if (o.rchol_frag_im[s]) hipFree(o.rchol_frag_im[s]);
            }
            k_free_atil(o.atil);
        }
        h->psi = nullptr; h->psic = nullptr; h->rH1 = nullptr; h->rchol_re = nullptr; h->rchol_im = nullptr;
        for (int s = 0; s < 2; ++s) { h->rchol_frag[s] = nullptr; h->rchol_frag_im[s] = nullptr; h->atil[s] = nullptr; }
    }
    h->dets.clear();
    h->ndet = 1; h->cur_det = 0;
    if (h->coeffs) { hipFree(h->coeffs); h->coeffs = nullptr; }
    dev_free(h->msd_psicT); h->msd_fb_gbar = false;
}

void free_system(afq_handle *h) {
    free_dets(h);
    dev_free(h->hs_pot); dev_free(h->hs_pair); dev_free(h->hs_pk); dev_free(h->L_full); dev_free(h->rchol_re); dev_free(h->rchol_im);
    for (int s = 0; s < 2; ++s) { dev_free(h->rchol_frag[s]); dev_free(h->rchol_frag_im[s]); }
    k_free_atil(h->atil);
    dev_free(h->H1); dev_free(h->rH1);
    dev_free(h->iA_colptr); dev_free(h->iA_row); dev_free(h->iA_val); dev_free(h->ell_row); dev_free(h->ell_val); dev_free(h->ueg_rmap); dev_free(h->ueg_rows);
    dev_free(h->ueg_kp); dev_free(h->ueg_pm); dev_free(h->ueg_koff); dev_free(h->ueg_poff);
    dev_free(h->iB_colptr); dev_free(h->iB_row); dev_free(h->iB_val);
    dev_free(h->iA_rowptr); dev_free(h->iA_col); dev_free(h->iA_rval);
    dev_free(h->iB_rowptr); dev_free(h->iB_col); dev_free(h->iB_rval);
    dev_free(h->kpq_off); dev_free(h->kpq_i); dev_free(h->kpq_kpq);
    dev_free(h->pmq_off); dev_free(h->pmq_i); dev_free(h->pmq_pmq);
    dev_free(h->vqvec); dev_free(h->H1diag);
    h->kind = 0;
}

void free_walkers(afq_handle *h) {
    dev_free(h->est_acc); h->est_acc_pending = false; h->fuse_est_req = false;
    dev_free(h->phi); dev_free(h->phi_t); dev_free(h->phi_t2);
    dev_free(h->weight); dev_free(h->unscaled); dev_free(h->detR); dev_free(h->log_detR);
    dev_free(h->ot); dev_free(h->ehyb); dev_free(h->phase); dev_free(h->eloc);
    dev_free(h->ghalf_all); h->ghalf = nullptr; dev_free(h->G); dev_free(h->ovlp_old); dev_free(h->ovlp_new);
    dev_free(h->xi); dev_free(h->vbias_all); h->vbias = nullptr; dev_free(h->ghalf_sum); h->gsum_version = 0; h->vbias_version = 0;
    dev_free(h->gdiag); h->gdiag_version = 0; h->gdiag_parts = 0;
    dev_free(h->detd); dev_free(h->detd_a); dev_free(h->detw); dev_free(h->energy_all);
    dev_free(h->msd_gs); dev_free(h->msd_S); h->msd_fb_gbar = false;
    dev_free(h->hs_oinv); dev_free(h->hs_u); dev_free(h->hs_fields); dev_free(h->hs_used); dev_free(h->hs_alive0);
    dev_free(h->hs_fbfac);
    dev_free(h->bp_hist); dev_free(h->bp_n); dev_free(h->bp_flag); dev_free(h->bp_cos); dev_free(h->bp_ph);
    dev_free(h->phi_old); dev_free(h->phi_bp); dev_free(h->BH1dag); dev_free(h->bp_xs); dev_free(h->bp_est);
    h->nbp = 0; dev_free(h->xbar); dev_free(h->xs);
    dev_free(h->cmf); dev_free(h->cfb); dev_free(h->vhs); dev_free(h->lu_ws);
    dev_free(h->gj_flag); dev_free(h->big_ws); dev_free(h->big_ws2); dev_free(h->detm); dev_free(h->dete); dev_free(h->qr_logd); dev_free(h->qr_fail);
    dev_free(h->energy); dev_free(h->exx_part); dev_free(h->gfrag); dev_free(h->exq_y); h->exq_y_len = 0;
    dev_free(h->alive); dev_free(h->parent_ix); dev_free(h->rdm_acc); h->rdm_on = false;
    dev_free(h->closed_w); h->closed_w_n = 0;
    if (h->pack_tmp) { hipFree(h->pack_tmp); h->pack_tmp = nullptr; }
    h->exx_part_len = 0; h->gfrag_bytes = 0; h->nw = 0;
}

// rH1[i, q] = sum_p conj(psi[p, i]) H1_s[p, q]  (s = spin of orbital i)
int build_rH1(afq_handle *h, const std::vector<double> &H1, const std::vector<double> &psi) {
    const int M = h->M, nt = h->nt;
    std::vector<double> r((size_t)nt * M * 2, 0.0);
    for (int i = 0; i < nt; ++i) {
        const int s = i < h->na ? 0 : 1;
        for (int p = 0; p < M; ++p) {
            const double cr = psi[((size_t)p * nt + i) * 2], ci = -psi[((size_t)p * nt + i) * 2 + 1];
            if (cr == 0.0 && ci == 0.0) continue;
            const double *hrow = &H1[(((size_t)s * M + p) * M) * 2];
            double *out = &r[(size_t)i * M * 2];
            for (int q = 0; q < M; ++q) {
                const double hr = hrow[2 * q], hi = hrow[2 * q + 1];
                out[2 * q] += cr * hr - ci * hi;
                out[2 * q + 1] += cr * hi + ci * hr;
            }
        }
    }
    return dev_upload(h, &h->rH1, r.data(), (size_t)nt * M);
}

}  // namespace

// host copies kept for derived operands (small: H1 and psi only)
struct afq_host_cache {
    std::vector<double> H1, psi;
};
static afq_host_cache *cache_of(afq_handle *h);

struct afq_handle_full : afq_handle {
    afq_host_cache cache;
};
static afq_host_cache *cache_of(afq_handle *h) { return &static_cast<afq_handle_full *>(h)->cache; }

static int upload_psi(afq_handle *h, const double *psi) {
    const size_t n = (size_t)h->M * h->nt;
    int rc = dev_upload(h, &h->psi, psi, n);
    if (rc) return rc;
    cache_of(h)->psi.assign(psi, psi + 2 * n);
    h->psi_real = true;
    for (size_t i = 0; i < n && h->psi_real; ++i) h->psi_real = psi[2 * i + 1] == 0.0;
    h->psi_closed = h->na == h->nb && h->na > 0;
    for (int p_ = 0; p_ < h->M && h->psi_closed; ++p_)
        h->psi_closed = memcmp(psi + 2 * (size_t)p_ * h->nt, psi + 2 * ((size_t)p_ * h->nt + h->na), sizeof(double) * 2 * h->na) == 0;
    std::vector<double> pc(psi, psi + 2 * n);
    for (size_t i = 0; i < n; ++i) pc[2 * i + 1] = -pc[2 * i + 1];
    if (h->ndet <= 1) {     // transposed copy (Hubbard force bias: diag of G from rows of conj(psi)^T and Ghalf)
        std::vector<double> pt(2 * n);
        for (int p_ = 0; p_ < h->M; ++p_)
            for (int i = 0; i < h->nt; ++i) {
                pt[2 * ((size_t)i * h->M + p_)] = pc[2 * ((size_t)p_ * h->nt + i)];
                pt[2 * ((size_t)i * h->M + p_) + 1] = pc[2 * ((size_t)p_ * h->nt + i) + 1];
            }
        int rc2 = dev_upload(h, &h->psicT, pt.data(), n);
        if (rc2) return rc2;
    }
    return dev_upload(h, &h->psic, pc.data(), n);
}

static int maybe_build_rH1(afq_handle *h) {
    afq_host_cache *c = cache_of(h);
    if (h->kind == 0 || !h->have_trial || c->H1.empty()) return AFQ_OK;
    return build_rH1(h, c->H1, c->psi);
}

__global__ void afq_marker_kernel(volatile unsigned long long *retired, unsigned long long n) { *retired = n; }

// afq_launch_trace: one event at the head of every launch (name != nullptr) and one behind it (name == nullptr, from
// afq_post_launch); launches queued without a post in between close their predecessor with their own head event.
void afq_launch_trace_mark(afq_handle *h, const char *name) {
    const size_t pairs = h->ltrace_name.size();
    if (name && pairs >= 65536) return;
    auto event_at = [&](size_t i) {
        while (h->ltrace_ev.size() <= i) { hipEvent_t e; hipEventCreate(&e); h->ltrace_ev.push_back(e); }
        return h->ltrace_ev[i];
    };
    if (h->ltrace_open) {            // close the launch in flight
        hipEventRecord(event_at(2 * pairs - 1), h->stream);
        h->ltrace_open = false;
    }
    if (name) {
        hipEventRecord(event_at(2 * pairs), h->stream);
        h->ltrace_name.push_back(name);
        h->ltrace_open = true;
    }
}

hipError_t afq_post_launch(afq_handle *h) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (h->ltrace_open) afq_launch_trace_mark(h, nullptr);
    if (h->debug_markers) {
        unsigned long long *dptr = nullptr;
        if (hipHostGetDevicePointer((void **)&dptr, (void *)h->retired, 0) == hipSuccess)
            hipLaunchKernelGGL(afq_marker_kernel, dim3(1), dim3(1), 0, h->stream, dptr, (unsigned long long)h->n_launch);
    }
    if (h->debug_sync) e = hipStreamSynchronize(h->stream);
    return e;
}

#define AFQ_API(h, name) do { if (h) (h)->crumb_api = name; } while (0)

extern "C" {

int afq_version(void) { return 2; }

int afq_create(int device_id, afq_handle **out) {
    if (!out) return AFQ_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return AFQ_EHIP;
    if (device_id < 0 || device_id >= ndev) return AFQ_EINVAL;
    if (hipSetDevice(device_id) != hipSuccess) return AFQ_EHIP;
    afq_handle_full *h = new afq_handle_full();
    h->device = device_id;
    if (hipStreamCreate(&h->stream) != hipSuccess) { delete h; return AFQ_EHIP; }
    hipEventCreate(&h->ev0); hipEventCreate(&h->ev1);
    if (hipMalloc(&h->estimates, sizeof(cplx) * AFQ_EST_COUNT_) != hipSuccess ||
        hipMalloc(&h->counters, sizeof(unsigned long long) * AFQ_NCOUNTERS) != hipSuccess ||
        hipMalloc(&h->closed_bad, sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&h->scal, sizeof(double) * AFQ_NSCAL) != hipSuccess) { delete h; return AFQ_ENOMEM; }
    hipMemset(h->estimates, 0, sizeof(cplx) * AFQ_EST_COUNT_);
    hipMemset(h->counters, 0, sizeof(unsigned long long) * AFQ_NCOUNTERS);
    hipMemset(h->closed_bad, 0, sizeof(unsigned long long));
    hipMemset(h->scal, 0, sizeof(double) * AFQ_NSCAL);
    if (hipMalloc(&h->zero_page, 256) != hipSuccess) { delete h; return AFQ_ENOMEM; }
    hipMemset(h->zero_page, 0, 256);
    // diagnostics (not tuning): AFQ_DEBUG_SYNC=1 synchronises and checks after every launch (a failing kernel is
    // named in afq_last_error), AFQ_DEBUG_MARKERS=1 queues a marker behind every launch so that afq_last_launch
    // can tell the last launch that RETIRED apart from the last one queued
    if (hipHostMalloc((void **)&h->retired, sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) {
        delete h; return AFQ_ENOMEM;
    }
    *h->retired = 0;
    h->debug_sync = getenv("AFQ_DEBUG_SYNC") != nullptr && atoi(getenv("AFQ_DEBUG_SYNC")) != 0;
    h->debug_markers = getenv("AFQ_DEBUG_MARKERS") != nullptr && atoi(getenv("AFQ_DEBUG_MARKERS")) != 0;
    h->no_ring = AFQ_KNOB_SET("AFQ_NO_RING");
    h->no_fused = AFQ_KNOB_SET("AFQ_NO_FUSED");
    h->no_vhs_upper = AFQ_KNOB_SET("AFQ_VHS_MIRROR");
    h->greens_cache = !AFQ_KNOB_SET("AFQ_NO_GREENS_CACHE");
    *out = h;
    return AFQ_OK;
}

int afq_destroy(afq_handle *h) {
    if (!h) return AFQ_EINVAL;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    k_comm_destroy(h);
    k_ueg_fast_free(h);
    free_walkers(h);
    free_system(h);
    dev_free(h->psi); dev_free(h->psic); dev_free(h->psicT); dev_free(h->BH1); dev_free(h->mf_shift);
    dev_free(h->estimates); dev_free(h->counters); dev_free(h->closed_bad); dev_free(h->scal);
    if (h->zero_page) hipFree(h->zero_page);
    if (h->retired) hipHostFree((void *)h->retired);
    hipEventDestroy(h->ev0); hipEventDestroy(h->ev1);
    if (h->est_stage) hipHostFree(h->est_stage);
    for (int k = 0; k < AFQ_K_COUNT; ++k) for (hipEvent_t e : h->ktrace_ev[k]) hipEventDestroy(e);
    for (hipEvent_t e : h->ltrace_ev) hipEventDestroy(e);
    hipStreamDestroy(h->stream);
    delete static_cast<afq_handle_full *>(h);
    return AFQ_OK;
}

const char *afq_last_error(afq_handle *h) { return h ? h->err.c_str() : "null handle"; }

int afq_sync(afq_handle *h) {
    if (!h) return AFQ_EINVAL;
    AFQ_HIP(h, hipStreamSynchronize(h->stream));
    return AFQ_OK;
}

int afq_stream(afq_handle *h, void **stream) {
    if (!h || !stream) return AFQ_EINVAL;
    *stream = (void *)h->stream;
    return AFQ_OK;
}

// ------------------------------------------------------------------ systems
static int set_dims(afq_handle *h, int kind, int M, int K, int na, int nb) {
    if (M <= 0 || K <= 0 || na <= 0 || nb < 0 || na > M || nb > M) AFQ_FAIL(h, AFQ_EINVAL, "bad dimensions");
    hipSetDevice(h->device);
    free_system(h);
    free_walkers(h);
    h->have_trial = false; h->have_prop = false;
    h->kind = kind; h->M = M; h->K = K; h->na = na; h->nb = nb; h->nt = na + nb;
    return AFQ_OK;
}

// half-rotated Cholesky vectors of ONE determinant into the handle's current operand slots:
// planar re / im [nt*M, ld_rc] for the force-bias GEMM + fragment-ordered copies for the exchange kernel
static bool rchol_is_real(const double *rchol, size_t n) {
    for (size_t i = 0; i < n; ++i) if (rchol[2 * i + 1] != 0.0) return false;
    return true;
}

static int upload_rchol(afq_handle *h, const double *rchol, bool real) {
    const size_t nq = (size_t)h->nt * h->M;
    const int K = h->K;
    int rc;
    std::vector<double> re(nq * h->ld_rc, 0.0);
    for (size_t q = 0; q < nq; ++q)
        for (int n = 0; n < K; ++n) re[q * h->ld_rc + n] = rchol[2 * (q * K + n)];
    h->rchol_real = real;
    ++h->ghalf_version;                 // force-bias partials contracted with the old vectors are stale
    h->rchol_same = h->na == h->nb && memcmp(rchol, rchol + 2 * (size_t)h->na * h->M * K, sizeof(double) * 2 * (size_t)h->na * h->M * K) == 0;
    k_free_atil(h->atil);                  // the quadratic-form operand belongs to the old vectors
    h->atil_unavailable = false;
    if ((rc = dev_upload(h, &h->rchol_re, re.data(), re.size()))) return rc;
    if (!real) {
        std::vector<double> im(nq * h->ld_rc, 0.0);
        for (size_t q = 0; q < nq; ++q)
            for (int n = 0; n < K; ++n) im[q * h->ld_rc + n] = rchol[2 * (q * K + n) + 1];
        if ((rc = dev_upload(h, &h->rchol_im, im.data(), im.size()))) return rc;
    }
    for (int s = 0; s < 2; ++s) {
        if (h->rchol_frag[s]) { hipFree(h->rchol_frag[s]); h->rchol_frag[s] = nullptr; }
        if (h->rchol_frag_im[s]) { hipFree(h->rchol_frag_im[s]); h->rchol_frag_im[s] = nullptr; }
    }
    return k_prepare_energy_operands(h, rchol);
}

int afq_set_system_generic(afq_handle *h, int M, int K, int na, int nb, const double *hs_pot,
                           const double *rchol, const double *H1, double ecore) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h || !hs_pot || !rchol || !H1) return AFQ_EINVAL;
    int rc = set_dims(h, AFQ_SYS_GENERIC, M, K, na, nb);
    if (rc) return rc;
    h->ecore = ecore;
    const size_t mm = (size_t)M * M, nq = (size_t)h->nt * M;
    {   // hs_pot^T : [K, ld_hs] so that a VHS B-fragment is contiguous (even, zero-padded rows: the
        // LDS-DMA path moves 16-byte pairs of doubles).  Cholesky matrices of real orbitals are
        // symmetric in (p, q); then only the columns p <= q are kept and the VHS GEMM does half the work.
        bool sym = !AFQ_KNOB_SET("AFQ_VHS_FULL");
        for (int p = 0; p < M && sym; ++p)
            for (int q = p + 1; q < M && sym; ++q) {
                const double *a = hs_pot + ((size_t)p * M + q) * K, *b = hs_pot + ((size_t)q * M + p) * K;
                for (int n = 0; n < K; ++n) if (a[n] != b[n]) { sym = false; break; }
            }
        h->hs_sym = sym;
        if (sym) {
            const size_t np = (size_t)M * (M + 1) / 2;
            h->ld_hs = (long)((np + 1) & ~(size_t)1);
            std::vector<double> t((size_t)K * h->ld_hs, 0.0);
            std::vector<int> pq(2 * np);
            size_t c = 0;
            for (int p = 0; p < M; ++p)
                for (int q = p; q < M; ++q, ++c) {
                    pq[2 * c] = p; pq[2 * c + 1] = q;
                    const double *a = hs_pot + ((size_t)p * M + q) * K;
                    for (int n = 0; n < K; ++n) t[(size_t)n * h->ld_hs + c] = a[n];
                }
            if ((rc = dev_upload(h, &h->hs_pot, t.data(), t.size()))) return rc;
            if ((rc = dev_upload(h, &h->hs_pair, pq.data(), np))) return rc;
        } else {
            h->ld_hs = (long)((mm + 1) & ~(size_t)1);
            std::vector<double> t((size_t)K * h->ld_hs, 0.0);
            for (size_t r = 0; r < mm; ++r)
                for (int n = 0; n < K; ++n) t[(size_t)n * h->ld_hs + r] = hs_pot[r * K + n];
            if ((rc = dev_upload(h, &h->hs_pot, t.data(), t.size()))) return rc;
        }
    }
    h->ld_rc = (long)((K + 1) & ~1);
    if ((rc = upload_rchol(h, rchol, rchol_is_real(rchol, nq * K)))) return rc;
    if ((rc = dev_upload(h, &h->H1, H1, 2 * mm))) return rc;
    cache_of(h)->H1.assign(H1, H1 + 4 * mm);
    return AFQ_OK;
}

int afq_set_system_hubbard(afq_handle *h, int M, int na, int nb, double U, const double *T) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h || !T) return AFQ_EINVAL;
    int rc = set_dims(h, AFQ_SYS_HUBBARD, M, M, na, nb);
    if (rc) return rc;
    h->U = U; h->ecore = 0.0;
    const size_t mm = (size_t)M * M;
    if ((rc = dev_upload(h, &h->H1, T, 2 * mm))) return rc;
    cache_of(h)->H1.assign(T, T + 4 * mm);
    return AFQ_OK;
}

static int csc_to_csr(afq_handle *h, int nrow, int ncol, const int64_t *cp, const int64_t *row, const double *val,
                      int64_t **rp_d, int64_t **col_d, cplx **val_d) {
    const int64_t nnz = cp[ncol];
    std::vector<int64_t> rp(nrow + 1, 0), col(nnz);
    std::vector<double> v(2 * nnz);
    for (int64_t z = 0; z < nnz; ++z) rp[row[z] + 1]++;
    for (int r = 0; r < nrow; ++r) rp[r + 1] += rp[r];
    std::vector<int64_t> fill(rp.begin(), rp.end() - 1);
    for (int c = 0; c < ncol; ++c)
        for (int64_t z = cp[c]; z < cp[c + 1]; ++z) {
            const int64_t d = fill[row[z]]++;
            col[d] = c; v[2 * d] = val[2 * z]; v[2 * d + 1] = val[2 * z + 1];
        }
    int rc;
    if ((rc = dev_upload(h, rp_d, rp.data(), rp.size()))) return rc;
    if ((rc = dev_upload(h, col_d, col.data(), col.size()))) return rc;
    return dev_upload(h, val_d, v.data(), (size_t)nnz);
}

int afq_set_system_ueg(afq_handle *h, int M, int nq, int na, int nb, const int64_t *iA_colptr,
                       const int64_t *iA_row, const double *iA_val, const int64_t *iB_colptr,
                       const int64_t *iB_row, const double *iB_val, const int64_t *kpq_off,
                       const int64_t *kpq_i, const int64_t *kpq_kpq, const int64_t *pmq_off,
                       const int64_t *pmq_i, const int64_t *pmq_pmq, const double *vqvec, double vol,
                       const double *H1diag, double ecore) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h || !iA_colptr || !iB_colptr || !kpq_off || !pmq_off || !vqvec || !H1diag) return AFQ_EINVAL;
    int rc = set_dims(h, AFQ_SYS_UEG, M, 2 * nq, na, nb);
    if (rc) return rc;
    h->nq = nq; h->vol = vol; h->ecore = ecore;
    h->nnzA = iA_colptr[nq]; h->nnzB = iB_colptr[nq];
    if ((rc = dev_upload(h, &h->iA_colptr, iA_colptr, (size_t)nq + 1))) return rc;
    if ((rc = dev_upload(h, &h->iA_row, iA_row, (size_t)h->nnzA))) return rc;
    if ((rc = dev_upload(h, &h->iA_val, iA_val, (size_t)h->nnzA))) return rc;
    if ((rc = dev_upload(h, &h->iB_colptr, iB_colptr, (size_t)nq + 1))) return rc;
    if ((rc = dev_upload(h, &h->iB_row, iB_row, (size_t)h->nnzB))) return rc;
    if ((rc = dev_upload(h, &h->iB_val, iB_val, (size_t)h->nnzB))) return rc;
    if ((rc = csc_to_csr(h, M * M, nq, iA_colptr, iA_row, iA_val, &h->iA_rowptr, &h->iA_col, &h->iA_rval))) return rc;
    if ((rc = csc_to_csr(h, M * M, nq, iB_colptr, iB_row, iB_val, &h->iB_rowptr, &h->iB_col, &h->iB_rval))) return rc;
    {   // column-ELL layout of [iA | iB]: adjacent columns (= adjacent threads) read adjacent entries
        int L = 0;
        for (int q = 0; q < nq; ++q) {
            L = std::max(L, (int)(iA_colptr[q + 1] - iA_colptr[q]));
            L = std::max(L, (int)(iB_colptr[q + 1] - iB_colptr[q]));
        }
        const size_t nc = (size_t)2 * nq;
        std::vector<int> er(nc * L, 0);
        std::vector<double> ev(2 * nc * L, 0.0);
        for (int c = 0; c < 2 * nq; ++c) {
            const bool isB = c >= nq;
            const int q = isB ? c - nq : c;
            const int64_t *cp = isB ? iB_colptr : iA_colptr, *rw = isB ? iB_row : iA_row;
            const double *vl = isB ? iB_val : iA_val;
            int k = 0;
            for (int64_t z = cp[q]; z < cp[q + 1]; ++z, ++k) {
                er[(size_t)k * nc + c] = (int)rw[z];
                ev[2 * ((size_t)k * nc + c)] = vl[2 * z]; ev[2 * ((size_t)k * nc + c) + 1] = vl[2 * z + 1];
            }
        }
        h->ell_len = L;
        if ((rc = dev_upload(h, &h->ell_row, er.data(), nc * L))) return rc;
        if ((rc = dev_upload(h, &h->ell_val, ev.data(), nc * L))) return rc;
    }
    if ((rc = dev_upload(h, &h->kpq_off, kpq_off, (size_t)nq + 1))) return rc;
    if ((rc = dev_upload(h, &h->kpq_i, kpq_i, (size_t)kpq_off[nq]))) return rc;
    if ((rc = dev_upload(h, &h->kpq_kpq, kpq_kpq, (size_t)kpq_off[nq]))) return rc;
    if ((rc = dev_upload(h, &h->pmq_off, pmq_off, (size_t)nq + 1))) return rc;
    if ((rc = dev_upload(h, &h->pmq_i, pmq_i, (size_t)pmq_off[nq]))) return rc;
    if ((rc = dev_upload(h, &h->pmq_pmq, pmq_pmq, (size_t)pmq_off[nq]))) return rc;
    {   // rows of G referenced by the energy's index lists (estimators/ueg.py:27-88)
        std::vector<int> rmap(M, -1), rows;
        for (int64_t z = 0; z < kpq_off[nq]; ++z) if (rmap[kpq_i[z]] < 0) { rmap[kpq_i[z]] = 1; }
        for (int64_t z = 0; z < pmq_off[nq]; ++z) if (rmap[pmq_i[z]] < 0) { rmap[pmq_i[z]] = 1; }
        for (int i = 0; i < M; ++i) if (rmap[i] > 0) { rmap[i] = (int)rows.size(); rows.push_back(i); }
        h->ueg_nrows = (int)rows.size();
        if ((rc = dev_upload(h, &h->ueg_rmap, rmap.data(), (size_t)M))) return rc;
        if ((rc = dev_upload(h, &h->ueg_rows, rows.data(), rows.size()))) return rc;
        // packed copies for energy_ueg_q_kernel (M and the staged row count fit 16 bits, the list lengths 31)
        dev_free(h->ueg_kp); dev_free(h->ueg_pm); dev_free(h->ueg_koff); dev_free(h->ueg_poff);
        if (M < 65536 && kpq_off[nq] < (1LL << 31) && pmq_off[nq] < (1LL << 31)) {
            std::vector<int> kp((size_t)kpq_off[nq]), pm((size_t)pmq_off[nq]), ko((size_t)nq + 1), po((size_t)nq + 1);
            for (int64_t z = 0; z < kpq_off[nq]; ++z) kp[z] = (rmap[kpq_i[z]] << 16) | (int)kpq_kpq[z];
            for (int64_t z = 0; z < pmq_off[nq]; ++z) pm[z] = (rmap[pmq_i[z]] << 16) | (int)pmq_pmq[z];
            for (int q = 0; q <= nq; ++q) { ko[q] = (int)kpq_off[q]; po[q] = (int)pmq_off[q]; }
            if ((rc = dev_upload(h, &h->ueg_kp, kp.data(), kp.size() ? kp.size() : 1))) return rc;
            if ((rc = dev_upload(h, &h->ueg_pm, pm.data(), pm.size() ? pm.size() : 1))) return rc;
            if ((rc = dev_upload(h, &h->ueg_koff, ko.data(), ko.size()))) return rc;
            if ((rc = dev_upload(h, &h->ueg_poff, po.data(), po.size()))) return rc;
        }
    }
    if ((rc = dev_upload(h, &h->vqvec, vqvec, (size_t)nq))) return rc;
    if ((rc = dev_upload(h, &h->H1diag, H1diag, (size_t)2 * M))) return rc;
    cache_of(h)->H1.clear();
    return k_ueg_fast_system(h, M, nq, iA_colptr, iA_row, iA_val, iB_colptr, iB_row, iB_val);
}

int afq_set_trial(afq_handle *h, const double *psi) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h || !psi) return AFQ_EINVAL;
    if (!h->kind) AFQ_FAIL(h, AFQ_ESTATE, "set the system before the trial");
    if (h->ndet > 1) AFQ_FAIL(h, AFQ_ESTATE, "a multi-determinant trial is set; set the system again first");
    hipSetDevice(h->device);
    int rc = upload_psi(h, psi);
    if (rc) return rc;
    h->have_trial = true;
    if (h->kind == AFQ_SYS_UEG && (rc = k_ueg_fast_trial(h, psi))) return rc;
    return maybe_build_rH1(h);
}

int afq_set_trial_multi(afq_handle *h, int ndet, const double *psi, const double *coeffs, const double *rchol) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h || !psi || !coeffs || !rchol || ndet < 1) return AFQ_EINVAL;
    if (h->kind != AFQ_SYS_GENERIC) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "multi-determinant trials need a generic system");
    if (h->nw) AFQ_FAIL(h, AFQ_ESTATE, "set the trial before allocating walkers");
    hipSetDevice(h->device);
    // drop the operands afq_set_system_generic uploaded for a single determinant
    free_dets(h);
    dev_free(h->psi); dev_free(h->psic); dev_free(h->rH1); dev_free(h->rchol_re); dev_free(h->rchol_im);
    for (int s = 0; s < 2; ++s) { dev_free(h->rchol_frag[s]); dev_free(h->rchol_frag_im[s]); }
    k_free_atil(h->atil);
    const size_t npsi = (size_t)h->M * h->nt, nrc = npsi * h->K;
    const bool real = rchol_is_real(rchol, nrc * ndet);
    h->ndet = ndet; h->cur_det = 0;
    h->dets.assign(ndet > 1 ? ndet : 0, afq_handle::DetOps());
    int rc;
    for (int d = 0; d < ndet; ++d) {
        if (ndet > 1) {
            // fresh (null) slots for determinant d; stash_det records what the uploads allocate
            if (d > 0) {
                stash_det(h);
                h->psi = nullptr; h->psic = nullptr; h->rH1 = nullptr; h->rchol_re = nullptr; h->rchol_im = nullptr;
                for (int s = 0; s < 2; ++s) { h->rchol_frag[s] = nullptr; h->rchol_frag_im[s] = nullptr; h->atil[s] = nullptr; }
                h->cur_det = d;
            }
        }
        if ((rc = upload_psi(h, psi + 2 * npsi * d))) return rc;
        if ((rc = upload_rchol(h, rchol + 2 * nrc * d, real))) return rc;
        h->have_trial = true;
        if ((rc = maybe_build_rH1(h))) return rc;
    }
    select_det(h, 0);
    if ((rc = dev_upload(h, &h->coeffs, coeffs, (size_t)ndet))) return rc;
    if (ndet > 1) {
        // conj(psi_d)^T of every determinant stacked, [ndet nt, M]: both operands of the averaged Green's function
        // (k_force_bias_msd_gbar) read it along the orbital index
        const int M = h->M, nt = h->nt;
        const size_t kkp = ((size_t)ndet * nt + 7) & ~(size_t)7;               // zero rows up to a whole chunk of 8
        std::vector<double> pt(2 * kkp * M, 0.0);
        for (int d = 0; d < ndet; ++d)
            for (int p_ = 0; p_ < M; ++p_)
                for (int i = 0; i < nt; ++i) {
                    const double *src = psi + 2 * (npsi * d + (size_t)p_ * nt + i);
                    double *dst = &pt[2 * (((size_t)d * nt + i) * M + p_)];
                    dst[0] = src[0]; dst[1] = -src[1];
                }
        if ((rc = dev_upload(h, &h->msd_psicT, pt.data(), kkp * M))) return rc;
    }
    return AFQ_OK;
}

int afq_set_propagator(afq_handle *h, const double *BH1, const double *mf_shift, double dt, int exp_order,
                       int flags) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h || !BH1 || !mf_shift || dt <= 0 || exp_order < 0) return AFQ_EINVAL;
    if (!h->kind) AFQ_FAIL(h, AFQ_ESTATE, "set the system before the propagator");
    hipSetDevice(h->device);
    int rc;
    if ((rc = dev_upload(h, &h->BH1, BH1, (size_t)2 * h->M * h->M))) return rc;
    h->bh1_real = true;
    h->bh1_same = memcmp(BH1, BH1 + (size_t)2 * h->M * h->M, sizeof(double) * 2 * h->M * h->M) == 0;
    for (size_t i = 0, n = (size_t)2 * h->M * h->M; i < n && h->bh1_real; ++i) h->bh1_real = BH1[2 * i + 1] == 0.0;
    if ((rc = dev_upload(h, &h->mf_shift, mf_shift, (size_t)h->K))) return rc;
    if (h->kind == AFQ_SYS_UEG && (rc = k_ueg_fast_propagator(h, BH1))) return rc;
    h->dt = dt; h->sqrt_dt = std::pow(dt, 0.5); h->exp_order = exp_order;
    if (flags & AFQ_PROP_FREE_PROJECTION) flags &= ~AFQ_PROP_FORCE_BIAS;   // continuous.py:30-33
    const int old_nv = h->nv; const bool old_diag = h->vhs_diag;
    h->flags = flags;
    h->vhs_diag = h->kind == AFQ_SYS_HUBBARD;
    h->nv = (h->kind == AFQ_SYS_HUBBARD && (flags & AFQ_PROP_HUBBARD_SPIN)) ? 2 : 1;
    h->have_prop = true;
    if (h->nw && (old_nv != h->nv || old_diag != h->vhs_diag || !h->vhs)) {
        // (diagonal potential: a second block of the same size holds its Taylor factors)
        const size_t per = h->vhs_diag ? (size_t)2 * h->nv * h->M : (size_t)h->nv * h->M * h->M;
        if ((rc = dev_alloc(h, &h->vhs, per * h->nw))) return rc;
    }
    return AFQ_OK;
}

// ------------------------------------------------------------------ walkers
int afq_walkers_alloc(afq_handle *h, int nw) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h || nw <= 0) return AFQ_EINVAL;
    if (!h->kind) AFQ_FAIL(h, AFQ_ESTATE, "set the system before allocating walkers");
    hipSetDevice(h->device);
    free_walkers(h);
    const size_t per = (size_t)h->M * h->nt, K = h->K, n = nw;
    int rc;
#define A_(ptr, cnt) if ((rc = dev_alloc(h, &(ptr), (cnt)))) return rc;
    A_(h->phi, per * n) A_(h->phi_t, per * n) A_(h->phi_t2, per * n)
    A_(h->weight, n) A_(h->unscaled, n) A_(h->detR, n) A_(h->log_detR, n)
    A_(h->ot, n) A_(h->ehyb, n) A_(h->phase, n) A_(h->eloc, n)
    A_(h->ghalf_all, per * n * h->ndet) A_(h->ovlp_old, n) A_(h->ovlp_new, n)
    h->ghalf = h->ghalf_all + (size_t)h->cur_det * n * per;
    if (h->ndet > 1) { A_(h->detd, n * h->ndet) A_(h->detd_a, n * h->ndet) A_(h->detw, n * h->ndet) A_(h->energy_all, 3 * n * h->ndet) }
    A_(h->xi, K * n) A_(h->xbar, K * n) A_(h->xs, K * n) A_(h->cmf, n) A_(h->cfb, n)
    A_(h->energy, 3 * n) A_(h->alive, n) A_(h->parent_ix, n)
    // force-bias contraction slices: enough wave-tasks to fill 1024 SIMDs
    h->fb_split = 1;
    if (h->kind == AFQ_SYS_GENERIC) {
        const long tiles = (long)((nw + 31) / 32) * ((h->K + 31) / 32) * 2;
        int sp = (int)std::max(1L, std::min(8L, 1024 / std::max(1L, tiles)));
        if (nw > 32) {
            // work-group-tiled kernel (64 walkers x 64 fields per work-group): aim at >= 512 work-groups
            const long wgt = (long)((nw + 63) / 64) * ((h->K + 63) / 64) * 2;
            sp = (int)std::max(1L, std::min(16L, (512 + wgt - 1) / wgt));
        }
        // both spins share the Cholesky block: the contraction runs once over Ghalf_a + Ghalf_b (k_force_bias_generic),
        // half as long -- measured at C3: 4 slices (8 partials for fields_kernel to add) beat 8 by 1 % of the step
        if (h->rchol_same && h->rchol_real && h->ndet == 1 && h->na == h->nb && nw > 32 && sp > 1) sp = (sp + 1) / 2;
        sp = AFQ_KNOB_INT("AFQ_FB_SPLIT", sp);
        const int nmax = std::max(h->na, h->nb) * h->M;
        while (sp > 1 && nmax / sp < 64) --sp;
        h->fb_split = sp;
        A_(h->vbias_all, (size_t)2 * sp * n * K * h->ndet)
        h->vbias = h->vbias_all + (size_t)h->cur_det * 2 * sp * n * K;
    } else if (h->kind == AFQ_SYS_UEG) {
        A_(h->vbias_all, n * K)
        h->vbias = h->vbias_all;
    }
    if (h->kind == AFQ_SYS_UEG) A_(h->G, (size_t)2 * h->M * h->M * n)
    {
        // Hubbard: diagonals of up to two HS matrices, and their Taylor factors behind them
        const size_t pv = h->kind == AFQ_SYS_HUBBARD ? (size_t)4 * h->M : (size_t)h->M * h->M;
        A_(h->vhs, pv * n)
    }
#undef A_
    AFQ_HIP(h, hipMalloc(&h->pack_tmp, std::max((size_t)nw * 2 * sizeof(int), (size_t)4096)));
    h->nw = nw;
    h->cap_frac = 0.0; h->cap_total = -1.0;       // a cap armed for an earlier population does not carry over
    {   // walker.total_weight starts as the population size (walkers/handler.py:164); with a communicator
        // the population is nw walkers on every rank
        double sc0[AFQ_NSCAL] = {0};
        sc0[0] = (double)nw * k_comm_size(h);
        AFQ_HIP(h, hipMemcpy(h->scal, sc0, sizeof(sc0), hipMemcpyHostToDevice));
    }
    // defaults of walkers/walker.py:24-61
    std::vector<double> one(nw, 1.0), zero2(2 * (size_t)nw, 0.0), one2(2 * (size_t)nw, 0.0);
    for (int i = 0; i < nw; ++i) one2[2 * i] = 1.0;
    AFQ_HIP(h, hipMemcpy(h->weight, one.data(), sizeof(double) * nw, hipMemcpyHostToDevice));
    AFQ_HIP(h, hipMemcpy(h->unscaled, one.data(), sizeof(double) * nw, hipMemcpyHostToDevice));
    AFQ_HIP(h, hipMemcpy(h->detR, one.data(), sizeof(double) * nw, hipMemcpyHostToDevice));
    AFQ_HIP(h, hipMemset(h->log_detR, 0, sizeof(double) * nw));
    AFQ_HIP(h, hipMemcpy(h->ot, one2.data(), sizeof(cplx) * nw, hipMemcpyHostToDevice));
    AFQ_HIP(h, hipMemcpy(h->phase, one2.data(), sizeof(cplx) * nw, hipMemcpyHostToDevice));
    AFQ_HIP(h, hipMemcpy(h->ehyb, zero2.data(), sizeof(cplx) * nw, hipMemcpyHostToDevice));
    AFQ_HIP(h, hipMemcpy(h->eloc, zero2.data(), sizeof(cplx) * nw, hipMemcpyHostToDevice));
    AFQ_HIP(h, hipMemset(h->phi, 0, sizeof(cplx) * per * n));
    AFQ_HIP(h, hipMemset(h->ghalf, 0, sizeof(cplx) * per * n));
    AFQ_HIP(h, hipMemset(h->energy, 0, sizeof(cplx) * 3 * n));
    AFQ_HIP(h, hipMemset(h->xi, 0, sizeof(double) * K * n));
    return AFQ_OK;
}

static int field_info(afq_handle *h, int field, void **base, size_t *bytes) {
    const size_t per = (size_t)h->M * h->nt;
    switch (field) {
    case AFQ_F_PHI: *base = h->phi; *bytes = per * sizeof(cplx); break;
    case AFQ_F_WEIGHT: *base = h->weight; *bytes = sizeof(double); break;
    case AFQ_F_UNSCALED_WEIGHT: *base = h->unscaled; *bytes = sizeof(double); break;
    case AFQ_F_OT: *base = h->ot; *bytes = sizeof(cplx); break;
    case AFQ_F_HYBRID_ENERGY: *base = h->ehyb; *bytes = sizeof(cplx); break;
    case AFQ_F_PHASE: *base = h->phase; *bytes = sizeof(cplx); break;
    case AFQ_F_DETR: *base = h->detR; *bytes = sizeof(double); break;
    case AFQ_F_ELOC: *base = h->eloc; *bytes = sizeof(cplx); break;
    case AFQ_F_GHALF: *base = h->ghalf; *bytes = per * sizeof(cplx); break;
    case AFQ_F_G: *base = h->G; *bytes = (size_t)2 * h->M * h->M * sizeof(cplx); break;
    case AFQ_F_XBAR: *base = h->xbar; *bytes = (size_t)h->K * sizeof(cplx); break;
    case AFQ_F_XSHIFTED: *base = h->xs; *bytes = (size_t)h->K * sizeof(cplx); break;
    case AFQ_F_ENERGY: *base = h->energy; *bytes = 3 * sizeof(cplx); break;
    case AFQ_F_LOG_DETR: *base = h->log_detR; *bytes = sizeof(double); break;
    default: AFQ_FAIL(h, AFQ_EINVAL, "unknown walker field");
    }
    if (!*base) AFQ_FAIL(h, AFQ_ESTATE, "walker field not allocated (afq_walkers_alloc / afq_greens first)");
    return AFQ_OK;
}

static int ensure_spin_ghalf(afq_handle *h);

int afq_walkers_set(afq_handle *h, int field, const void *host, int first, int count) {
    if (h && (field == AFQ_F_PHI || field == AFQ_F_GHALF)) { h->greens_valid = false; h->gsum_only = false; ++h->ghalf_version; }
    if (!h || !host) return AFQ_EINVAL;
    if (!h->nw) AFQ_FAIL(h, AFQ_ESTATE, "no walkers allocated");
    if (first < 0 || count < 0 || first + count > h->nw) AFQ_FAIL(h, AFQ_EINVAL, "walker range out of bounds");
    hipSetDevice(h->device);
    void *base; size_t bytes;
    int rc = field_info(h, field, &base, &bytes);
    if (rc) return rc;
    AFQ_HIP(h, hipMemcpyAsync((char *)base + bytes * first, host, bytes * count, hipMemcpyHostToDevice, h->stream));
    AFQ_HIP(h, hipStreamSynchronize(h->stream));
    return AFQ_OK;
}

int afq_walkers_get(afq_handle *h, int field, void *host, int first, int count) {
    if (!h || !host) return AFQ_EINVAL;
    if (!h->nw) AFQ_FAIL(h, AFQ_ESTATE, "no walkers allocated");
    if (first < 0 || count < 0 || first + count > h->nw) AFQ_FAIL(h, AFQ_EINVAL, "walker range out of bounds");
    hipSetDevice(h->device);
    void *base; size_t bytes;
    int rc = field_info(h, field, &base, &bytes);
    if (rc) return rc;
    if (field == AFQ_F_GHALF && (rc = ensure_spin_ghalf(h))) return rc;
    AFQ_HIP(h, hipMemcpyAsync(host, (char *)base + bytes * first, bytes * count, hipMemcpyDeviceToHost, h->stream));
    AFQ_HIP(h, hipStreamSynchronize(h->stream));
    return AFQ_OK;
}

int afq_walkers_device_ptr(afq_handle *h, int field, void **dev_ptr, int64_t *bytes_per_walker) {
    if (!h || !dev_ptr) return AFQ_EINVAL;
    void *base; size_t bytes;
    int rc = field_info(h, field, &base, &bytes);
    if (rc) return rc;
    // a reader of the per-spin Ghalf must find the CURRENT walkers' there: a step announced with afq_estimates_fuse_next
    // leaves only overlap + spin sum behind
    if (field == AFQ_F_GHALF && h->gsum_only) {
        hipSetDevice(h->device);
        if ((rc = ensure_spin_ghalf(h))) return rc;
    }
    h->greens_valid = false; h->gsum_only = false; h->greens_cache = false;             // the caller may write through the pointer
    *dev_ptr = base;
    if (bytes_per_walker) *bytes_per_walker = (int64_t)bytes;
    return AFQ_OK;
}

// ----------------------------------------------------------------- hot path
static int greens_any(afq_handle *h, cplx *det_out, bool with_ghalf);

static int need_ready(afq_handle *h, bool prop) {
    if (!h->kind || !h->have_trial || !h->nw) AFQ_FAIL(h, AFQ_ESTATE, "system, trial and walkers must be set");
    if (prop && !h->have_prop) AFQ_FAIL(h, AFQ_ESTATE, "propagator not set");
    if (h->prop_pending) AFQ_FAIL(h, AFQ_ESTATE, "a step is half done: afq_propagate_finish first");
    hipSetDevice(h->device);
    return AFQ_OK;
}

// The last step left its Green's function behind as overlap + spin sum of Ghalf only (afq_propagate_finish on a step the
// driver announced with afq_estimates_fuse_next): somebody wants the per-spin Ghalf after all -- evaluate it.
static int ensure_spin_ghalf(afq_handle *h) {
    if (!h->gsum_only) return AFQ_OK;
    h->gsum_only = false;
    const int rc = greens_any(h, h->ovlp_new, true);
    h->greens_valid = rc == AFQ_OK;
    return rc;
}

static int ensure_G(afq_handle *h) {
    if (!h->G) {
        int rc = dev_alloc(h, &h->G, (size_t)2 * h->M * h->M * h->nw);
        if (rc) return rc;
    }
    return AFQ_OK;
}

static int copy_out(afq_handle *h, void *host, const void *dev, size_t bytes) {
    if (!host) return AFQ_OK;
    AFQ_HIP(h, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, h->stream));
    AFQ_HIP(h, hipStreamSynchronize(h->stream));
    return AFQ_OK;
}

int afq_greens(afq_handle *h, int want_G, double *ovlp_out) {
    AFQ_API(h, "afq_greens");
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h) return AFQ_EINVAL;
    int rc = need_ready(h, false);
    if (rc) return rc;
    {
        PhaseTimer t(h, T_GREENS);
        if ((rc = greens_any(h, h->ovlp_old, true))) return rc;
        if (want_G || h->kind == AFQ_SYS_UEG) {
            if ((rc = ensure_G(h))) return rc;
            if ((rc = k_full_G(h))) return rc;
        }
    }
    return copy_out(h, ovlp_out, h->ovlp_old, sizeof(cplx) * h->nw);
}

int afq_inverse_overlap(afq_handle *h, double *oinv_out, double *ovlp_out) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h || !oinv_out) return AFQ_EINVAL;
    int rc = need_ready(h, false);
    if (rc) return rc;
    if (h->ndet > 1) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "inverse overlaps: single-determinant trials");
    const size_t nmax = h->na > h->nb ? h->na : h->nb, n = (size_t)h->nw * 2 * nmax * nmax;
    cplx *tmp = nullptr;
    if ((rc = dev_alloc(h, &tmp, n))) return rc;
    AFQ_HIP(h, hipMemsetAsync(tmp, 0, sizeof(cplx) * n, h->stream));
    if ((rc = k_alive(h))) { dev_free(tmp); return rc; }
    // every walker, dead or alive: flag them all for this call
    std::vector<int> ones(h->nw, 1);
    AFQ_HIP(h, hipMemcpyAsync(h->alive, ones.data(), sizeof(int) * h->nw, hipMemcpyHostToDevice, h->stream));
    rc = k_inverse_overlap(h, tmp, h->ovlp_new);
    if (!rc) rc = copy_out(h, oinv_out, tmp, sizeof(cplx) * n);
    if (!rc && ovlp_out) rc = copy_out(h, ovlp_out, h->ovlp_new, sizeof(cplx) * h->nw);
    dev_free(tmp);
    if (!rc) rc = k_alive(h);
    return rc;
}

int afq_calc_overlap(afq_handle *h, double *ovlp_out) {
    AFQ_API(h, "afq_calc_overlap");
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h) return AFQ_EINVAL;
    int rc = need_ready(h, false);
    if (rc) return rc;
    {
        PhaseTimer t(h, T_OVLP);
        if ((rc = greens_any(h, h->ovlp_new, false))) return rc;
    }
    return copy_out(h, ovlp_out, h->ovlp_new, sizeof(cplx) * h->nw);
}

// force bias from the current Ghalf / G -> h->xbar (unclipped)
// Green's function (with_ghalf) or overlap only, for every determinant of the trial.  Multi-determinant:
// per-determinant Ghalf_d / <D_d|phi> into the slices, then weights conj(c_d) <D_d|phi> and their sum
// (walkers/multi_det.py:194-229 and :135-162).
static int greens_any(afq_handle *h, cplx *det_out, bool with_ghalf) {
    int rc;
    if (h->ndet <= 1) return with_ghalf ? k_greens(h, det_out) : k_overlap(h, det_out);
    for (int d = 0; d < h->ndet; ++d) {
        select_det(h, d);
        cplx *dd = h->detd + (size_t)d * h->nw;
        h->det_a_out = with_ghalf ? h->detd_a + (size_t)d * h->nw : nullptr;
        rc = with_ghalf ? k_greens(h, dd) : k_overlap(h, dd);
        h->det_a_out = nullptr;
        if (rc) { select_det(h, 0); return rc; }
    }
    select_det(h, 0);
    return k_msd_combine(h, det_out, with_ghalf);       // (multi_det.py:209,218 skip; calc_overlap :135-162 does not)
}

static int force_bias(afq_handle *h, bool with_xbar = true) {
    int rc;
    if (h->flags & AFQ_PROP_FORCE_BIAS) {
        if (h->kind == AFQ_SYS_GENERIC) {
            h->msd_fb_gbar = false;
            // (automatic mode: when every determinant's own contraction is current -- the Coulomb vectors of the energy
            //  evaluation that has just run on these Green's functions -- their weighted average costs nothing)
            if (k_msd_gbar_wanted(h) && !(h->msd_fb_mode == 0 && k_msd_vbias_current(h))) {
                // the reference's own formulation: ONE contraction with the determinant-averaged G (generic.py:154-157)
                if ((rc = k_force_bias_msd_gbar(h))) return rc;
                h->msd_fb_gbar = true;
            } else {
                for (int d = 0; d < h->ndet; ++d) {
                    select_det(h, d);
                    if ((rc = k_force_bias_generic(h))) { select_det(h, 0); return rc; }
                }
                select_det(h, 0);
            }
        }
        else if (h->kind == AFQ_SYS_UEG) { if ((rc = k_vbias_ueg(h))) return rc; }
    }
    return with_xbar ? k_xbar(h) : AFQ_OK;
}

static int build_vhs(afq_handle *h) {
    if (h->kind == AFQ_SYS_GENERIC) return k_vhs_generic(h);
    if (h->kind == AFQ_SYS_HUBBARD) return k_vhs_hubbard(h);
    return k_vhs_ueg(h);
}

static int local_energy(afq_handle *h);
static int local_energy_dets(afq_handle *h);

static int apply_exp(afq_handle *h, const cplx *vhs) {
    if (h->vhs_diag) return k_apply_exponential_diag(h, vhs);
    return k_apply_exponential(h, vhs);
}

int afq_propagate(afq_handle *h, const double *xi, double eshift_re, double eshift_im) {
    const int rc = afq_propagate_begin(h, xi);
    return rc ? rc : afq_propagate_finish(h, eshift_re, eshift_im);
}

int afq_propagate_begin(afq_handle *h, const double *xi) {
    AFQ_API(h, "afq_propagate");
    if (!h) return AFQ_EINVAL;
    int rc = need_ready(h, true);
    if (rc) return rc;
    if (h->hirsch) AFQ_FAIL(h, AFQ_ESTATE, "discrete Hirsch propagator set: use afq_propagate_hirsch");
    // Hubbard, continuous fields, one field per site: fields_kernel makes the fields (and draws them), the diagonal HS
    // potential and its Taylor factors.  ONE flag decides the inline draw, the branch below and who writes the factors.
    const bool hubbard_fused = h->vhs_diag && !h->no_fused;
    const bool hubbard_fused_fields = hubbard_fused && h->K == h->M;
    if (xi) {
        if ((rc = k_alive(h))) return rc;
        AFQ_HIP(h, hipMemcpyAsync(h->xi, xi, sizeof(double) * (size_t)h->nw * h->K, hipMemcpyHostToDevice, h->stream));
    } else if (k_prop_fused_supported(h) || k_ueg_fast_supported(h) || hubbard_fused_fields) {
        // nothing ahead of fields_kernel reads the fields or the alive flags on this path (the Green's function is
        // evaluated for every walker, the one-body product sits inside the fused propagator): fields_kernel draws
        // the same Philox stream itself and sets the flags
        h->rng_inline = true;
        h->rng_inline_counter = h->rng_counter++;
    } else {
        if ((rc = k_rng_normal(h))) return rc;           // draws the fields and sets the alive flags
    }
    const bool fp = (h->flags & AFQ_PROP_FREE_PROJECTION) != 0;
    if (!fp || (h->flags & AFQ_PROP_FORCE_BIAS)) {
        PhaseTimer t(h, T_GREENS);                         // continuous.py:245
        // (the spin sum alone will not do when this step needs more than the force bias: local-energy weights, one_rdm)
        const bool sum_ok = h->gsum_only && (h->flags & AFQ_PROP_HYBRID) && !h->rdm_on &&
                            (h->kind == AFQ_SYS_GENERIC ||
                             (h->kind == AFQ_SYS_HUBBARD && h->gdiag && h->gdiag_version == h->ghalf_version));
        if (h->greens_valid || sum_ok) std::swap(h->ovlp_old, h->ovlp_new);   // computed at the end of the last step
                                                                   // (gsum_only: as overlap + spin sum of Ghalf, all the force bias reads)
        else if ((rc = greens_any(h, h->ovlp_old, true))) return rc;
        h->greens_valid = false; h->gsum_only = false;
        const bool le = !fp && !(h->flags & AFQ_PROP_HYBRID);
        // (the plane-wave fast step gathers the force bias from the occupied rows of G built from Ghalf: no full G)
        if ((h->kind == AFQ_SYS_UEG && ((h->flags & AFQ_PROP_FORCE_BIAS) || le) && !k_ueg_fast_supported(h)) ||
            (h->rdm_on && h->ndet == 1)) {
            // (one_rdm: walker.G = the Green's function of the walker before this step, continuous.py:245)
            if ((rc = ensure_G(h))) return rc;
            if ((rc = k_full_G(h))) return rc;
        }
        if (le) {
            // walker.local_energy(system) of the un-propagated walker (continuous.py:296)
            PhaseTimer te(h, T_ENERGY);
            // Multi-determinant: the reference combines the per-determinant energies of the UN-propagated
            // walker with the weights calc_overlap refreshes for the PROPAGATED walker
            // (continuous.py:296 after :261, multi_det.py:160), so only the per-determinant part runs here.
            if (h->ndet > 1) { if ((rc = local_energy_dets(h))) return rc; }
            else if ((rc = local_energy(h))) return rc;
        }
    }
    if (k_ueg_fast_supported(h)) {
        // plane waves: force bias from the occupied rows of G, fields, and the ~2 nq coefficients that ARE the HS
        // potential, in one launch; B exp(V) B from those coefficients in a second one (k_ueg.hip)
        // (one launch -- ueg_step_kernel -- when the propagator's LDS arrays fit behind the field part's, else two)
        { PhaseTimer t(h, T_EXP); if ((rc = k_ueg_step(h))) return rc; }                         // :133-171, :251, :258
        h->prop_pending = true;
        return AFQ_OK;
    }
    if (hubbard_fused) {
        // Hubbard, continuous fields: the HS potential is diagonal, exp(V) a row scaling.  The force bias reads Ghalf of
        // the un-propagated walker, so fields, potential and the scaling factors are made FIRST and the factors ride on
        // the store of the first one-body product: phi <- B [exp(V) (B phi)] in two GEMM launches, the walkers pass
        // through memory twice instead of three times (exp_diag_kernel: 537 MB of traffic at C4).
        cplx *fac = h->vhs + (size_t)h->nw * h->nv * h->M;                              // second half of the vhs buffer
        const bool fields_make_factors = hubbard_fused_fields;                          // (one field per site)
        {
            PhaseTimer t(h, T_FB);                                                      // :133-158
            if ((rc = force_bias(h, false))) return rc;
            // ... and with the fields the diagonal HS potential (:161) and its Taylor factors (:162-171): the same
            // arithmetic as vhs_hubbard_kernel + exp_diag_factor_kernel, two launches less
            if ((rc = k_xbar_fields(h, fields_make_factors ? fac : nullptr))) return rc;
        }
        if (!fields_make_factors) {
            { PhaseTimer t(h, T_VHS); if ((rc = build_vhs(h))) return rc; }             // :161
            { PhaseTimer t(h, T_EXP); if ((rc = k_exp_diag_factors(h, h->vhs, fac))) return rc; }   // :162-171
        }
        if (h->rng_inline) AFQ_FAIL(h, AFQ_ESTATE, "internal: the inline field draw of this step was not consumed by the field kernel");
        { PhaseTimer t(h, T_ONEBODY); if ((rc = k_onebody(h, fac))) return rc; }        // :251 + the row scaling
        { PhaseTimer t(h, T_ONEBODY); if ((rc = k_onebody(h))) return rc; }             // :258
        h->prop_pending = true;
        return AFQ_OK;
    }
    // The force bias reads Ghalf of the un-propagated walker, so building the HS potential commutes
    // with the first one-body product; the fused path uses that to run B exp(V) B in one launch.
    const bool fused = k_prop_fused_supported(h);
    // Large systems (the GEMM chain): one matrix serves both spins and the HS potential never depends on the spin, so a walker
    // whose spin blocks are bitwise equal -- every walker of a run that starts closed-shell -- keeps them equal through
    // B exp(V) B.  Checked on the walkers themselves at every step (closed_flags_kernel); the GEMMs then leave out the tiles of
    // the beta columns of such walkers and the alpha block is copied over the beta block behind the closing one-body product.
    h->closed_large = !fused && h->kind == AFQ_SYS_GENERIC && h->nv == 1 && h->bh1_same && h->na == h->nb && h->na > 0 &&
                      h->M > 128 && h->nt > 32 && h->nw >= 64 && !h->no_ring && !AFQ_KNOB_SET("AFQ_NO_CLOSED_LARGE");
    struct ClosedLargeOff { afq_handle *h; ~ClosedLargeOff() { h->closed_large = false; } } closed_large_off{h};   // on every way out
    if (h->closed_large && (rc = k_closed_flags(h))) return rc;
    if (!fused) { PhaseTimer t(h, T_ONEBODY); if ((rc = k_onebody(h))) return rc; }   // :251
    {
        PhaseTimer t(h, T_FB);                                                      // :133-158
        if ((rc = force_bias(h, false))) return rc;
        if ((rc = k_xbar_fields(h))) return rc;
    }
    // symmetric L_n and the fused propagator as the only consumer: the HS potential is stored as its upper
    // triangle only (no scattered mirror writes) and the propagator fetches V[k][row] for k < row
    h->vhs_upper = fused && h->hs_sym && !h->no_vhs_upper;
    { PhaseTimer t(h, T_VHS); rc = build_vhs(h); }                                  // :161
    if (rc) { h->vhs_upper = false; return rc; }
    if (fused) {
        PhaseTimer t(h, T_EXP);                                                     // :251, :162-171, :258
        rc = k_prop_fused(h);
        h->vhs_upper = false;
        if (rc) return rc;
    } else {
        { PhaseTimer t(h, T_EXP); if ((rc = apply_exp(h, h->vhs))) return rc; }    // :162-171
        { PhaseTimer t(h, T_ONEBODY); if ((rc = k_onebody(h))) return rc; }        // :258
        if (h->closed_large && (rc = k_closed_copy_beta(h))) return rc;
    }
    h->prop_pending = true;
    return AFQ_OK;
}

int afq_propagate_finish(afq_handle *h, double eshift_re, double eshift_im) {
    AFQ_API(h, "afq_propagate");
    if (!h) return AFQ_EINVAL;
    if (!h->prop_pending) AFQ_FAIL(h, AFQ_ESTATE, "afq_propagate_finish without afq_propagate_begin");
    hipSetDevice(h->device);
    h->prop_pending = false;
    int rc;
    const bool fp = (h->flags & AFQ_PROP_FREE_PROJECTION) != 0;
    {
        PhaseTimer t(h, T_OVLP);                                                    // :261-262
        // The overlap of the propagated walker is the determinant of the matrix whose inverse the next
        // step's Green's function needs (continuous.py:245 of step n+1), so factorise once: this call
        // leaves Ghalf of the NEW phi behind and the next afq_propagate / afq_estimates_update reuses it.
        const bool le_msd = h->ndet > 1 && !fp && !(h->flags & AFQ_PROP_HYBRID);
        h->fuse_weight_done = false;
        h->fuse_weight_req = h->ndet == 1 && !le_msd;       // single determinant: k_greens may take the weight update along
        h->fuse_eshift = cmake(eshift_re, eshift_im);
        if (h->greens_cache && !fp) {
            // afq_estimates_fuse_next told us that nothing but the next step's force bias will read this Green's function
            // (no comb, no energy, no block end behind this step): when that force bias contracts the spin sum
            // Ghalf_a + Ghalf_b (k_fb_use_sum), the per-spin Ghalf -- 2/3 of the kernel's stores, 20 MB per step at C3 --
            // is not written at all.  Everything else sees "no cached Green's function" and recomputes if it asks.
            // (Hubbard, continuous fields, N > 45: the force bias reads the diagonal sums the Ghalf GEMM leaves behind --
            //  268 MB of Ghalf per step at C4 that nobody reads)
            const bool fb_sum = h->kind == AFQ_SYS_GENERIC && k_fb_use_sum(h);
            const bool fb_diag = h->kind == AFQ_SYS_HUBBARD && !h->hirsch && h->psicT && k_greens_big_supported(h);
            h->ghalf_skip_store = h->fuse_est_req && h->ndet == 1 && (fb_sum || fb_diag) &&
                                  (h->flags & AFQ_PROP_HYBRID) && (h->flags & AFQ_PROP_FORCE_BIAS) && !h->rdm_on &&
                                  h->nbp == 0 && h->psi_stride == 0 && !AFQ_KNOB_SET("AFQ_NO_GHALF_SKIP");
            h->ghalf_skipped = false;
            rc = greens_any(h, h->ovlp_new, true);
            h->fuse_weight_req = false;
            h->ghalf_skip_store = false;
            if (rc) return rc;
            h->greens_valid = !h->ghalf_skipped;
            h->gsum_only = h->ghalf_skipped;
        } else {
            rc = greens_any(h, h->ovlp_new, false);
            h->fuse_weight_req = false;
            if (rc) return rc;
        }
        if (h->ndet > 1 && !fp && !(h->flags & AFQ_PROP_HYBRID)) {
            if ((rc = k_msd_energy_combine(h))) return rc;
        }
        if ((rc = k_update_weight(h, cmake(eshift_re, eshift_im)))) return rc;
        if (h->nbp > 0 && !fp && (rc = k_bp_push(h))) return rc;     // FieldConfig.update (continuous.py:288-289)
    }
    if (h->fuse_est_req) {                                // afq_estimates_fuse_next: the weight update took the terms along
        h->fuse_est_req = false;
        h->est_acc_pending = true;
    }
    return AFQ_OK;
}

int afq_estimates_fuse_next(afq_handle *h) {
    if (!h) return AFQ_EINVAL;
    if (!h->nw) AFQ_FAIL(h, AFQ_ESTATE, "no walkers");
    if (h->rdm_on) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "afq_estimates_fuse_next with the one-body RDM accumulation on");
    if (h->hirsch) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "afq_estimates_fuse_next: continuous propagator only");
    hipSetDevice(h->device);
    if (!h->est_acc) {
        int rc = dev_alloc(h, &h->est_acc, (size_t)6 * h->nw);
        if (rc) return rc;
        AFQ_HIP(h, hipMemsetAsync(h->est_acc, 0, sizeof(double) * 6 * h->nw, h->stream));
    }
    h->fuse_est_req = true;
    return AFQ_OK;
}

// use_log_shift (walkers/handler.py:45,228,456-475).  The shifts are the same for every walker (walker.py:49-52 and
// the update at handler.py:471-474), so they live on the handle; log_detR is walker state.
int afq_set_log_shift(afq_handle *h, int on, double log_shift, double detR_shift) {
    AFQ_API(h, "afq_set_log_shift");
    if (!h) return AFQ_EINVAL;
    if (on && h->ndet > 1) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "use_log_shift: single-determinant walkers only");
    h->log_shift_on = on != 0;
    h->log_shift = on ? log_shift : 0.0;
    h->detR_shift = on ? detR_shift : 0.0;
    return AFQ_OK;
}

int afq_log_ovlp_sums(afq_handle *h, double *sums3) {
    AFQ_API(h, "afq_log_ovlp_sums");
    if (!h || !sums3) return AFQ_EINVAL;
    int rc = need_ready(h, false);
    if (rc) return rc;
    return k_log_ovlp_sums(h, sums3);
}

int afq_reortho(afq_handle *h, double *detR_out) {
    AFQ_API(h, "afq_reortho");
    if (!h) return AFQ_EINVAL;
    int rc = need_ready(h, false);
    if (rc) { h->greens_valid = false; h->gsum_only = false; return rc; }
    // Ghalf = (phi^T psi*)^-1 phi^T does not change when phi -> phi R^-1, so a Green's function kept from
    // the end of the last step stays valid across the QR; only the cached overlap picks up 1 / det R.
    const bool keep = (h->greens_valid || h->gsum_only) && h->ndet == 1;
    const bool sum_only = h->gsum_only;                  // (the spin sum of Ghalf is as invariant as Ghalf itself)
    h->greens_valid = false; h->gsum_only = false;
    bool scaled = false;
    { PhaseTimer t(h, T_QR); if ((rc = k_reortho(h, keep ? h->ovlp_new : nullptr, &scaled))) return rc; }
    if (keep) {
        if (!scaled && (rc = k_scale_by_inverse(h, h->ovlp_new, h->detR))) return rc;
        h->greens_valid = !sum_only; h->gsum_only = sum_only;
    }
    // the cached overlap above needs det R itself; walker.detR / walker.ot / log_detR take the shifted one
    if (h->log_shift_on && (rc = k_log_shift_reortho(h))) return rc;
    return copy_out(h, detR_out, h->detR, sizeof(double) * h->nw);
}

// per-determinant energies E[G_d] from the half-rotated operands of determinant d (algebraically the
// full-G energy of estimators/generic.py:398-434 evaluated on G_d = conj(psi_d) Ghalf_d)
static int local_energy_dets(afq_handle *h) {
    cplx *final_e = h->energy;
    int rc = AFQ_OK;
    for (int d = 0; d < h->ndet && !rc; ++d) {
        select_det(h, d);
        h->energy = h->energy_all + (size_t)d * 3 * h->nw;
        rc = k_energy_generic(h);
    }
    h->energy = final_e;
    select_det(h, 0);
    return rc;
}

static int local_energy(afq_handle *h) {
    if (h->kind == AFQ_SYS_GENERIC && h->ndet > 1) {
        int rc = local_energy_dets(h);
        if (rc) return rc;
        return k_msd_energy_combine(h);              // estimators/mixed.py:439-448
    }
    if (h->kind == AFQ_SYS_GENERIC) return k_energy_generic(h);
    if (h->kind == AFQ_SYS_HUBBARD) return k_energy_hubbard(h);
    if (!h->G) AFQ_FAIL(h, AFQ_ESTATE, "afq_greens must run before afq_local_energy");
    return k_energy_ueg(h);
}

int afq_local_energy(afq_handle *h, double *E_out) {
    AFQ_API(h, "afq_local_energy");
    if (!h) return AFQ_EINVAL;
    int rc = need_ready(h, false);
    if (rc) return rc;
    if (!h->rH1 && h->kind != AFQ_SYS_UEG) AFQ_FAIL(h, AFQ_ESTATE, "half-rotated H1 missing (set trial)");
    if ((rc = ensure_spin_ghalf(h))) return rc;
    { PhaseTimer t(h, T_ENERGY); if ((rc = local_energy(h))) return rc; }
    return copy_out(h, E_out, h->energy, sizeof(cplx) * 3 * h->nw);
}

int afq_set_exchange_algorithm(afq_handle *h, int mode) {
    if (!h || mode < 0 || mode > 2) return AFQ_EINVAL;
    h->exx_mode = mode;
    return AFQ_OK;
}

int afq_set_msd_force_bias(afq_handle *h, int mode) {
    if (!h || mode < 0 || mode > 2) return AFQ_EINVAL;
    h->msd_fb_mode = mode;
    return AFQ_OK;
}

int afq_msd_force_bias(afq_handle *h, int *mode) {
    if (!h || !mode) return AFQ_EINVAL;
    if (h->kind != AFQ_SYS_GENERIC || !h->have_trial || !h->nw) AFQ_FAIL(h, AFQ_ESTATE, "generic system, trial and walkers must be set");
    *mode = h->ndet <= 1 ? 0 : k_msd_gbar_wanted(h) ? 2 : 1;
    return AFQ_OK;
}

int afq_exchange_algorithm(afq_handle *h, int *mode) {
    if (!h || !mode) return AFQ_EINVAL;
    if (h->kind != AFQ_SYS_GENERIC || !h->have_trial) AFQ_FAIL(h, AFQ_ESTATE, "generic system and trial must be set");
    *mode = k_exchange_uses_quadratic(h) ? 2 : 1;
    return AFQ_OK;
}

// ---------------------------------------------------------------- test hooks
int afq_force_bias(afq_handle *h, double *xbar_out) {
    if (!h) return AFQ_EINVAL;
    int rc = need_ready(h, true);
    if (rc) return rc;
    if ((rc = force_bias(h))) return rc;
    return copy_out(h, xbar_out, h->xbar, sizeof(cplx) * (size_t)h->nw * h->K);
}

int afq_shift_fields(afq_handle *h, const double *xi, const double *xbar, double *xs_out, double *cmf_out,
                     double *cfb_out) {
    if (!h || !xi || !xbar) return AFQ_EINVAL;
    int rc = need_ready(h, true);
    if (rc) return rc;
    const size_t n = (size_t)h->nw * h->K;
    AFQ_HIP(h, hipMemcpyAsync(h->xi, xi, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    AFQ_HIP(h, hipMemcpyAsync(h->xbar, xbar, sizeof(cplx) * n, hipMemcpyHostToDevice, h->stream));
    if ((rc = k_fields_explicit(h, h->xi, h->xbar, h->xs, h->cmf, h->cfb))) return rc;
    if ((rc = copy_out(h, xs_out, h->xs, sizeof(cplx) * n))) return rc;
    if ((rc = copy_out(h, cmf_out, h->cmf, sizeof(cplx) * h->nw))) return rc;
    return copy_out(h, cfb_out, h->cfb, sizeof(cplx) * h->nw);
}

int afq_vhs_count(afq_handle *h, int *nv) {
    if (!h || !nv) return AFQ_EINVAL;
    *nv = h->nv;
    return AFQ_OK;
}

// Dense M x M matrices out, whatever the internal storage
int afq_vhs(afq_handle *h, const double *xs, double *vhs_out) {
    if (!h || !xs || !vhs_out) return AFQ_EINVAL;
    int rc = need_ready(h, true);
    if (rc) return rc;
    const size_t n = (size_t)h->nw * h->K;
    AFQ_HIP(h, hipMemcpyAsync(h->xs, xs, sizeof(cplx) * n, hipMemcpyHostToDevice, h->stream));
    if ((rc = build_vhs(h))) return rc;
    const size_t mm = (size_t)h->M * h->M;
    if (!h->vhs_diag) return copy_out(h, vhs_out, h->vhs, sizeof(cplx) * mm * h->nv * h->nw);
    std::vector<double> d((size_t)2 * h->nw * h->nv * h->M);
    if ((rc = copy_out(h, d.data(), h->vhs, sizeof(cplx) * (size_t)h->nw * h->nv * h->M))) return rc;
    std::memset(vhs_out, 0, sizeof(cplx) * mm * h->nv * h->nw);
    for (size_t b = 0; b < (size_t)h->nw * h->nv; ++b)
        for (int p = 0; p < h->M; ++p) {
            vhs_out[2 * (b * mm + (size_t)p * h->M + p)] = d[2 * (b * h->M + p)];
            vhs_out[2 * (b * mm + (size_t)p * h->M + p) + 1] = d[2 * (b * h->M + p) + 1];
        }
    return AFQ_OK;
}

int afq_apply_exponential(afq_handle *h, const double *vhs) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h || !vhs) return AFQ_EINVAL;
    int rc = need_ready(h, true);
    if (rc) return rc;
    if ((rc = k_alive(h))) return rc;
    const size_t mm = (size_t)h->M * h->M;
    if (!h->vhs_diag) {
        AFQ_HIP(h, hipMemcpyAsync(h->vhs, vhs, sizeof(cplx) * mm * h->nv * h->nw, hipMemcpyHostToDevice, h->stream));
    } else {
        std::vector<double> d((size_t)2 * h->nw * h->nv * h->M);
        for (size_t b = 0; b < (size_t)h->nw * h->nv; ++b)
            for (int p = 0; p < h->M; ++p) {
                d[2 * (b * h->M + p)] = vhs[2 * (b * mm + (size_t)p * h->M + p)];
                d[2 * (b * h->M + p) + 1] = vhs[2 * (b * mm + (size_t)p * h->M + p) + 1];
            }
        AFQ_HIP(h, hipMemcpy(h->vhs, d.data(), sizeof(double) * d.size(), hipMemcpyHostToDevice));
    }
    if ((rc = apply_exp(h, h->vhs))) return rc;
    return afq_sync(h);
}

int afq_kinetic(afq_handle *h) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h) return AFQ_EINVAL;
    int rc = need_ready(h, true);
    if (rc) return rc;
    if ((rc = k_alive(h))) return rc;
    if ((rc = k_onebody(h))) return rc;
    return afq_sync(h);
}

// -------------------------------------------------------------- driver glue
int afq_set_weight_cap(afq_handle *h, double frac, double total_weight) {
    if (!h) return AFQ_EINVAL;
    h->cap_frac = frac > 0.0 ? frac : 0.0;
    h->cap_total = total_weight;
    return AFQ_OK;
}

int afq_cap_weights(afq_handle *h, double frac, double total_weight) {
    if (!h) return AFQ_EINVAL;
    int rc = need_ready(h, false);
    if (rc) return rc;
    return k_cap_weights(h, frac, total_weight);
}

int afq_popcontrol_comb(afq_handle *h, double r, double target_weight, int32_t *parent_ix,
                        double *total_weight_out) {
    if (h) h->scal_cache_valid = false;
    AFQ_API(h, "afq_popcontrol_comb");
    if (!h) return AFQ_EINVAL;
    int rc = need_ready(h, false);
    if (rc) { h->greens_valid = false; h->gsum_only = false; return rc; }
    const int nranks = k_comm_size(h);
    if (h->nw * nranks == 1) return AFQ_OK;              // handler.py:226-227
    // a kept Green's function travels with the cloned walkers (the clone / pack kernels copy Ghalf and the cached overlap)
    const bool keep = h->greens_valid && h->ndet == 1;
    h->greens_valid = false; h->gsum_only = false;
    if ((rc = k_comb(h, r, target_weight, keep))) return rc;
    h->greens_valid = keep; h->gsum_only = false;
    if (!parent_ix && !total_weight_out) return AFQ_OK;  // asynchronous: nothing read back, no host sync
    double sc[8];
    if ((rc = copy_out(h, sc, h->scal, sizeof(sc)))) return rc;
    if (total_weight_out) *total_weight_out = sc[0];
    if (sc[1] < 0) AFQ_FAIL(h, AFQ_EWEIGHT, "total walker weight below 1e-8");
    if (sc[3] != 0.0) AFQ_FAIL(h, AFQ_EOVERFLOW, "more walkers moved between two ranks than the exchange slots hold");
    if (sc[6] != 0.0) AFQ_FAIL(h, AFQ_ECOMM, "communicator: a peer rank never signalled (the wait budget of afq_comm_set_timeout ran out on the device)");
    if (!parent_ix) return AFQ_OK;
    if (h->comm) return afq_comm_parent_ix(h, parent_ix);   // the global comb, [nranks * nw]
    return copy_out(h, parent_ix, h->parent_ix, sizeof(int) * h->nw);
}

int afq_walkers_scale_weights(afq_handle *h, double scale) {
    if (!h || scale == 0.0) return AFQ_EINVAL;
    int rc = need_ready(h, false);
    if (rc) return rc;
    return k_scale_weights(h, scale);
}

int afq_walkers_reset_weights(afq_handle *h) {
    if (!h) return AFQ_EINVAL;
    int rc = need_ready(h, false);
    if (rc) return rc;
    return k_reset_weights(h);
}

// packed walker: phi | ot, ehyb, phase, eloc (c128) | unscaled_weight, detR, weight, log_detR (f64)
int afq_walker_pack_bytes(afq_handle *h, int64_t *bytes) {
    if (!h || !bytes) return AFQ_EINVAL;
    size_t b = sizeof(cplx) * ((size_t)h->M * h->nt + 4) + sizeof(double) * 4;
    if (h->nbp > 0)   // phi_old + field history + FieldConfig.step + the two running weight factors
        b += sizeof(cplx) * ((size_t)h->M * h->nt + (size_t)h->nbp * h->K + 1) + sizeof(double) * 2;
    if (h->rdm_on && h->G) b += sizeof(cplx) * (size_t)2 * h->M * h->M;     // walker.G (mixed one_rdm)
    *bytes = (int64_t)b;
    return AFQ_OK;
}

static int pack_io(afq_handle *h, int iw, char *buf, bool pack) {
    const size_t per = (size_t)h->M * h->nt;
    struct Item { void *p; size_t b; };
    const Item items[] = {{h->phi + per * iw, per * sizeof(cplx)}, {h->ot + iw, sizeof(cplx)},
                          {h->ehyb + iw, sizeof(cplx)}, {h->phase + iw, sizeof(cplx)},
                          {h->eloc + iw, sizeof(cplx)}, {h->unscaled + iw, sizeof(double)},
                          {h->detR + iw, sizeof(double)}, {h->weight + iw, sizeof(double)},
                          {h->log_detR + iw, sizeof(double)}};
    size_t off = 0;
    for (const Item &it : items) {
        if (pack) AFQ_HIP(h, hipMemcpyAsync(buf + off, it.p, it.b, hipMemcpyDeviceToDevice, h->stream));
        else AFQ_HIP(h, hipMemcpyAsync(it.p, buf + off, it.b, hipMemcpyDeviceToDevice, h->stream));
        off += it.b;
    }
    if (h->nbp > 0) {
        const size_t hk = (size_t)h->nbp * h->K;
        const Item bp[] = {{h->phi_old + per * iw, per * sizeof(cplx)}, {h->bp_hist + hk * iw, hk * sizeof(cplx)},
                           {h->bp_ph + iw, sizeof(cplx)}, {h->bp_cos + iw, sizeof(double)},
                           {h->bp_n + iw, sizeof(int)}};
        for (const Item &it : bp) {
            if (pack) AFQ_HIP(h, hipMemcpyAsync(buf + off, it.p, it.b, hipMemcpyDeviceToDevice, h->stream));
            else AFQ_HIP(h, hipMemcpyAsync(it.p, buf + off, it.b, hipMemcpyDeviceToDevice, h->stream));
            off += it.b;
        }
    }
    if (h->rdm_on && h->G) {
        const size_t gb = sizeof(cplx) * (size_t)2 * h->M * h->M;
        cplx *g = h->G + (size_t)iw * 2 * h->M * h->M;
        if (pack) AFQ_HIP(h, hipMemcpyAsync(buf + off, g, gb, hipMemcpyDeviceToDevice, h->stream));
        else AFQ_HIP(h, hipMemcpyAsync(g, buf + off, gb, hipMemcpyDeviceToDevice, h->stream));
        off += gb;
    }
    return AFQ_OK;
}

int afq_walker_pack(afq_handle *h, int iw, void *dev_buf) {
    if (!h || !dev_buf) return AFQ_EINVAL;
    int rc = need_ready(h, false);
    if (rc) return rc;
    if (iw < 0 || iw >= h->nw) AFQ_FAIL(h, AFQ_EINVAL, "walker index out of range");
    return pack_io(h, iw, (char *)dev_buf, true);
}

int afq_walker_unpack(afq_handle *h, int iw, const void *dev_buf) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h || !dev_buf) return AFQ_EINVAL;
    int rc = need_ready(h, false);
    if (rc) return rc;
    if (iw < 0 || iw >= h->nw) AFQ_FAIL(h, AFQ_EINVAL, "walker index out of range");
    return pack_io(h, iw, (char *)dev_buf, false);
}

int afq_walkers_copy(afq_handle *h, int src, int dst) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h) return AFQ_EINVAL;
    int rc = need_ready(h, false);
    if (rc) return rc;
    if (src < 0 || dst < 0 || src >= h->nw || dst >= h->nw) AFQ_FAIL(h, AFQ_EINVAL, "walker index out of range");
    if (src == dst) return AFQ_OK;
    const size_t per = (size_t)h->M * h->nt;
#define C_(ptr, n) AFQ_HIP(h, hipMemcpyAsync((ptr) + (size_t)dst * (n), (ptr) + (size_t)src * (n), sizeof(*(ptr)) * (n), hipMemcpyDeviceToDevice, h->stream));
    C_(h->phi, per) C_(h->ot, 1) C_(h->ehyb, 1) C_(h->phase, 1) C_(h->eloc, 1)
    C_(h->unscaled, 1) C_(h->detR, 1) C_(h->weight, 1) C_(h->log_detR, 1)
    if (h->nbp > 0) {
        C_(h->phi_old, per) C_(h->bp_hist, (size_t)h->nbp * h->K) C_(h->bp_ph, 1) C_(h->bp_cos, 1) C_(h->bp_n, 1)
    }
    if (h->rdm_on && h->G) { C_(h->G, (size_t)2 * h->M * h->M) }      // walker.G is walker state for the mixed one_rdm
#undef C_
    return AFQ_OK;
}

static int est_publish_args(afq_handle *h, int zero, EstPublish *pub);

// (publish_zero < 0: plain update; 0 / 1: the block's sums are handed to the host by the same summation launch, see
//  afq_estimates_update_publish)
static int estimates_update_impl(afq_handle *h, int eval_energy, int publish_zero) {
    int rc = need_ready(h, false);
    if (rc) return rc;
    if (publish_zero >= 0 && h->est_pending) AFQ_FAIL(h, AFQ_ESTATE, "afq_estimates_update_publish: a fetch is already in flight");
    if (eval_energy) {
        {
            PhaseTimer t(h, T_GREENS);
            if (!h->greens_valid) {
                // (e.g. behind a comb that did not carry the Green's functions along: multi-determinant trials.)  It is
                // the Green's function of the CURRENT walkers, so it also is what the next step starts from: the overlaps go
                // where the end-of-step evaluation leaves them and the evaluation is kept -- with it the Coulomb vectors the
                // energy contracts below, which are the next step's force bias
                if ((rc = greens_any(h, h->ovlp_new, true))) return rc;
                if (h->greens_cache && !(h->flags & AFQ_PROP_FREE_PROJECTION) && h->psi_stride == 0) { h->greens_valid = true; h->gsum_only = false; }
            }
            if (h->kind == AFQ_SYS_UEG || h->rdm_on) {       // (one_rdm: w.greens_function(trial) refreshes walker.G, mixed.py:212)
                if ((rc = ensure_G(h))) return rc;
                if ((rc = k_full_G(h))) return rc;
            }
        }
        PhaseTimer t(h, T_ENERGY);
        if ((rc = local_energy(h))) return rc;
    }
    if (publish_zero >= 0) {
        // (the arguments are made HERE, behind the energy evaluation: the closed-shell word that rides along is the one of
        //  the newest checking launch)
        EstPublish pub;
        if ((rc = est_publish_args(h, publish_zero, &pub))) return rc;
        if ((rc = k_estimates(h, eval_energy, false, &pub))) return rc;
        h->est_pending = true;
    } else if ((rc = k_estimates(h, eval_energy))) return rc;
    if (h->rdm_on && !(h->flags & AFQ_PROP_FREE_PROJECTION)) {   // the free-projection branch has no RDM (mixed.py:151-175)
        if (!h->G) AFQ_FAIL(h, AFQ_ESTATE, "one_rdm: no Green's function evaluated yet");
        return k_rdm_accumulate(h);
    }
    return AFQ_OK;
}

int afq_estimates_update(afq_handle *h, int eval_energy) {
    AFQ_API(h, "afq_estimates_update");
    if (!h) return AFQ_EINVAL;
    return estimates_update_impl(h, eval_energy, -1);
}

int afq_estimates_update_publish(afq_handle *h, int eval_energy, int zero) {
    AFQ_API(h, "afq_estimates_update_publish");
    if (!h) return AFQ_EINVAL;
    return estimates_update_impl(h, eval_energy, zero ? 1 : 0);
}

int afq_estimates_rdm(afq_handle *h, int on) {
    if (!h) return AFQ_EINVAL;
    int rc = need_ready(h, false);
    if (rc) return rc;
    if (h->ndet > 1 || h->hirsch) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "mixed one_rdm: single-determinant trial, continuous propagator");
    h->rdm_on = on != 0;
    if (h->rdm_on && !h->rdm_acc) {
        const size_t n = (size_t)2 * h->M * h->M;
        if ((rc = dev_alloc(h, &h->rdm_acc, n))) return rc;
        AFQ_HIP(h, hipMemsetAsync(h->rdm_acc, 0, sizeof(double) * n, h->stream));
    }
    return AFQ_OK;
}

int afq_estimates_rdm_get(afq_handle *h, double *rdm_out, int zero) {
    if (!h || !rdm_out) return AFQ_EINVAL;
    if (!h->rdm_acc) AFQ_FAIL(h, AFQ_ESTATE, "mixed one_rdm accumulation is not switched on");
    hipSetDevice(h->device);
    const size_t n = (size_t)2 * h->M * h->M;
    int rc = copy_out(h, rdm_out, h->rdm_acc, sizeof(double) * n);
    if (rc) return rc;
    if (zero) AFQ_HIP(h, hipMemsetAsync(h->rdm_acc, 0, sizeof(double) * n, h->stream));
    return AFQ_OK;
}

int afq_estimates_get(afq_handle *h, double *est_out, int zero) {
    if (!h || !est_out) return AFQ_EINVAL;
    const int rc = afq_estimates_get_begin(h, zero);
    return rc ? rc : afq_estimates_get_end(h, est_out);
}

// The sums of a block go to the host through ONE small kernel that writes them into mapped, coherent host memory, zeroes
// them and then publishes a sequence number there; afq_estimates_get_end polls that number.  The copy / copy / event /
// memset sequence this replaces cost ~25 us of idle device per block (rocprofv3 kernel trace: 6 us ahead of the fill
// kernel, 17 us behind it) although more work was already queued.
__global__ void est_publish_kernel(cplx *est, const double *scal, double *host_out, unsigned long long *host_seq,
                                   unsigned long long seq, int nest, int zero, const unsigned long long *closed_bad) {
    const int t = threadIdx.x;
    if (t < nest) host_out[t] = ((const double *)est)[t];
    if (t < AFQ_NSCAL) host_out[nest + t] = scal[t];
    // the closed-shell word rides along (afq_internal.h: closed_bad): a HINT for the host's choice between the one-spin-first
    // and the one-launch two-spin form of the exchange energy -- never a decision about results
    if (t == 0) host_seq[1] = closed_bad ? *closed_bad : 0ull;
    __threadfence_system();
    __syncthreads();
    if (t < nest && zero) ((double *)est)[t] = 0.0;
    if (t == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int afq_estimates_get_begin(afq_handle *h, int zero) {
    AFQ_API(h, "afq_estimates_get");
    if (!h) return AFQ_EINVAL;
    hipSetDevice(h->device);
    if (h->est_pending) AFQ_FAIL(h, AFQ_ESTATE, "afq_estimates_get_begin: a fetch is already in flight");
    EstPublish pub;
    { const int rc = est_publish_args(h, zero, &pub); if (rc) return rc; }
    if (h->est_acc_pending) {
        // sums still sitting in the per-walker accumulators: the launch that folds them in hands the block over too
        const int rc = k_estimates(h, 0, true, &pub);
        if (rc) return rc;
    } else {
        AFQ_LAUNCH(h, est_publish_kernel, dim3(1), dim3(64), 0, h->stream, h->estimates, h->scal, pub.host_out,
                   pub.host_seq, pub.seq, pub.nest, pub.zero, pub.closed_bad);
        AFQ_POST(h);
    }
    h->est_pending = true;
    return AFQ_OK;
}

// the mapped staging area, the next sequence number and the closed-shell epoch of one hand-over
static int est_publish_args(afq_handle *h, int zero, EstPublish *pub) {
    const size_t nest = 2 * (size_t)AFQ_EST_COUNT_;
    if (!h->est_stage) {
        // [nest sums | scal[AFQ_NSCAL] | sequence number], written by the device, polled by the host
        AFQ_HIP(h, hipHostMalloc((void **)&h->est_stage, sizeof(double) * (nest + AFQ_NSCAL + 2),
                                 hipHostMallocMapped | hipHostMallocCoherent));
        memset(h->est_stage, 0, sizeof(double) * (nest + AFQ_NSCAL + 2));
    }
    // the one host synchronisation of a block of steps also reports a population that collapsed in an
    // asynchronous comb (scal[2], set by comb_plan_kernel; walkers/handler.py:236-241 exits there)
    double *dev_view = nullptr;
    AFQ_HIP(h, hipHostGetDevicePointer((void **)&dev_view, h->est_stage, 0));
    ++h->est_seq;
    h->closed_epoch_pub = h->closed_epoch;                   // the newest launch whose verdict the published word can hold
    pub->host_out = dev_view; pub->host_seq = (unsigned long long *)(dev_view + nest + AFQ_NSCAL); pub->seq = h->est_seq;
    pub->scal = h->scal; pub->closed_bad = h->closed_bad; pub->nest = (int)nest; pub->zero = zero;
    return AFQ_OK;
}

int afq_estimates_get_end(afq_handle *h, double *est_out) {
    AFQ_API(h, "afq_estimates_get");
    if (!h || !est_out) return AFQ_EINVAL;
    if (!h->est_pending) AFQ_FAIL(h, AFQ_ESTATE, "afq_estimates_get_end without afq_estimates_get_begin");
    hipSetDevice(h->device);
    h->est_pending = false;
    const size_t nest = 2 * (size_t)AFQ_EST_COUNT_;
    {
        // poll the sequence number; the stream query catches a failed launch or device (no endless wait)
        const unsigned long long *seq = (const unsigned long long *)(h->est_stage + nest + AFQ_NSCAL);
        unsigned spins = 0;
        while (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != h->est_seq) {
            if ((++spins & 0x3ffu) == 0) {
                const hipError_t q = hipStreamQuery(h->stream);
                if (q == hipSuccess) break;                       // everything enqueued has run
                if (q != hipErrorNotReady) AFQ_HIP(h, q);
            }
        }
        if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != h->est_seq)
            AFQ_FAIL(h, AFQ_EHIP, "estimator sums were not published by the device");
    }
    memcpy(est_out, h->est_stage, sizeof(double) * nest);
    {   // at the time of the publish, had the newest checking Green's function launch found an open-shell walker?
        const unsigned long long bad = ((const unsigned long long *)(h->est_stage + nest + AFQ_NSCAL))[1];
        h->exx_open_hint = h->closed_epoch_pub != 0 && bad >= h->closed_epoch_pub;
    }
    const double *sc = h->est_stage + nest;
    // the population-control scalars of the block ride along: afq_comm_stats right behind this call needs no synchronisation
    memcpy(h->scal_cache, sc, sizeof(double) * AFQ_NSCAL);
    h->scal_cache_valid = true;
    if (sc[2] != 0.0) AFQ_FAIL(h, AFQ_EWEIGHT, "total walker weight below 1e-8 in an earlier population control");
    if (sc[3] != 0.0) AFQ_FAIL(h, AFQ_EOVERFLOW, "population control: more walkers moved between two ranks than the "
                                                 "exchange slots hold (afq_comm_init capacity)");
    if (sc[6] != 0.0) AFQ_FAIL(h, AFQ_ECOMM, "communicator: a peer rank never signalled an all-gather row / walker slots "
                                             "(the wait budget of afq_comm_set_timeout ran out on the device)");
    return AFQ_OK;
}

// ---------------------------------------------------------------------- misc
int afq_rng_seed(afq_handle *h, uint64_t seed, uint64_t stream) {
    if (!h) return AFQ_EINVAL;
    h->rng_seed = seed; h->rng_stream = stream; h->rng_counter = 0;
    return AFQ_OK;
}

int afq_rng_normal(afq_handle *h, double *out, int64_t n) {
    if (!h || !out || n < 1) return AFQ_EINVAL;
    hipSetDevice(h->device);
    double *tmp = nullptr;
    int rc = dev_alloc(h, &tmp, (size_t)n);
    if (rc) return rc;
    rc = k_rng_normal_into(h, tmp, (long)n);
    if (!rc) rc = copy_out(h, out, tmp, sizeof(double) * (size_t)n);
    dev_free(tmp);
    return rc;
}

int afq_rng_philox4x32(afq_handle *h, const uint32_t *ctr_key, uint32_t *out, int n) {
    if (!h || !ctr_key || !out || n < 1) return AFQ_EINVAL;
    hipSetDevice(h->device);
    uint32_t *in_d = nullptr, *out_d = nullptr;
    int rc = dev_upload(h, &in_d, ctr_key, (size_t)6 * n);
    if (!rc) rc = dev_alloc(h, &out_d, (size_t)4 * n);
    if (!rc) rc = k_philox_raw(h, in_d, out_d, n);
    if (!rc) rc = copy_out(h, out, out_d, sizeof(uint32_t) * 4 * (size_t)n);
    dev_free(in_d); dev_free(out_d);
    return rc;
}

int afq_debug(afq_handle *h, int sync_every_launch, int markers) {
    if (!h) return AFQ_EINVAL;
    h->debug_sync = sync_every_launch != 0;
    h->debug_markers = markers != 0;
    return AFQ_OK;
}

// Readable from a watchdog thread while the owning thread is blocked inside a synchronising call: touches
// host memory only (no HIP call, no lock).
int afq_last_launch(afq_handle *h, char *buf, int len, uint64_t *queued, uint64_t *retired) {
    if (!h || !buf || len < 1) return AFQ_EINVAL;
    const unsigned long long n = h->n_launch, r = h->retired ? *h->retired : 0ull;
    if (queued) *queued = n;
    if (retired) *retired = h->debug_markers ? r : 0ull;
    std::string txt = std::string("api=") + (h->crumb_api ? h->crumb_api : "") + " queued=" + std::to_string(n);
    if (h->debug_markers) {
        txt += " retired=" + std::to_string(r);
        if (r < n && h->crumb_name[r & 63] && n - r <= 64) txt += std::string(" first-unretired=") + h->crumb_name[r & 63];
    }
    txt += " last:";
    for (unsigned long long i = n > 8 ? n - 8 : 0; i < n; ++i)
        txt += std::string(" ") + (h->crumb_name[i & 63] ? h->crumb_name[i & 63] : "?");
    snprintf(buf, (size_t)len, "%s", txt.c_str());
    return AFQ_OK;
}

int afq_counters_ext(afq_handle *h, int64_t *out, int n, int reset) {
    if (!h || !out || n < 0) return AFQ_EINVAL;
    hipSetDevice(h->device);
    unsigned long long c[AFQ_NCOUNTERS];
    int rc = copy_out(h, c, h->counters, sizeof(c));
    if (rc) return rc;
    for (int i = 0; i < n; ++i) out[i] = i < AFQ_NCOUNTERS ? (int64_t)c[i] : 0;
    if (reset) AFQ_HIP(h, hipMemsetAsync(h->counters, 0, sizeof(c), h->stream));
    return AFQ_OK;
}

int afq_counters(afq_handle *h, int64_t *out, int reset) { return afq_counters_ext(h, out, 4, reset); }

int afq_enable_timers(afq_handle *h, int on) {
    if (!h) return AFQ_EINVAL;
    h->timers_on = on != 0;
    return AFQ_OK;
}

int afq_timers(afq_handle *h, double *out_ms, int reset) {
    if (!h || !out_ms) return AFQ_EINVAL;
    for (int i = 0; i < T_COUNT; ++i) { out_ms[i] = h->t_ms[i]; if (reset) h->t_ms[i] = 0.0; }
    return AFQ_OK;
}

// ---------------------------------------------------------------- discrete Hirsch propagator
static int hirsch_buffers(afq_handle *h) {
    if (h->hs_oinv) return AFQ_OK;
    const size_t n = h->nw, nmax = h->na > h->nb ? h->na : h->nb;
    int rc;
    if ((rc = dev_alloc(h, &h->hs_oinv, n * 2 * nmax * nmax))) return rc;
    if ((rc = dev_alloc(h, &h->hs_u, n * h->M))) return rc;
    if ((rc = dev_alloc(h, &h->hs_fields, n * h->M))) return rc;
    if ((rc = dev_alloc(h, &h->hs_used, n))) return rc;
    if ((rc = dev_alloc(h, &h->hs_alive0, n))) return rc;
    AFQ_HIP(h, hipMemsetAsync(h->hs_alive0, 0, sizeof(int) * n, h->stream));
    return AFQ_OK;
}

static int hirsch_ready(afq_handle *h) {
    int rc = need_ready(h, true);
    if (rc) return rc;
    if (!h->hirsch) AFQ_FAIL(h, AFQ_ESTATE, "the discrete Hirsch propagator is not set");
    return hirsch_buffers(h);
}

int afq_set_propagator_hirsch(afq_handle *h, const double *bt2, double dt, int charge_decomposition) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h || !bt2 || dt <= 0) return AFQ_EINVAL;
    if (h->kind != AFQ_SYS_HUBBARD) AFQ_FAIL(h, AFQ_ESTATE, "the Hirsch transformation needs a Hubbard system");
    if (h->ndet > 1) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "Hirsch propagator: single-determinant trials");
    hipSetDevice(h->device);
    int rc;
    if ((rc = dev_upload(h, &h->BH1, bt2, (size_t)2 * h->M * h->M))) return rc;
    h->bh1_real = true;         // (expm of a real hopping matrix: the one-body GEMM then needs two real products per pair)
    for (size_t i = 0, n = (size_t)2 * h->M * h->M; i < n && h->bh1_real; ++i) h->bh1_real = bt2[2 * i + 1] == 0.0;
    h->bh1_same = false;
    // propagation/hubbard.py:66-82
    typedef std::complex<double> C;
    C gamma, auxf[2][2], wfac[2];
    const double e = std::exp(-0.5 * dt * h->U);
    if (charge_decomposition) {
        gamma = std::acosh(C(e, 0.0));
        auxf[0][0] = auxf[0][1] = std::exp(gamma);
        auxf[1][0] = auxf[1][1] = std::exp(-gamma);
        wfac[0] = std::exp(0.5 * dt * h->U) * std::exp(-gamma);
        wfac[1] = std::exp(0.5 * dt * h->U) * std::exp(gamma);
    } else {
        gamma = C(std::acosh(std::exp(0.5 * dt * h->U)), 0.0);
        auxf[0][0] = std::exp(gamma); auxf[0][1] = std::exp(-gamma);
        auxf[1][0] = std::exp(-gamma); auxf[1][1] = std::exp(gamma);
        wfac[0] = wfac[1] = C(1.0, 0.0);
    }
    for (int x = 0; x < 2; ++x) {
        h->hs_wfac[x] = cmake(wfac[x].real(), wfac[x].imag());
        for (int sp = 0; sp < 2; ++sp) {
            const C f = auxf[x][sp] * e, d = f - 1.0;
            h->hs_auxf[x][sp] = cmake(f.real(), f.imag());
            h->hs_delta[x][sp] = cmake(d.real(), d.imag());
        }
    }
    h->hs_gamma = cmake(gamma.real(), gamma.imag());
    h->hs_charge = charge_decomposition != 0;
    h->hs_direct = false;
    if (!h->mf_shift) {
        std::vector<double> z(2 * (size_t)h->K, 0.0);
        if ((rc = dev_upload(h, &h->mf_shift, z.data(), (size_t)h->K))) return rc;
    }
    h->dt = dt; h->sqrt_dt = std::sqrt(dt);
    h->flags = 0; h->nv = 1; h->vhs_diag = true;
    h->hirsch = true; h->have_prop = true;
    return AFQ_OK;
}

int afq_hirsch_kinetic(afq_handle *h) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h) return AFQ_EINVAL;
    int rc = hirsch_ready(h);
    if (rc) return rc;
    if ((rc = k_hirsch_alive(h, 0))) return rc;                 // walkers the driver propagates (|w| > 1e-8)
    return k_hirsch_kinetic(h);
}

int afq_hirsch_two_body(afq_handle *h, const double *u, int32_t *fields_out, int32_t *used_out) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h || !u) return AFQ_EINVAL;
    int rc = hirsch_ready(h);
    if (rc) return rc;
    AFQ_HIP(h, hipMemcpyAsync(h->hs_u, u, sizeof(double) * (size_t)h->nw * h->M, hipMemcpyHostToDevice, h->stream));
    if ((rc = k_hirsch_alive(h, 1))) return rc;                 // if abs(walker.weight) > 0 (hubbard.py:308)
    if ((rc = k_hirsch_two_body(h))) return rc;
    if (fields_out && (rc = copy_out(h, fields_out, h->hs_fields, sizeof(int) * (size_t)h->nw * h->M))) return rc;
    if (used_out && (rc = copy_out(h, used_out, h->hs_used, sizeof(int) * (size_t)h->nw))) return rc;
    return AFQ_OK;
}

int afq_hirsch_finish(afq_handle *h, double eshift) {
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h) return AFQ_EINVAL;
    int rc = hirsch_ready(h);
    if (rc) return rc;
    if ((rc = k_hirsch_alive(h, 1))) return rc;                 // if abs(walker.weight.real) > 0 (:310)
    if ((rc = k_hirsch_kinetic(h))) return rc;
    if ((rc = k_hirsch_eshift(h, std::exp(h->dt * eshift)))) return rc;
    return k_alive(h);
}

int afq_propagate_hirsch(afq_handle *h, double eshift) {
    AFQ_API(h, "afq_propagate_hirsch");
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h) return AFQ_EINVAL;
    int rc = hirsch_ready(h);
    if (rc) return rc;
    if ((rc = k_hirsch_alive(h, 0))) return rc;
    if ((rc = k_hirsch_kinetic(h))) return rc;
    if ((rc = k_rng_uniform(h, h->hs_u, (long)h->nw * h->M))) return rc;
    if ((rc = k_hirsch_alive(h, 1))) return rc;
    if ((rc = k_hirsch_two_body(h))) return rc;
    if ((rc = k_hirsch_alive(h, 1))) return rc;
    if ((rc = k_hirsch_kinetic(h))) return rc;
    if ((rc = k_hirsch_eshift(h, std::exp(h->dt * eshift)))) return rc;
    return k_alive(h);
}

int afq_hirsch_single_site(afq_handle *h, int on) {
    if (!h) return AFQ_EINVAL;
    if (!h->hirsch) AFQ_FAIL(h, AFQ_ESTATE, "afq_set_propagator_hirsch first");
    if (!on && h->nbp > 0) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "the direct Hirsch update records no field history (hubbard.py:222-275)");
    h->hs_direct = !on;
    return AFQ_OK;
}

int afq_hirsch_free_projection(afq_handle *h, int on) {
    if (!h) return AFQ_EINVAL;
    if (!h->hirsch) AFQ_FAIL(h, AFQ_ESTATE, "afq_set_propagator_hirsch first");
    if (on && h->nbp > 0) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "no field history in free projection");
    // the estimators (mixed.py:151-175) and the re-orthogonalisation (walkers/handler.py:176-180) read the flag
    if (on) h->flags |= AFQ_PROP_FREE_PROJECTION; else h->flags &= ~AFQ_PROP_FREE_PROJECTION;
    return AFQ_OK;
}

int afq_propagate_hirsch_free(afq_handle *h, const double *u, int32_t *fields_out, double eshift) {
    AFQ_API(h, "afq_propagate_hirsch_free");
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h) return AFQ_EINVAL;
    int rc = hirsch_ready(h);
    if (rc) return rc;
    if (!(h->flags & AFQ_PROP_FREE_PROJECTION)) AFQ_FAIL(h, AFQ_ESTATE, "afq_hirsch_free_projection(h, 1) first");
    if ((rc = k_hirsch_alive(h, 0))) return rc;                   // the driver's test |weight| > 1e-8 (qmc/afqmc.py:232)
    if (u) AFQ_HIP(h, hipMemcpyAsync(h->hs_u, u, sizeof(double) * (size_t)h->nw * h->M, hipMemcpyHostToDevice, h->stream));
    else if ((rc = k_rng_uniform(h, h->hs_u, (long)h->nw * h->M))) return rc;
    if ((rc = k_hirsch_free(h, eshift))) return rc;
    if (fields_out && (rc = copy_out(h, fields_out, h->hs_fields, sizeof(int) * (size_t)h->nw * h->M))) return rc;
    return k_alive(h);
}

// ---------------------------------------------------------------- back-propagation
int afq_bp_configure(afq_handle *h, int nbp) {
    if (!h || nbp < 1) return AFQ_EINVAL;
    int rc = need_ready(h, true);
    if (rc) return rc;
    if (h->ndet > 1) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "back-propagation needs a single-determinant trial");
    if (h->kind == AFQ_SYS_HUBBARD && !h->hirsch)
        AFQ_FAIL(h, AFQ_EUNSUPPORTED, "back-propagation of a Hubbard system: discrete fields only (the reference's propagation/hubbard.py:568-672 reads the history as 0 / 1 fields)");
    if (h->hirsch && h->K != h->M) AFQ_FAIL(h, AFQ_ESTATE, "discrete fields: one per site expected");
    if (h->flags & AFQ_PROP_FREE_PROJECTION) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "no field history in free projection");
    const size_t per = (size_t)h->M * h->nt, n = h->nw;
    if ((rc = dev_alloc(h, &h->bp_hist, n * nbp * h->K))) return rc;
    if ((rc = dev_alloc(h, &h->bp_n, n))) return rc;
    if ((rc = dev_alloc(h, &h->bp_flag, n))) return rc;
    if ((rc = dev_alloc(h, &h->bp_cos, n))) return rc;
    if ((rc = dev_alloc(h, &h->bp_ph, n))) return rc;
    if ((rc = dev_alloc(h, &h->phi_old, per * n))) return rc;
    if ((rc = dev_alloc(h, &h->phi_bp, 2 * per * n))) return rc;        // phi_bp and conj(phi_bp)
    if ((rc = dev_alloc(h, &h->BH1dag, (size_t)2 * h->M * h->M))) return rc;
    if ((rc = dev_alloc(h, &h->bp_xs, n * h->K))) return rc;
    if ((rc = dev_alloc(h, &h->bp_est, (size_t)4 + 2 * h->M * h->M))) return rc;
    h->nbp = nbp;
    AFQ_HIP(h, hipMemsetAsync(h->bp_hist, 0, sizeof(cplx) * n * nbp * h->K, h->stream));
    AFQ_HIP(h, hipMemsetAsync(h->bp_flag, 0, sizeof(int) * n, h->stream));
    if ((rc = k_bp_reset(h, true))) return rc;
    if ((rc = k_conj_transpose(h, h->BH1, h->BH1dag))) return rc;
    // walkers/walker.py:43: phi_old starts as the walker itself
    AFQ_HIP(h, hipMemcpyAsync(h->phi_old, h->phi, sizeof(cplx) * per * n, hipMemcpyDeviceToDevice, h->stream));
    return AFQ_OK;
}

int afq_bp_steps(afq_handle *h, int32_t *steps_out) {
    if (!h || !steps_out) return AFQ_EINVAL;
    if (!h->nbp) AFQ_FAIL(h, AFQ_ESTATE, "back-propagation is not configured");
    hipSetDevice(h->device);
    int rc = copy_out(h, steps_out, h->bp_n, sizeof(int) * h->nw);
    // discrete fields are recorded one at a time: FieldConfig.step counts completed configurations
    if (!rc && h->hirsch) for (int i = 0; i < h->nw; ++i) steps_out[i] /= h->M;
    return rc;
}

int afq_bp_update(afq_handle *h, const double *phi_bp0, int nstblz, int restore_weights, int eval_energy,
                  int reset, double *est_out) {
    AFQ_API(h, "afq_bp_update");
    if (h) { h->greens_valid = false; h->gsum_only = false; }
    if (!h || !phi_bp0 || !est_out || nstblz < 1 || restore_weights < 0 || restore_weights > 2) return AFQ_EINVAL;
    int rc = need_ready(h, true);
    if (rc) return rc;
    if (!h->nbp) AFQ_FAIL(h, AFQ_ESTATE, "back-propagation is not configured");
    if (h->hirsch && restore_weights)
        AFQ_FAIL(h, AFQ_EUNSUPPORTED, "restore_weights with discrete fields: FieldConfig.push records no weight factors (walkers/stack.py:35-49)");
    const size_t per = (size_t)h->M * h->nt, n = h->nw;
    // trial (or initial) determinant for every walker; the second half of phi_bp is upload scratch first
    AFQ_HIP(h, hipMemcpyAsync(h->phi_bp + per * n, phi_bp0, sizeof(cplx) * per, hipMemcpyHostToDevice, h->stream));
    if ((rc = k_bp_init(h, h->phi_bp + per * n))) return rc;
    // borrow the step machinery: phi <- phi_bp, BH1 <- BH1^H, fields <- -conj(x), every walker "alive"
    // while it still has recorded steps; the walkers' own overlaps / detR / weights are parked
    cplx *s_phi = h->phi, *s_xs = h->xs, *s_BH1 = h->BH1, *s_ot = h->ot;
    double *s_detR = h->detR;
    const int s_flags = h->flags;
    cplx *tmp_ot = nullptr; double *tmp_detR = nullptr;
    if ((rc = dev_alloc(h, &tmp_ot, n))) return rc;
    if ((rc = dev_alloc(h, &tmp_detR, n))) { dev_free(tmp_ot); return rc; }
    h->phi = h->phi_bp; h->xs = h->bp_xs; h->BH1 = h->BH1dag; h->ot = tmp_ot; h->detR = tmp_detR;
    h->flags &= ~AFQ_PROP_FREE_PROJECTION;
    const bool fused = k_prop_fused_supported(h);
    rc = AFQ_OK;
    for (int i = 0; i < h->nbp && !rc; ++i) {                       // propagation/generic.py:279-288
        if (h->hirsch) {                                            // propagation/hubbard.py:661-671
            rc = k_bp_hirsch_step(h, i);
            if (!rc && i != 0 && i % nstblz == 0) rc = k_reortho(h);
            continue;
        }
        if ((rc = k_bp_fields(h, i))) break;
        h->vhs_upper = fused && h->hs_sym && !h->no_vhs_upper;
        rc = build_vhs(h);
        if (!rc && fused) rc = k_prop_fused(h);
        h->vhs_upper = false;
        if (rc) break;
        if (!fused) {
            if ((rc = k_onebody(h))) break;
            if ((rc = apply_exp(h, h->vhs))) break;
            rc = k_onebody(h);
        }
        if (!rc && i != 0 && i % nstblz == 0) rc = k_reortho(h);    // utils/linalg.py:82-105 on both spins
    }
    h->phi = s_phi; h->xs = s_xs; h->BH1 = s_BH1; h->ot = s_ot; h->detR = s_detR; h->flags = s_flags;
    dev_free(tmp_ot); dev_free(tmp_detR);
    if (rc) return rc;
    // G_bp[w] = gab(phi_bp, phi_old)^T (back_propagation.py:156-157) = the Green's function of phi_old with
    // phi_bp[w] in the role of the trial
    if ((rc = k_conj_copy(h, h->phi_bp, h->phi_bp + per * n, (long)(per * n)))) return rc;
    cplx *s_psi = h->psi, *s_psic = h->psic;
    h->phi = h->phi_old; h->psi = h->phi_bp; h->psic = h->phi_bp + per * n; h->psi_stride = (long)per;
    rc = k_greens(h, h->ovlp_old);
    if (!rc) rc = ensure_G(h);
    if (!rc) rc = k_full_G(h);
    h->phi = s_phi; h->psi = s_psi; h->psic = s_psic; h->psi_stride = 0;
    if (rc) return rc;
    AFQ_HIP(h, hipMemsetAsync(h->bp_est, 0, sizeof(cplx) * ((size_t)4 + 2 * h->M * h->M), h->stream));
    if (eval_energy) {
        // local_energy(system, G_bp, opt=False) (back_propagation.py:159-163): the full-G Cholesky energy
        if (h->kind != AFQ_SYS_GENERIC) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "back-propagated energies: generic systems only");
        if ((rc = k_energy_full_g(h, h->G, h->nw, h->energy))) return rc;
    }
    if ((rc = k_bp_accumulate(h, restore_weights, eval_energy))) return rc;
    if (reset) {
        // FieldConfig.reset + Walkers.copy_historic_wfn (walkers/stack.py:124-127, handler.py:200-203)
        if ((rc = k_bp_reset(h, false))) return rc;
        AFQ_HIP(h, hipMemcpyAsync(h->phi_old, h->phi, sizeof(cplx) * per * n, hipMemcpyDeviceToDevice, h->stream));
    }
    if ((rc = k_alive(h))) return rc;
    return copy_out(h, est_out, h->bp_est, sizeof(cplx) * ((size_t)4 + 2 * h->M * h->M));
}

int afq_local_energy_full_g(afq_handle *h, const double *G, int n, double *E_out) {
    if (!h || !G || !E_out || n < 1) return AFQ_EINVAL;
    if (h->kind != AFQ_SYS_GENERIC) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "full-G Cholesky energy: generic systems only");
    hipSetDevice(h->device);
    const size_t gsz = (size_t)2 * h->M * h->M * n;
    cplx *Gd = nullptr, *Ed = nullptr;
    int rc;
    if ((rc = dev_upload(h, &Gd, G, gsz))) return rc;
    if ((rc = dev_alloc(h, &Ed, (size_t)3 * n))) { dev_free(Gd); return rc; }
    rc = k_energy_full_g(h, Gd, n, Ed);
    if (!rc) rc = copy_out(h, E_out, Ed, sizeof(cplx) * 3 * n);
    dev_free(Gd); dev_free(Ed);
    return rc;
}

int afq_walkers_det_weights(afq_handle *h, double *weights_out) {
    if (!h || !weights_out) return AFQ_EINVAL;
    if (!h->nw) AFQ_FAIL(h, AFQ_ESTATE, "no walkers allocated");
    hipSetDevice(h->device);
    if (h->ndet <= 1) AFQ_FAIL(h, AFQ_ESTATE, "single-determinant trial: the overlap is the only weight");
    return copy_out(h, weights_out, h->detw, sizeof(cplx) * (size_t)h->nw * h->ndet);
}

int afq_kernel_trace(afq_handle *h, int on) {
    if (!h) return AFQ_EINVAL;
    // on == 1: every kind; on > 1: bit (k + 1) selects kind k, e.g. 2 = AFQ_K_PROPAGATOR only; 0: off
    h->ktrace_mask = on == 1 ? ~0u : on > 1 ? (unsigned)on >> 1 : 0u;
    if (on) for (int k = 0; k < AFQ_K_COUNT; ++k) { h->ktrace_used[k] = 0; h->ktrace_seen[k] = 0; }
    return AFQ_OK;
}

int afq_kernel_trace_stride(afq_handle *h, int kind, int stride) {
    if (!h || stride < 1 || kind < 0 || kind >= AFQ_K_COUNT) return AFQ_EINVAL;
    h->ktrace_stride[kind] = stride;
    return AFQ_OK;
}

int afq_kernel_trace_get(afq_handle *h, int kind, double *ms_out, int max_n, int *n_out) {
    if (!h || !n_out || kind < 0 || kind >= AFQ_K_COUNT || (max_n > 0 && !ms_out)) return AFQ_EINVAL;
    hipSetDevice(h->device);
    AFQ_HIP(h, hipStreamSynchronize(h->stream));
    const int n = h->ktrace_used[kind] < max_n ? h->ktrace_used[kind] : max_n;
    for (int i = 0; i < n; ++i) {
        float f = 0;
        AFQ_HIP(h, hipEventElapsedTime(&f, h->ktrace_ev[kind][2 * i], h->ktrace_ev[kind][2 * i + 1]));
        ms_out[i] = f;
    }
    *n_out = h->ktrace_used[kind];
    return AFQ_OK;
}

int afq_launch_trace(afq_handle *h, int on) {
    if (!h) return AFQ_EINVAL;
    hipSetDevice(h->device);
    if (h->ltrace_open) afq_launch_trace_mark(h, nullptr);
    if (on) h->ltrace_name.clear();
    h->ltrace_on = on != 0;
    return AFQ_OK;
}

int afq_launch_trace_get(afq_handle *h, char *names_out, int names_len, double *total_ms, int64_t *launches,
                         int max_names, int *n_out) {
    if (!h || !n_out || (max_names > 0 && (!names_out || !total_ms || !launches))) return AFQ_EINVAL;
    hipSetDevice(h->device);
    if (h->ltrace_open) afq_launch_trace_mark(h, nullptr);
    AFQ_HIP(h, hipStreamSynchronize(h->stream));
    std::vector<std::string> names;
    std::vector<double> ms;
    std::vector<int64_t> cnt;
    for (size_t i = 0; i < h->ltrace_name.size(); ++i) {
        float f = 0;
        AFQ_HIP(h, hipEventElapsedTime(&f, h->ltrace_ev[2 * i], h->ltrace_ev[2 * i + 1]));
        size_t k = 0;
        while (k < names.size() && names[k] != h->ltrace_name[i]) ++k;
        if (k == names.size()) { names.push_back(h->ltrace_name[i]); ms.push_back(0.0); cnt.push_back(0); }
        ms[k] += f; cnt[k] += 1;
    }
    *n_out = (int)names.size();
    size_t at = 0;
    for (int k = 0; k < (int)names.size() && k < max_names; ++k) {
        if (at + names[k].size() + 1 > (size_t)names_len) AFQ_FAIL(h, AFQ_EINVAL, "afq_launch_trace_get: names buffer too small");
        memcpy(names_out + at, names[k].c_str(), names[k].size() + 1);
        at += names[k].size() + 1;
        total_ms[k] = ms[k]; launches[k] = cnt[k];
    }
    return AFQ_OK;
}

int afq_propagator_issued_flops(afq_handle *h, double *open_per_walker, double *closed_per_walker) {
    if (!h || !open_per_walker || !closed_per_walker) return AFQ_EINVAL;
    *open_per_walker = h->prop_issued_open;
    *closed_per_walker = h->prop_issued_closed;
    return AFQ_OK;
}

int afq_kernel_issued_flops(afq_handle *h, int kind, double *flops_out) {
    if (!h || !flops_out || kind < 0 || kind >= AFQ_K_COUNT) return AFQ_EINVAL;
    *flops_out = h->issued_flops[kind];
    return AFQ_OK;
}

}  // extern "C"
