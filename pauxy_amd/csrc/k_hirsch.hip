// Discrete Hirsch Hubbard-Stratonovich propagation for the Hubbard model (constrained-path AFQMC with
// single-site updates), propagation/hubbard.py:12-343, SURVEY section 8f-4.
//
// One step of a walker is  kinetic half step + importance sampling -> M sequential single-site updates ->
// kinetic half step + importance sampling -> weight *= exp(dt eshift).  The kinetic parts reuse the one-body
// GEMM and the Green's-function kernel (which hands out O^-1, O = phi^T conj(psi)); the site loop is one
// work-group per walker with both inverse overlaps in LDS: per site two N x N mat-vecs, the overlap ratios of
// the two field values (:549-551), the choice of the field from a uniform number, a row scaling of phi and a
// Sherman-Morrison rank-1 update of the inverse (utils/linalg.py:6-30, walkers/single_det.py:117-139).
#include "afq_internal.h"

struct HirschArgs {
    int M, na, nb, nt, nw, nmax;
    cplx *phi;
    const cplx *psi;
    cplx *oinv;
    double *weight;
    cplx *ot;
    const double *u;                 // [nw, M]
    int *fields, *used;
    const int *alive;
    int inv_in_lds;
    cplx delta[2][2], wfac[2];
    // back-propagation history (FieldConfig.push, walkers/stack.py:35-49): one field at a time, bp_n = fields
    // recorded so far = step * M + ib; null when off
    cplx *bp_hist;
    int *bp_n;
    long hist_cap;                   // nbp * M
};

__global__ __launch_bounds__(256) void hirsch_two_body_kernel(HirschArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ cplx q_s[2][128], a_s[2][128], gii_s[2], den_s[2];
    __shared__ int xi_s, stop_s;
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!a.alive[w]) { if (tid == 0) a.used[w] = 0; return; }
    const int M = a.M, nt = a.nt, nmax = a.nmax;
    cplx *phi = a.phi + (long)w * M * nt;
    cplx *og = a.oinv + (long)w * 2 * nmax * nmax;
    // O^-1 of both spins: in LDS when it fits (N <= 68), else updated in place in global memory (L2 / MALL resident)
    cplx *inv = a.inv_in_lds ? (cplx *)smem : og;               // [2][nmax][nmax]
    if (a.inv_in_lds) for (int e = tid; e < 2 * nmax * nmax; e += 256) inv[e] = og[e];
    double weight = a.weight[w];
    cplx ot = a.ot[w];
    int used = 0;
    long pos = a.bp_hist ? a.bp_n[w] : 0;
    __syncthreads();
    for (int i = 0; i < M; ++i) {
        // q_k = sum_l O^-1[k][l] phi[i,l]: one wave per row k, lanes over l;
        // a_l = sum_k O^-1[k][l] conj(psi[i,k]): one thread per column l, rows read coalesced   (both spins)
        for (int s = 0; s < 2; ++s) {
            const int ns = s == 0 ? a.na : a.nb, off = s == 0 ? 0 : a.na;
            const cplx *iv = inv + (long)s * nmax * nmax;
            for (int k = wave; k < ns; k += 4) {
                cplx acc = cmake(0.0, 0.0);
                for (int l = lane; l < ns; l += 64) cfma(acc, iv[k * nmax + l], phi[(long)i * nt + off + l]);
                for (int o = 32; o > 0; o >>= 1) { acc.x += __shfl_down(acc.x, o); acc.y += __shfl_down(acc.y, o); }
                if (lane == 0) q_s[s][k] = acc;
            }
            for (int l = tid; l < ns; l += 256) {
                cplx acc = cmake(0.0, 0.0);
                for (int k = 0; k < ns; ++k) cfma(acc, iv[k * nmax + l], cconj(a.psi[(long)i * nt + off + k]));
                a_s[s][l] = acc;
            }
        }
        __syncthreads();
        if (wave < 2) {                                       // G_ii of spin `wave` (hubbard.py:110-122)
            const int s = wave, ns = s == 0 ? a.na : a.nb, off = s == 0 ? 0 : a.na;
            cplx g = cmake(0.0, 0.0), d = cmake(0.0, 0.0);
            for (int k = lane; k < ns; k += 64) {
                cfma(g, cconj(a.psi[(long)i * nt + off + k]), q_s[s][k]);
                cfma(d, phi[(long)i * nt + off + k], a_s[s][k]);      // vt . (inv u) of the Sherman-Morrison denominator
            }
            for (int o = 32; o > 0; o >>= 1) {
                g.x += __shfl_down(g.x, o); g.y += __shfl_down(g.y, o);
                d.x += __shfl_down(d.x, o); d.y += __shfl_down(d.y, o);
            }
            if (lane == 0) { gii_s[s] = g; den_s[s] = d; }
        }
        __syncthreads();
        if (tid == 0) {
            cplx probs[2];
            for (int x = 0; x < 2; ++x) {
                const cplx r0 = cadd(cmake(1.0, 0.0), cmul(a.delta[x][0], gii_s[0]));
                const cplx r1 = cadd(cmake(1.0, 0.0), cmul(a.delta[x][1], gii_s[1]));
                const cplx r = cmul(r0, r1);
                probs[x] = cmul(cmake(0.5 * r.x, 0.5 * r.y), a.wfac[x]);          // :551, :198
            }
            const double p0 = fmax(probs[0].x, 0.0), p1 = fmax(probs[1].x, 0.0);
            const double norm = p0 + p1;
            const double r = a.u[(long)w * M + i];
            ++used;
            if (norm > 0) {
                weight *= norm;
                const int xi = r < p0 / norm ? 0 : 1;
                ot = cmul(cmake(2.0 * ot.x, 2.0 * ot.y), probs[xi]);                 // single_det.py:213
                xi_s = xi; stop_s = 0;
                a.fields[(long)w * M + i] = xi;
                if (a.bp_hist) {                                                      // hubbard.py:215-216
                    if (pos < a.hist_cap) a.bp_hist[(long)w * a.hist_cap + pos] = cmake((double)xi, 0.0);
                    ++pos;
                }
                // Sherman-Morrison denominators 1 + vt . (inv u) with vt = phi[i,:] delta
                for (int s = 0; s < 2; ++s) den_s[s] = cadd(cmake(1.0, 0.0), cmul(a.delta[xi][s], den_s[s]));
            } else {
                weight = 0.0; stop_s = 1;
            }
        }
        __syncthreads();
        if (stop_s) break;
        const int xi = xi_s;
        // O^-1[k][l] -= b_k a_l / denom with b_k = q_k delta  (inv_ovlp - inv u vt inv / (1 + vt inv u))
        for (int e = tid; e < 2 * nmax * nmax; e += 256) {
            const int s = e / (nmax * nmax), r = e % (nmax * nmax), k = r / nmax, l = r % nmax;
            const int ns = s == 0 ? a.na : a.nb;
            if (k >= ns || l >= ns) continue;
            const cplx b = cmul(q_s[s][k], a.delta[xi][s]);
            inv[e] = csub(inv[e], cdiv(cmul(b, a_s[s][l]), den_s[s]));
        }
        // phi[i, :] <- phi[i, :] (1 + delta)
        for (int c = tid; c < nt; c += 256) {
            const int s = c < a.na ? 0 : 1;
            const cplx f = cadd(cmake(1.0, 0.0), a.delta[xi][s]);
            phi[(long)i * nt + c] = cmul(phi[(long)i * nt + c], f);
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.weight[w] = weight; a.ot[w] = ot; a.used[w] = used;
        if (a.bp_hist) a.bp_n[w] = (int)(pos < a.hist_cap ? pos : a.hist_cap);
    }
}

// kinetic importance sampling (hubbard.py:163-172) after phi <- bt2 phi and the new overlap
// use_log_shift: calc_otrial applies the shift to the determinant of the INVERSE overlap (single_det.py:159), so the
// overlap it returns is det * exp(+log_shift); `scale` carries that factor (1 when the option is off)
__global__ void hirsch_kin_weight_kernel(double *weight, cplx *ot, const cplx *ot_new, const int *alive, int nw,
                                         double scale) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw || !alive[w]) return;
    const cplx on = cscale(ot_new[w], scale);
    const cplx ratio = cdiv(on, ot[w]);
    if (fabs(atan2(ratio.y, ratio.x)) < 0.5 * 3.14159265358979323846) {
        weight[w] *= ratio.x;
        ot[w] = on;
    } else {
        weight[w] = 0.0;
    }
}

// mode 0: alive = alive0 = |w| > 1e-8 (the driver's test, qmc/afqmc.py:232); 1: alive = alive0 && |w| > 0
__global__ void hirsch_alive_kernel(const double *weight, int *alive, int *alive0, int nw, int mode) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    if (mode == 0) { const int al = fabs(weight[w]) > 1e-8 ? 1 : 0; alive0[w] = al; alive[w] = al; }
    else alive[w] = (alive0[w] && fabs(weight[w]) > 0.0) ? 1 : 0;
}

__global__ void hirsch_eshift_kernel(double *weight, const int *alive0, int nw, double fac) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nw && alive0[w]) weight[w] *= fac;                 // hubbard.py:312
}

__global__ void hirsch_fill_kernel(int *fields, long n) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) fields[i] = -1;
}

int k_hirsch_alive(afq_handle *h, int mode) {
    AFQ_LAUNCH(h, hirsch_alive_kernel, dim3((h->nw + 127) / 128), dim3(128), 0, h->stream, h->weight, h->alive,
                       h->hs_alive0, h->nw, mode);
    AFQ_POST(h);
    return AFQ_OK;
}

// phi <- bt2 phi, inverse overlap, importance sampling, for the walkers flagged alive
int k_hirsch_kinetic(afq_handle *h) {
    int rc;
    if ((rc = k_onebody(h))) return rc;
    if ((rc = k_inverse_overlap(h, h->hs_oinv, h->ovlp_new))) return rc;
    AFQ_LAUNCH(h, hirsch_kin_weight_kernel, dim3((h->nw + 127) / 128), dim3(128), 0, h->stream, h->weight, h->ot,
                       h->ovlp_new, h->alive, h->nw, h->log_shift_on ? exp(h->log_shift) : 1.0);
    AFQ_POST(h);
    return AFQ_OK;
}

// ---- two_body_direct (propagation/hubbard.py:222-275, `single_site_update: False`): every site's field is drawn from the
// dynamic force bias of the walker's CURRENT Green's function (n_i = G_ii per spin, from the inverse overlap the kinetic step
// left behind), all rows of phi are scaled at once, ONE overlap and one importance-sampling test follow.
struct HirschDirectArgs {
    int M, na, nb, nt, nmax, charge;
    cplx *phi;
    const cplx *psi, *oinv;
    const double *u;
    const int *alive;
    int *fields, *used;
    double *fbfac;
    cplx *wfac_out;
    cplx gamma, auxf[2][2], wfac[2];
};

__device__ inline cplx cexp_(cplx z) {
    double sn, cs;
    sincos(z.y, &sn, &cs);
    const double e = exp(z.x);
    return cmake(e * cs, e * sn);
}

__global__ __launch_bounds__(256) void hirsch_direct_sites_kernel(HirschDirectArgs a) {
    __shared__ double fac_s[256];
    __shared__ int xi_s[256];
    const int w = blockIdx.x, tid = threadIdx.x;
    if (!a.alive[w]) { if (tid == 0) a.used[w] = 0; return; }
    const int M = a.M, nt = a.nt, nmax = a.nmax;
    cplx *phi = a.phi + (long)w * M * nt;
    const cplx *og = a.oinv + (long)w * 2 * nmax * nmax;
    for (int i0 = 0; i0 < M; i0 += 256) {
        const int i = i0 + tid;
        if (i < M) {
            cplx n[2];
            for (int s = 0; s < 2; ++s) {
                const int ns = s == 0 ? a.na : a.nb, off = s == 0 ? 0 : a.na;
                const cplx *iv = og + (long)s * nmax * nmax;
                cplx g = cmake(0.0, 0.0);
                for (int k = 0; k < ns; ++k) {
                    cplx q = cmake(0.0, 0.0);
                    for (int l = 0; l < ns; ++l) cfma(q, iv[k * nmax + l], phi[(long)i * nt + off + l]);
                    cfma(g, cconj(a.psi[(long)i * nt + off + k]), q);
                }
                n[s] = g;
            }
            const cplx fb = a.charge ? cmake(n[0].x + n[1].x - 1.0, n[0].y + n[1].y) : cmake(n[0].x - n[1].x, n[0].y - n[1].y);
            const cplx gf = cmul(a.gamma, fb);
            const double ep = cexp_(gf).x, em = cexp_(cmake(-gf.x, -gf.y)).x;            // exp(+-gamma fb).real, :248-249
            const double pp = 0.5 * ep, pm = 0.5 * em, norm = pp + pm;
            const int xi = a.u[(long)w * M + i] < pp / norm ? 0 : 1;
            xi_s[tid] = xi;
            fac_s[tid] = 0.5 * norm * (xi == 0 ? em : ep);                                 // :254, :257
            a.fields[(long)w * M + i] = xi;
        }
        __syncthreads();
        if (tid == 0) {
            double fb_fac = i0 == 0 ? 1.0 : a.fbfac[w];
            cplx wf = i0 == 0 ? cmake(1.0, 0.0) : a.wfac_out[w];
            const int cnt = M - i0 < 256 ? M - i0 : 256;
            for (int j = 0; j < cnt; ++j) { fb_fac *= fac_s[j]; wf = cmul(wf, a.wfac[xi_s[j]]); }   // in site order
            a.fbfac[w] = fb_fac; a.wfac_out[w] = wf;
            a.used[w] = M;
        }
        // rows i0 .. i0 + 255: phi[i, :na] *= auxf[x_i, 0], phi[i, na:] *= auxf[x_i, 1]   (:259-264)
        const int cnt = M - i0 < 256 ? M - i0 : 256;
        for (int e = tid; e < cnt * nt; e += 256) {
            const int j = e / nt, c = e - j * nt;
            const cplx f = a.auxf[xi_s[j]][c < a.na ? 0 : 1];
            cplx *x = phi + (long)(i0 + j) * nt + c;
            *x = cmul(*x, f);
        }
        __syncthreads();
    }
}

// ratio = wfac ovlp / ot; |arg| < pi / 2: ot = ovlp, weight *= (fb_fac ratio).real, else weight = 0   (:265-275)
__global__ void hirsch_direct_weight_kernel(double *weight, cplx *ot, const cplx *ot_new, const cplx *wfac, const double *fbfac,
                                            const int *alive, int nw, double scale) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw || !alive[w]) return;
    const cplx ov = cscale(ot_new[w], scale);
    const cplx ratio = cdiv(cmul(wfac[w], ov), ot[w]);
    if (fabs(atan2(ratio.y, ratio.x)) < 0.5 * 3.14159265358979323846) {
        ot[w] = ov;
        weight[w] *= fbfac[w] * ratio.x;
    } else {
        weight[w] = 0.0;
    }
}

static int k_hirsch_two_body_direct(afq_handle *h) {
    const int nmax = h->na > h->nb ? h->na : h->nb;
    if (h->nbp > 0) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "the direct Hirsch update records no field history (hubbard.py:222-275)");
    if (!h->hs_fbfac) AFQ_HIP(h, hipMalloc(&h->hs_fbfac, sizeof(double) * h->nw));
    {
        const long n = (long)h->nw * h->M;
        AFQ_LAUNCH(h, hirsch_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->hs_fields, n);
        AFQ_POST(h);
    }
    HirschDirectArgs a;
    a.M = h->M; a.na = h->na; a.nb = h->nb; a.nt = h->nt; a.nmax = nmax; a.charge = h->hs_charge ? 1 : 0;
    a.phi = h->phi; a.psi = h->psi; a.oinv = h->hs_oinv; a.u = h->hs_u; a.alive = h->alive;
    a.fields = h->hs_fields; a.used = h->hs_used; a.fbfac = h->hs_fbfac; a.wfac_out = h->cmf;
    a.gamma = h->hs_gamma;
    for (int x = 0; x < 2; ++x) { a.wfac[x] = h->hs_wfac[x]; for (int s = 0; s < 2; ++s) a.auxf[x][s] = h->hs_auxf[x][s]; }
    AFQ_LAUNCH(h, hirsch_direct_sites_kernel, dim3(h->nw), dim3(256), 0, h->stream, a);
    AFQ_POST(h);
    int rc;
    // walker.calc_overlap(trial), :265 -- with use_log_shift the reference's two overlap routines carry OPPOSITE factors:
    // calc_overlap (single_det.py:192) returns det * exp(-log_shift), calc_otrial (:159-161, used by the kinetic and the
    // free-projection updates) 1 / (det(O^-1) exp(-log_shift)) = det * exp(+log_shift).  Both are reproduced as they are
    // (traj_hirsch_logshift.npz); the factors below are therefore exp(-log_shift) here and exp(+log_shift) there.
    if ((rc = k_inverse_overlap(h, h->hs_oinv, h->ovlp_new))) return rc;
    AFQ_LAUNCH(h, hirsch_direct_weight_kernel, dim3((h->nw + 127) / 128), dim3(128), 0, h->stream, h->weight, h->ot,
               h->ovlp_new, h->cmf, h->hs_fbfac, h->alive, h->nw, h->log_shift_on ? exp(-h->log_shift) : 1.0);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_hirsch_two_body(afq_handle *h) {
    if (h->hs_direct) return k_hirsch_two_body_direct(h);
    const int nmax = h->na > h->nb ? h->na : h->nb;
    {
        const long n = (long)h->nw * h->M;
        AFQ_LAUNCH(h, hirsch_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->hs_fields, n);
    }
    HirschArgs a;
    a.M = h->M; a.na = h->na; a.nb = h->nb; a.nt = h->nt; a.nw = h->nw; a.nmax = nmax;
    a.phi = h->phi; a.psi = h->psi; a.oinv = h->hs_oinv; a.weight = h->weight; a.ot = h->ot; a.u = h->hs_u;
    a.fields = h->hs_fields; a.used = h->hs_used; a.alive = h->alive;
    for (int x = 0; x < 2; ++x) { a.wfac[x] = h->hs_wfac[x]; for (int s = 0; s < 2; ++s) a.delta[x][s] = h->hs_delta[x][s]; }
    a.bp_hist = h->nbp > 0 ? h->bp_hist : nullptr; a.bp_n = h->bp_n; a.hist_cap = (long)h->nbp * h->M;
    if (nmax > 128) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "Hirsch propagator: N <= 128 per spin");
    size_t lds = sizeof(cplx) * 2 * (size_t)nmax * nmax;
    a.inv_in_lds = lds <= 150 * 1024;
    if (!a.inv_in_lds) lds = 0;
    static size_t lds_set[AFQ_MAX_DEVICES] = {0};
    AFQ_HIP(h, afq_raise_lds((const void *)hirsch_two_body_kernel, lds, lds_set));
    AFQ_LAUNCH(h, hirsch_two_body_kernel, dim3(h->nw), dim3(256), lds, h->stream, a);
    AFQ_POST(h);
    return AFQ_OK;
}

// ---- free projection (propagation/hubbard.py:303-343): no importance sampling, every site takes field 0 or 1 with
// probability 1/2 (r < 0.5 -> 0), row i of phi is scaled by 1 + delta[x_i, spin], wfac = prod_i aux_wfac[x_i] (in site
// order, as the reference multiplies), then weight *= exp(dt eshift) |wfac|, phase *= exp(i arg wfac), ot = <psi_T|phi>.
__global__ __launch_bounds__(256) void hirsch_free_sites_kernel(cplx *phi_all, const double *u, const int *alive, int *fields,
                                                                cplx *wfac_out, int M, int nt, int na, cplx d00, cplx d01,
                                                                cplx d10, cplx d11, cplx wf0, cplx wf1) {
    const int w = blockIdx.x;
    if (!alive[w]) return;
    cplx *phi = phi_all + (long)w * M * nt;
    const double *uw = u + (long)w * M;
    for (int e = threadIdx.x; e < M * nt; e += blockDim.x) {
        const int i = e / nt, c = e - i * nt;
        const int xi = uw[i] < 0.5 ? 0 : 1;                                  // :326-330
        const cplx d = xi == 0 ? (c < na ? d00 : d01) : (c < na ? d10 : d11);
        const cplx v = phi[e];
        phi[e] = cadd(v, cmul(v, d));                                         // phi + phi * delta, :331-334
    }
    if (threadIdx.x == 0) {
        cplx wf = cmake(1.0, 0.0);
        for (int i = 0; i < M; ++i) {
            const int xi = uw[i] < 0.5 ? 0 : 1;
            fields[(long)w * M + i] = xi;
            wf = cmul(wf, xi == 0 ? wf0 : wf1);                               // :335
        }
        wfac_out[w] = wf;
    }
}

__global__ void hirsch_free_weight_kernel(double *weight, cplx *phase, cplx *ot, const cplx *ot_new, const cplx *wfac,
                                          const int *alive, int nw, double efac, double scale) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw || !alive[w]) return;
    const cplx wf = wfac[w];
    const double magn = hypot(wf.x, wf.y), dtheta = atan2(wf.y, wf.x);        // cmath.polar(wfac), :341
    weight[w] *= efac * magn;                                                 // :342
    double sn, cs;
    sincos(dtheta, &sn, &cs);
    phase[w] = cmul(phase[w], cmake(cs, sn));                                 // :343
    ot[w] = cscale(ot_new[w], scale);                                         // :344
}

int k_hirsch_free(afq_handle *h, double eshift) {
    int rc;
    if ((rc = k_onebody(h))) return rc;                                       // kinetic_real, :320
    {
        const long n = (long)h->nw * h->M;
        AFQ_LAUNCH(h, hirsch_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->hs_fields, n);
        AFQ_POST(h);
    }
    AFQ_LAUNCH(h, hirsch_free_sites_kernel, dim3(h->nw), dim3(256), 0, h->stream, h->phi, h->hs_u, h->alive, h->hs_fields,
               h->cmf, h->M, h->nt, h->na, h->hs_delta[0][0], h->hs_delta[0][1], h->hs_delta[1][0], h->hs_delta[1][1],
               h->hs_wfac[0], h->hs_wfac[1]);
    AFQ_POST(h);
    if ((rc = k_onebody(h))) return rc;                                       // :336
    if ((rc = k_inverse_overlap(h, h->hs_oinv, h->ovlp_new))) return rc;      // :337-339
    AFQ_LAUNCH(h, hirsch_free_weight_kernel, dim3((h->nw + 127) / 128), dim3(128), 0, h->stream, h->weight, h->phase, h->ot,
               h->ovlp_new, h->cmf, h->alive, h->nw, exp(h->dt * eshift), h->log_shift_on ? exp(h->log_shift) : 1.0);
    AFQ_POST(h);
    return AFQ_OK;
}

int k_hirsch_eshift(afq_handle *h, double fac) {
    AFQ_LAUNCH(h, hirsch_eshift_kernel, dim3((h->nw + 127) / 128), dim3(128), 0, h->stream, h->weight, h->hs_alive0,
                       h->nw, fac);
    AFQ_POST(h);
    return AFQ_OK;
}

// ---- back-propagation of the discrete fields (propagation/hubbard.py:568-600,634-672) -------------------------
// B(x)^H = BT2^H diag(auxf[x_j, spin]) BT2^H with the SPIN decomposition's real auxf = exp(+-gamma),
// gamma = arccosh(exp(dt U / 2)), whatever decomposition the propagator used (:589-591).
__global__ void bp_hirsch_alive_kernel(const int *bp_n, int *alive, int nw, int K, int i) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nw) alive[w] = i < bp_n[w] / K ? 1 : 0;
}

__global__ void bp_hirsch_scale_kernel(const cplx *hist, const int *bp_n, cplx *phi, int M, int nt, int na, int nbp,
                                       int i, double eg, double emg) {
    const int w = blockIdx.x;
    const int n = bp_n[w] / M;
    if (i >= n) return;
    const cplx *row = hist + ((long)w * nbp + (n - 1 - i)) * M;
    cplx *p = phi + (long)w * M * nt;
    for (int e = threadIdx.x; e < M * nt; e += blockDim.x) {
        const int j = e / nt, c = e % nt;
        const int x = (int)row[j].x;                             // int(xi.real), :590
        const bool up = c < na;
        const double f = (x == 0) == up ? eg : emg;             // auxf[0] = (e^g, e^-g), auxf[1] = (e^-g, e^g)
        p[e] = cscale(p[e], f);
    }
}

int k_bp_hirsch_step(afq_handle *h, int i) {
    AFQ_LAUNCH(h, bp_hirsch_alive_kernel, dim3((h->nw + 127) / 128), dim3(128), 0, h->stream, h->bp_n, h->alive,
               h->nw, h->M, i);
    AFQ_POST(h);
    int rc = k_onebody(h);                                       // BT2^H (h->BH1 points at the adjoint here)
    if (rc) return rc;
    const double g = acosh(exp(0.5 * h->dt * h->U));
    AFQ_LAUNCH(h, bp_hirsch_scale_kernel, dim3(h->nw), dim3(256), 0, h->stream, h->bp_hist, h->bp_n, h->phi, h->M,
               h->nt, h->na, h->nbp, i, exp(g), exp(-g));
    AFQ_POST(h);
    return k_onebody(h);
}
