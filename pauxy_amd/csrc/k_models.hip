// Model-specific pieces that are NOT dense contractions:
//   Hubbard: diagonal HS potential (propagation/hubbard.py:409-413, :475-480), its
//            degree-n Taylor propagator as a row scaling, energy (estimators/hubbard.py:93-114)
//   UEG:     sparse force bias / HS potential (propagation/planewave.py:57-112) and the
//            index-list energy (estimators/ueg.py:27-88, ueg_kernels.pyx:42-75)
// These are HBM/gather bound: coalesced loads, one wavefront per output where a
// reduction is needed, no MFMA.
#include "afq_internal.h"

// ------------------------------------------------------------------ Hubbard
__global__ void vhs_hubbard_kernel(const cplx *xs, cplx *vd, int nw, int M, int nv, double sqrt_dt, double dt,
                                   double U, int spin) {
    const int w = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= M) return;
    const cplx x = xs[(long)w * M + n];
    if (!spin) {
        // sqrt(dt) * i sqrt(U) * x
        const double f = sqrt_dt * sqrt(U);
        vd[(long)w * M + n] = cmake(-f * x.y, f * x.x);
    } else {
        const double f = sqrt(dt * U);           // (dt U)^0.5, propagation/hubbard.py:445
        vd[((long)w * 2 + 0) * M + n] = cmake(-f * x.x, -f * x.y);
        vd[((long)w * 2 + 1) * M + n] = cmake(f * x.x, f * x.y);
    }
}

int k_vhs_hubbard(afq_handle *h) {
    const int spin = (h->flags & AFQ_PROP_HUBBARD_SPIN) ? 1 : 0;
    AFQ_LAUNCH(h, vhs_hubbard_kernel, dim3((h->M + 127) / 128, h->nw), dim3(128), 0, h->stream, h->xs,
                       h->vhs, h->nw, h->M, h->nv, h->sqrt_dt, h->dt, h->U, spin);
    AFQ_POST(h);
    return AFQ_OK;
}

// phi[p, col] <- sum_{n<=order} d_p^n / n! phi[p, col], same recurrence as
// propagation/continuous.py:104-107 (Temp = d*Temp/n; phi += Temp)
__global__ void exp_diag_kernel(cplx *phi, const cplx *vd, const int *alive, int M, int nt, int na, int nv,
                                int order) {
    const int w = blockIdx.y;
    if (!alive[w]) return;
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= (long)M * nt) return;
    const int p = (int)(e / nt), col = (int)(e % nt);
    const int s = (nv == 2 && col >= na) ? 1 : 0;
    const cplx d = vd[((long)w * nv + s) * M + p];
    cplx acc = phi[(long)w * M * nt + e];
    cplx t = acc;
    for (int n = 1; n <= order; ++n) {
        t = cmul(d, t);
        t = cmake(t.x / n, t.y / n);
        acc = cadd(acc, t);
    }
    phi[(long)w * M * nt + e] = acc;
}

int k_apply_exponential_diag(afq_handle *h, const cplx *vd) {
    const long per = (long)h->M * h->nt;
    AFQ_LAUNCH(h, exp_diag_kernel, dim3((unsigned)((per + 255) / 256), h->nw), dim3(256), 0, h->stream,
                       h->phi, vd, h->alive, h->M, h->nt, h->na, h->nv, h->exp_order);
    AFQ_POST(h);
    return AFQ_OK;
}

// The same propagator as ONE factor per (walker, HS matrix, site): f = sum_{n<=order} d^n / n! with the recurrence
// t = d t / n.  The step multiplies row p of B phi by f in the epilogue of the one-body product ahead of it
// (k_onebody(h, factors)) instead of sweeping the walkers through memory once more.
__global__ void exp_diag_factor_kernel(const cplx *vd, cplx *out, long n, int order) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const cplx d = vd[e];
    cplx acc = cmake(1.0, 0.0), t = acc;
    for (int k = 1; k <= order; ++k) {
        t = cmul(d, t);
        t = cmake(t.x / k, t.y / k);
        acc = cadd(acc, t);
    }
    out[e] = acc;
}

int k_exp_diag_factors(afq_handle *h, const cplx *vd, cplx *out) {
    const long n = (long)h->nw * h->nv * h->M;
    AFQ_LAUNCH(h, exp_diag_factor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, vd, out, n, h->exp_order);
    AFQ_POST(h);
    return AFQ_OK;
}

// ke = sum_s sum_{i,q} rT[i,q] Ghalf_s[i,q];  pe = U sum_n G_up[n,n] G_dn[n,n]
// (1024 threads per walker, four independent 16-byte loads in flight per thread: the kernel streams the walker's 1 MB
//  Ghalf once -- 268 MB at C4 -- and was latency bound at 256 threads with one load in flight: 255 us, 1 TB/s)
constexpr int EH_THR = 1024;
__global__ __launch_bounds__(EH_THR) void energy_hubbard_kernel(const cplx *rH1, const cplx *ghalf, const cplx *psi,
                                                                cplx *energy, int M, int na, int nb, int nt,
                                                                double U) {
    __shared__ double red[4][EH_THR / 64];
    const int w = blockIdx.x, tid = threadIdx.x;
    const cplx *gh = ghalf + (long)w * nt * M;
    double kr = 0, ki = 0, pr = 0, pi = 0;
    const long total = (long)nt * M;
    long q = tid;
    for (; q + 3 * EH_THR < total; q += 4 * EH_THR) {
        cplx a[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = rH1[q + u * EH_THR]; g[u] = gh[q + u * EH_THR]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            kr += a[u].x * g[u].x - a[u].y * g[u].y;
            ki += a[u].x * g[u].y + a[u].y * g[u].x;
        }
    }
    for (; q < total; q += EH_THR) {
        const cplx a = rH1[q], g = gh[q];
        kr += a.x * g.x - a.y * g.y;
        ki += a.x * g.y + a.y * g.x;
    }
    // G_s[n, n] = sum_i conj(psi_s[n, i]) Ghalf_s[i, n]: four threads per site, each a quarter of the orbitals of both spins
    {
        const int n = tid >> 2, part = tid & 3;
        cplx g[2] = {cmake(0.0, 0.0), cmake(0.0, 0.0)};
        for (int nn = n; nn < M; nn += EH_THR / 4) {
            g[0] = cmake(0.0, 0.0); g[1] = cmake(0.0, 0.0);
            for (int s = 0; s < 2; ++s) {
                const int ns = s == 0 ? na : nb, off = s == 0 ? 0 : na;
                for (int i = part; i < ns; i += 4) cfma(g[s], cconj(psi[(long)nn * nt + off + i]), gh[(long)(off + i) * M + nn]);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                g[s].x += __shfl_xor(g[s].x, 1); g[s].y += __shfl_xor(g[s].y, 1);
                g[s].x += __shfl_xor(g[s].x, 2); g[s].y += __shfl_xor(g[s].y, 2);
            }
            if (part == 0) {
                const cplx t = cmul(g[0], g[1]);
                pr += U * t.x; pi += U * t.y;
            }
        }
    }
    double v[4] = {kr, ki, pr, pi};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        if ((tid & 63) == 0) red[k][tid >> 6] = x;
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double x = 0.0;
            for (int i = 0; i < EH_THR / 64; ++i) x += red[k][i];
            v[k] = x;
        }
        energy[3 * w + 0] = cmake(v[0] + v[2], v[1] + v[3]);
        energy[3 * w + 1] = cmake(v[0], v[1]);
        energy[3 * w + 2] = cmake(v[2], v[3]);
    }
}

int k_energy_hubbard(afq_handle *h) {
    AFQ_LAUNCH(h, energy_hubbard_kernel, dim3(h->nw), dim3(EH_THR), 0, h->stream, h->rH1, h->ghalf, h->psi,
                       h->energy, h->M, h->na, h->nb, h->nt, h->U);
    AFQ_POST(h);
    return AFQ_OK;
}

// ---------------------------------------------------------------------- UEG
// vbias[w, q]      = sum_{nz in column q of iA} val * (G_up + G_dn)[row]
// vbias[w, nq + q] = same with iB                      (propagation/planewave.py:70-73)
// one wavefront per (walker, column); G is the full [2, M, M] Green's function.
__global__ void vbias_ueg_kernel(const cplx *G, cplx *vbias, int nw, int M, int nq, const int64_t *Acp,
                                 const int64_t *Arow, const cplx *Aval, const int64_t *Bcp,
                                 const int64_t *Brow, const cplx *Bval) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (item >= (long)nw * 2 * nq) return;
    const int w = (int)(item / (2 * nq));
    const int col = (int)(item % (2 * nq));
    const bool isB = col >= nq;
    const int q = isB ? col - nq : col;
    const int64_t *cp = isB ? Bcp : Acp, *row = isB ? Brow : Arow;
    const cplx *val = isB ? Bval : Aval;
    const cplx *Ga = G + (long)w * 2 * M * M, *Gb = Ga + (long)M * M;
    double sr = 0, si = 0;
    for (int64_t z = cp[q] + lane; z < cp[q + 1]; z += 64) {
        const int64_t r = row[z];
        const cplx g = cadd(Ga[r], Gb[r]), v = val[z];
        sr += g.x * v.x - g.y * v.y;
        si += g.x * v.y + g.y * v.x;
    }
    for (int off = 32; off > 0; off >>= 1) { sr += __shfl_down(sr, off); si += __shfl_down(si, off); }
    if (lane == 0) vbias[(long)w * 2 * nq + col] = cmake(sr, si);
}

// Same contraction with the walker's spin-summed Green's function staged in LDS (M^2 x 16 B, 138 KB at
// C2): one work-group per walker, one thread per column in turn.  The gathers run along shifted diagonals
// of G (one element per cache line), which makes the global-memory version L2-transaction bound.
__global__ __launch_bounds__(512) void vbias_ueg_lds_kernel(const cplx *G, cplx *vbias, int M, int ncol, int L,
                                                            const int *ell_row, const cplx *ell_val) {
    extern __shared__ __align__(16) unsigned char smem[];
    cplx *gs = (cplx *)smem;
    const int w = blockIdx.x;
    const cplx *Ga = G + (long)w * 2 * M * M, *Gb = Ga + (long)M * M;
    for (int e = threadIdx.x; e < M * M; e += 512) gs[e] = cadd(Ga[e], Gb[e]);
    __syncthreads();
    for (int col = threadIdx.x; col < ncol; col += 512) {
        double sr = 0, si = 0;
#pragma unroll 4
        for (int k = 0; k < L; ++k) {                          // padded entries carry a zero value
            const cplx g = gs[ell_row[(long)k * ncol + col]], v = ell_val[(long)k * ncol + col];
            sr += g.x * v.x - g.y * v.y;
            si += g.x * v.y + g.y * v.x;
        }
        vbias[(long)w * ncol + col] = cmake(sr, si);
    }
}

int k_vbias_ueg(afq_handle *h) {
    const size_t lds = sizeof(cplx) * (size_t)h->M * h->M;
    if (lds <= 160 * 1024) {
        static size_t lds_set[AFQ_MAX_DEVICES] = {0};
        AFQ_HIP(h, afq_raise_lds((const void *)vbias_ueg_lds_kernel, lds, lds_set));
        AFQ_LAUNCH(h, vbias_ueg_lds_kernel, dim3(h->nw), dim3(512), lds, h->stream, h->G, h->vbias, h->M, 2 * h->nq,
                           h->ell_len, h->ell_row, h->ell_val);
        AFQ_POST(h);
        return AFQ_OK;
    }
    const long items = (long)h->nw * 2 * h->nq;
    AFQ_LAUNCH(h, vbias_ueg_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, h->stream, h->G,
                       h->vbias, h->nw, h->M, h->nq, h->iA_colptr, h->iA_row, h->iA_val, h->iB_colptr,
                       h->iB_row, h->iB_val);
    AFQ_POST(h);
    return AFQ_OK;
}

// VHS[w, r] = sqrt(dt) * (sum_{nz in row r of iA} val xs[q] + sum_{nz in row r of iB} val xs[nq+q])
// (propagation/planewave.py:109-112); one thread per (walker, matrix element), CSR rows are short.
__global__ void vhs_ueg_kernel(const cplx *xs, cplx *vhs, int nw, int M, int nq, double sqrt_dt,
                               const int64_t *Arp, const int64_t *Acol, const cplx *Aval, const int64_t *Brp,
                               const int64_t *Bcol, const cplx *Bval) {
    const int w = blockIdx.y;
    const long r = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (r >= (long)M * M) return;
    const cplx *x = xs + (long)w * 2 * nq;
    cplx acc = cmake(0.0, 0.0);
    for (int64_t z = Arp[r]; z < Arp[r + 1]; ++z) cfma(acc, Aval[z], x[Acol[z]]);
    for (int64_t z = Brp[r]; z < Brp[r + 1]; ++z) cfma(acc, Bval[z], x[nq + Bcol[z]]);
    vhs[(long)w * M * M + r] = cmake(sqrt_dt * acc.x, sqrt_dt * acc.y);
}

int k_vhs_ueg(afq_handle *h) {
    const long mm = (long)h->M * h->M;
    AFQ_LAUNCH(h, vhs_ueg_kernel, dim3((unsigned)((mm + 255) / 256), h->nw), dim3(256), 0, h->stream, h->xs,
                       h->vhs, h->nw, h->M, h->nq, h->sqrt_dt, h->iA_rowptr, h->iA_col, h->iA_rval,
                       h->iB_rowptr, h->iB_col, h->iB_rval);
    AFQ_POST(h);
    return AFQ_OK;
}

// Energy: one workgroup per walker; each wavefront takes q-vectors in turn and
// gathers Gkpq, Gpmq and the exchange double sum for both spins.
// STAGED: only the rows of G that the index lists touch (the trial's occupied orbitals, ueg_rows) are
// copied to LDS for both spins and every gather goes there; otherwise the gathers read global memory.
template <bool STAGED>
__global__ __launch_bounds__(1024) void energy_ueg_kernel(const cplx *G, cplx *energy, int M, int nq,
                                                         const int64_t *kpq_off, const int64_t *kpq_i,
                                                         const int64_t *kpq_kpq, const int64_t *pmq_off,
                                                         const int64_t *pmq_i, const int64_t *pmq_pmq,
                                                         const double *vqvec, double vol, const double *H1diag,
                                                         const int *rmap, const int *rows, int nrows) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ double red[16];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
    const cplx *Gg[2] = {G + (long)w * 2 * M * M, G + (long)w * 2 * M * M + (long)M * M};
    const cplx *Gs[2] = {Gg[0], Gg[1]};
    if (STAGED) {
        cplx *st = (cplx *)smem;
        for (int e = tid; e < 2 * nrows * M; e += (int)blockDim.x) {
            const int s = e / (nrows * M), r = (e / M) % nrows, c = e % M;
            st[e] = Gg[s][(long)rows[r] * M + c];
        }
        Gs[0] = st; Gs[1] = st + (long)nrows * M;
    }
    auto RI = [&](int64_t i) -> long { return STAGED ? (long)rmap[i] : (long)i; };
    double ker = 0, kei = 0;
    for (int e = tid; e < 2 * M; e += (int)blockDim.x) {
        const int s = e / M, i = e % M;
        const cplx g = Gg[s][(long)i * M + i];
        ker += H1diag[e] * g.x; kei += H1diag[e] * g.y;
    }
    if (STAGED) __syncthreads();
    double per = 0, pei = 0;
    for (int q = wave; q < nq; q += nwave) {
        const int64_t k0 = kpq_off[q], p0 = pmq_off[q];
        const int nk = (int)(kpq_off[q + 1] - k0), np = (int)(pmq_off[q + 1] - p0);   // <= N: 32-bit index math
        cplx gk[2], gp[2], gx[2];
        for (int s = 0; s < 2; ++s) {
            double ar = 0, ai = 0, br = 0, bi = 0, cr = 0, ci = 0;
            for (int z = lane; z < nk; z += 64) {
                const cplx g = Gs[s][RI(kpq_i[k0 + z]) * M + kpq_kpq[k0 + z]];
                ar += g.x; ai += g.y;
            }
            for (int z = lane; z < np; z += 64) {
                const cplx g = Gs[s][RI(pmq_i[p0 + z]) * M + pmq_pmq[p0 + z]];
                br += g.x; bi += g.y;
            }
            // sum_{a,b} G[pmq_i[b], kpq[a]] * G[kpq_i[a], pmq[b]]
            for (int z = lane; z < nk * np; z += 64) {
                const int ia = z / np, ibb = z % np;
                const cplx g1 = Gs[s][RI(pmq_i[p0 + ibb]) * M + kpq_kpq[k0 + ia]];
                const cplx g2 = Gs[s][RI(kpq_i[k0 + ia]) * M + pmq_pmq[p0 + ibb]];
                cr += g1.x * g2.x - g1.y * g2.y;
                ci += g1.x * g2.y + g1.y * g2.x;
            }
            for (int off = 32; off > 0; off >>= 1) {
                ar += __shfl_xor(ar, off); ai += __shfl_xor(ai, off);
                br += __shfl_xor(br, off); bi += __shfl_xor(bi, off);
                cr += __shfl_xor(cr, off); ci += __shfl_xor(ci, off);
            }
            gk[s] = cmake(ar, ai); gp[s] = cmake(br, bi); gx[s] = cmake(cr, ci);
        }
        if (lane == 0) {
            // (Gkpq Gpmq - Gprod)_aa + (..)_bb + Gkpq_a Gpmq_b + Gkpq_b Gpmq_a
            cplx t = csub(cmul(gk[0], gp[0]), gx[0]);
            t = cadd(t, csub(cmul(gk[1], gp[1]), gx[1]));
            t = cadd(t, cmul(gk[0], gp[1]));
            t = cadd(t, cmul(gk[1], gp[0]));
            const double f = vqvec[q] / (2.0 * vol);
            per += f * t.x; pei += f * t.y;
        }
    }
    double v[4] = {ker, kei, per, pei};
    for (int k = 0; k < 4; ++k) {
        double x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        __syncthreads();
        if (lane == 0) red[wave] = x;
        __syncthreads();
        double t = 0.0;
        for (int i = 0; i < nwave; ++i) t += red[i];
        v[k] = t;
    }
    if (tid == 0) {
        energy[3 * w + 0] = cmake(v[0] + v[2], v[1] + v[3]);
        energy[3 * w + 1] = cmake(v[0], v[1]);
        energy[3 * w + 2] = cmake(v[2], v[3]);
    }
}

// Thread-per-q variant of the staged kernel: the index lists are short (at most N entries per q-vector), so a wave per
// q spends its time on dependent index loads and cross-lane reductions of mostly idle lanes (296 us at the C2 sizes:
// 47 q-vectors per wave, one after the other).  Here every thread owns one q-vector: the packed lists (row of the staged
// G, column) are read straight from the cached global arrays, all gathers go to LDS, nothing is reduced until the end.
__global__ __launch_bounds__(1024) void energy_ueg_q_kernel(const cplx *G, cplx *energy, int M, int nq, const int *koff,
                                                            const int *kp, const int *poff, const int *pm,
                                                            const double *vqvec, double vol, const double *H1diag,
                                                            const int *rows, int nrows) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ double red[16];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
    const cplx *Gg[2] = {G + (long)w * 2 * M * M, G + (long)w * 2 * M * M + (long)M * M};
    cplx *st = (cplx *)smem;                                     // [2][nrows][M]
    for (int e = tid; e < 2 * nrows * M; e += (int)blockDim.x) {
        const int s = e / (nrows * M), r = (e / M) % nrows, c = e % M;
        st[e] = Gg[s][(long)rows[r] * M + c];
    }
    double ker = 0, kei = 0;
    for (int e = tid; e < 2 * M; e += (int)blockDim.x) {
        const int s = e / M, i = e % M;
        const cplx g = Gg[s][(long)i * M + i];
        ker += H1diag[e] * g.x; kei += H1diag[e] * g.y;
    }
    __syncthreads();
    double per = 0, pei = 0;
    for (int q = tid; q < nq; q += (int)blockDim.x) {
        const int k0 = koff[q], nk = koff[q + 1] - k0, p0 = poff[q], np = poff[q + 1] - p0;
        cplx gk[2], gp[2], gx[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const cplx *Gs = st + (long)s * nrows * M;
            double ar = 0, ai = 0, br = 0, bi = 0, cr = 0, ci = 0;
            for (int z = 0; z < nk; ++z) {
                const int e = kp[k0 + z];
                const cplx g = Gs[(e >> 16) * M + (e & 0xffff)];
                ar += g.x; ai += g.y;
            }
            for (int z = 0; z < np; ++z) {
                const int e = pm[p0 + z];
                const cplx g = Gs[(e >> 16) * M + (e & 0xffff)];
                br += g.x; bi += g.y;
            }
            // sum_{a,b} G[pmq_i[b], kpq[a]] * G[kpq_i[a], pmq[b]]
            for (int ia = 0; ia < nk; ++ia) {
                const int ea = kp[k0 + ia];
                const int ra = (ea >> 16) * M, ca = ea & 0xffff;
                for (int ib = 0; ib < np; ++ib) {
                    const int eb = pm[p0 + ib];
                    const cplx g1 = Gs[(eb >> 16) * M + ca];
                    const cplx g2 = Gs[ra + (eb & 0xffff)];
                    cr += g1.x * g2.x - g1.y * g2.y;
                    ci += g1.x * g2.y + g1.y * g2.x;
                }
            }
            gk[s] = cmake(ar, ai); gp[s] = cmake(br, bi); gx[s] = cmake(cr, ci);
        }
        // (Gkpq Gpmq - Gprod)_aa + (..)_bb + Gkpq_a Gpmq_b + Gkpq_b Gpmq_a
        cplx t = csub(cmul(gk[0], gp[0]), gx[0]);
        t = cadd(t, csub(cmul(gk[1], gp[1]), gx[1]));
        t = cadd(t, cmul(gk[0], gp[1]));
        t = cadd(t, cmul(gk[1], gp[0]));
        const double f = vqvec[q] / (2.0 * vol);
        per += f * t.x; pei += f * t.y;
    }
    double v[4] = {ker, kei, per, pei};
    for (int k = 0; k < 4; ++k) {
        double x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        __syncthreads();
        if (lane == 0) red[wave] = x;
        __syncthreads();
        double t = 0.0;
        for (int i = 0; i < nwave; ++i) t += red[i];
        v[k] = t;
    }
    if (tid == 0) {
        energy[3 * w + 0] = cmake(v[0] + v[2], v[1] + v[3]);
        energy[3 * w + 1] = cmake(v[0], v[1]);
        energy[3 * w + 2] = cmake(v[2], v[3]);
    }
}

int k_energy_ueg(afq_handle *h) {
    const size_t lds = sizeof(cplx) * 2 * (size_t)h->ueg_nrows * h->M;
    if (lds <= 120 * 1024 && h->ueg_kp && !AFQ_KNOB_SET("AFQ_UEG_WAVE_Q")) {
        static size_t lds_set[AFQ_MAX_DEVICES] = {0};
        AFQ_HIP(h, afq_raise_lds((const void *)energy_ueg_q_kernel, lds, lds_set));
        AFQ_LAUNCH(h, energy_ueg_q_kernel, dim3(h->nw), dim3(1024), lds, h->stream, h->G, h->energy, h->M, h->nq,
                   h->ueg_koff, h->ueg_kp, h->ueg_poff, h->ueg_pm, h->vqvec, h->vol, h->H1diag, h->ueg_rows, h->ueg_nrows);
        AFQ_POST(h);
        return AFQ_OK;
    }
    if (lds <= 120 * 1024) {
        static size_t lds_set[AFQ_MAX_DEVICES] = {0};
        AFQ_HIP(h, afq_raise_lds((const void *)energy_ueg_kernel<true>, lds, lds_set));
        AFQ_LAUNCH(h, energy_ueg_kernel<true>, dim3(h->nw), dim3(1024), lds, h->stream, h->G, h->energy, h->M, h->nq,
                           h->kpq_off, h->kpq_i, h->kpq_kpq, h->pmq_off, h->pmq_i, h->pmq_pmq, h->vqvec, h->vol,
                           h->H1diag, h->ueg_rmap, h->ueg_rows, h->ueg_nrows);
    } else {
        AFQ_LAUNCH(h, energy_ueg_kernel<false>, dim3(h->nw), dim3(1024), 0, h->stream, h->G, h->energy, h->M, h->nq,
                           h->kpq_off, h->kpq_i, h->kpq_kpq, h->pmq_off, h->pmq_i, h->pmq_pmq, h->vqvec, h->vol,
                           h->H1diag, h->ueg_rmap, h->ueg_rows, h->ueg_nrows);
    }
    AFQ_POST(h);
    return AFQ_OK;
}
