// Local energy of a generic Cholesky Hamiltonian from the FULL one-body Green's function
// (estimators/generic.py:398-434):
//   e1b   = sum_s sum_pq H1_s[p,q] G_s[p,q]
//   X_s[n] = sum_pq L_n[p,q] G_s[p,q],  ecoul = Xa.Xa + Xb.Xb + 2 Xa.Xb
//   T_s[l,k,n] = sum_i G_s[i,l] L_n[i,k],  exx_s = sum_n sum_lk T_s[l,k,n] T_s[k,l,n]
//   E = (e1b + (ecoul - exx)/2 + ecore, e1b + ecore, (ecoul - exx)/2)
// Used where no half-rotated form exists: Green's functions handed in by the caller (the reference's
// local_energy(system, G) free function) and the back-propagated G_bp.  The M^3 K contraction runs on
// the work-group MFMA GEMM engine in chunks of Cholesky vectors (T is 16 M^2 bytes per (G, n) and is
// only kept for one chunk); it is O(M/N) more work than the half-rotated energy kernel and is not on
// the per-step hot path.
#include "mfma_gemm_wg.h"

struct FullGTProb {
    static constexpr bool A_CPLX = true, B_CPLX = false;
    int batch, rows, cols, kdim;     // ng * 2 * nc, M, M, M
    int nc, n0, M, Mp;
    const cplx *G;                   // [ng, 2, M, M]
    const double *L;                 // [K, M, Mp] (Mp = M rounded up to even, zero padded)
    cplx *T;                         // [batch, M, M]
    __device__ bool active(int) const { return true; }
    __device__ cplx loadA(int b, int row, int k) const { return G[((long)(b / nc) * M + k) * M + row]; }
    __device__ cplx loadB(int b, int k, int col) const {
        return cmake(L[((long)(n0 + b % nc) * M + k) * Mp + col], 0.0);
    }
    __device__ const cplx *ptrA(int b, int row, int k) const { return G + ((long)(b / nc) * M + k) * M + row; }
    __device__ const double *ptrB(int b, int k, int col) const {
        return L + ((long)(n0 + b % nc) * M + k) * Mp + col;
    }
    static constexpr bool INCR = true;       // incremental refill of the ring engine
    __device__ int klimit(int) const { return kdim; }
    __device__ const cplx *baseA(int b, int row) const { return G + (long)(b / nc) * M * M + row; }
    __device__ const double *baseB(int b, int col) const { return L + (long)(n0 + b % nc) * M * Mp + col; }
    __device__ long kstepA() const { return M; }
    __device__ long kstepB(int) const { return Mp; }
    __device__ bool rowok(int, int) const { return true; }
    __device__ bool colok(int, int) const { return true; }
    __device__ void store(int b, int row, int col, double re, double im) const {
        T[((long)b * M + row) * M + col] = cmake(re, im);
    }
};

// L_n[i][k] for every n from the (possibly symmetric-packed) transposed hs_pot of the handle
__global__ void fullg_expand_kernel(const double *hsT, long ld, int sym, int M, int Mp, int K, double *L) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long per = (long)M * Mp;
    if (e >= per * K) return;
    const int n = (int)(e / per), r = (int)(e % per), i = r / Mp, k = r % Mp;
    double v = 0.0;
    if (k < M) {
        if (sym) {
            const int p = i < k ? i : k, q = i < k ? k : i;
            v = hsT[(long)n * ld + ((long)p * M - (long)p * (p - 1) / 2 + (q - p))];
        } else {
            v = hsT[(long)n * ld + (long)i * M + k];
        }
    }
    L[e] = v;
}

// one wave per (g, s, n): X = sum_pq L_n[p,q] G_s[p,q]
__global__ void fullg_coulomb_kernel(const cplx *G, const double *L, cplx *X, int ng2, int M, int Mp, int K) {
    const long t = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= (long)ng2 * K) return;
    const int gs = (int)(t / K), n = (int)(t % K), lane = threadIdx.x & 63;
    const cplx *g = G + (long)gs * M * M;
    const double *l = L + (long)n * M * Mp;
    double sx = 0.0, sy = 0.0;
    for (int e = lane; e < M * M; e += 64) {
        const int i = e / M, k = e % M;
        const double v = l[i * Mp + k];
        sx += v * g[e].x; sy += v * g[e].y;
    }
    for (int o = 32; o > 0; o >>= 1) { sx += __shfl_down(sx, o); sy += __shfl_down(sy, o); }
    if (lane == 0) X[t] = cmake(sx, sy);
}

// one work-group per batch entry: sum_lk T[l,k] T[k,l]
__global__ __launch_bounds__(256) void fullg_trace_kernel(const cplx *T, cplx *part, int M) {
    __shared__ double rx[4], ry[4];
    const long b = blockIdx.x;
    const cplx *t = T + b * M * M;
    double sx = 0.0, sy = 0.0;
    for (int e = threadIdx.x; e < M * M; e += 256) {
        const int l = e / M, k = e % M;
        const cplx a = t[e], c = t[k * M + l];
        sx += a.x * c.x - a.y * c.y; sy += a.x * c.y + a.y * c.x;
    }
    for (int o = 32; o > 0; o >>= 1) { sx += __shfl_down(sx, o); sy += __shfl_down(sy, o); }
    if ((threadIdx.x & 63) == 0) { rx[threadIdx.x >> 6] = sx; ry[threadIdx.x >> 6] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) part[b] = cmake(rx[0] + rx[1] + rx[2] + rx[3], ry[0] + ry[1] + ry[2] + ry[3]);
}

// exx[gs] += sum over the chunk's partial traces, in order
__global__ void fullg_exx_add_kernel(const cplx *part, cplx *exx, int ng2, int nc) {
    const int gs = blockIdx.x * blockDim.x + threadIdx.x;
    if (gs >= ng2) return;
    cplx s = exx[gs];
    for (int n = 0; n < nc; ++n) s = cadd(s, part[(long)gs * nc + n]);
    exx[gs] = s;
}

__global__ __launch_bounds__(256) void fullg_finish_kernel(const cplx *G, const cplx *H1, const cplx *X,
                                                           const cplx *exx, cplx *E, int M, int K, double ecore) {
    __shared__ double red[8];
    const int g = blockIdx.x;
    double ex = 0.0, ey = 0.0;
    for (int e = threadIdx.x; e < 2 * M * M; e += 256) {
        const cplx a = H1[e], b = G[(long)g * 2 * M * M + e];
        ex += a.x * b.x - a.y * b.y; ey += a.x * b.y + a.y * b.x;
    }
    double cx = 0.0, cy = 0.0;
    for (int n = threadIdx.x; n < K; n += 256) {
        const cplx xa = X[((long)g * 2) * K + n], xb = X[((long)g * 2 + 1) * K + n];
        const cplx s = cadd(xa, xb);
        cx += s.x * s.x - s.y * s.y; cy += 2.0 * s.x * s.y;            // (Xa + Xb)^2, unconjugated
    }
    double v[4] = {ex, ey, cx, cy};
    for (int c = 0; c < 4; ++c) {
        double t = v[c];
        for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
        __syncthreads();
        v[c] = red[0] + red[1] + red[2] + red[3];
    }
    if (threadIdx.x == 0) {
        const cplx x = cadd(exx[2 * g], exx[2 * g + 1]);
        const cplx e1 = cmake(v[0], v[1]);
        const cplx e2 = cmake(0.5 * (v[2] - x.x), 0.5 * (v[3] - x.y));
        E[3 * g] = cmake(e1.x + e2.x + ecore, e1.y + e2.y);
        E[3 * g + 1] = cmake(e1.x + ecore, e1.y);
        E[3 * g + 2] = e2;
    }
}

namespace {
// scratch device buffer released on every exit path
template <class T> struct Scratch {
    T *p = nullptr;
    ~Scratch() { if (p) hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, n * sizeof(T)); }
};
}  // namespace

// G_dev [ng, 2, M, M] -> E_dev [ng, 3]
int k_energy_full_g(afq_handle *h, const cplx *G_dev, int ng, cplx *E_dev) {
    const int M = h->M, K = h->K, Mp = (M + 1) & ~1;
    if (!h->L_full) {
        AFQ_HIP(h, hipMalloc(&h->L_full, sizeof(double) * (size_t)K * M * Mp));
        const long n = (long)K * M * Mp;
        AFQ_LAUNCH(h, fullg_expand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->hs_pot,
                           h->ld_hs, h->hs_sym ? 1 : 0, M, Mp, K, h->L_full);
        AFQ_POST(h);
    }
    const int ng2 = 2 * ng;
    // chunk of Cholesky vectors: T workspace <= ~1.5 GB
    const size_t per_n = (size_t)ng2 * M * M * sizeof(cplx);
    int nc = (int)std::max<size_t>(1, std::min<size_t>((size_t)K, ((size_t)3 << 29) / per_n));
    Scratch<cplx> sT, spart, sX, sexx;
    AFQ_HIP(h, sT.alloc(per_n / sizeof(cplx) * nc));
    AFQ_HIP(h, spart.alloc((size_t)ng2 * nc));
    AFQ_HIP(h, sX.alloc((size_t)ng2 * K));
    AFQ_HIP(h, sexx.alloc((size_t)ng2));
    cplx *T = sT.p, *part = spart.p, *X = sX.p, *exx = sexx.p;
    AFQ_HIP(h, hipMemsetAsync(exx, 0, sizeof(cplx) * (size_t)ng2, h->stream));
    {
        const long nt = (long)ng2 * K;
        AFQ_LAUNCH(h, fullg_coulomb_kernel, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, h->stream, G_dev,
                           h->L_full, X, ng2, M, Mp, K);
        AFQ_POST(h);
    }
    int rc = AFQ_OK;
    for (int n0 = 0; n0 < K && !rc; n0 += nc) {
        const int cur = std::min(nc, K - n0);
        FullGTProb p;
        p.batch = ng2 * cur; p.rows = M; p.cols = M; p.kdim = M; p.nc = cur; p.n0 = n0; p.M = M; p.Mp = Mp;
        p.G = G_dev; p.L = h->L_full; p.T = T;
        afq_note_launch(h, "FullGTProb GEMM");
        hipError_t e = launch_mfma_gemm_wg<2, 2, 2, 2, 4, FullGTProb, MAP_COLS_FAST>(p, h->stream, h->zero_page);
        if (e == hipSuccess) e = afq_post_launch(h);
        if (e != hipSuccess) { h->err = hipGetErrorString(e); rc = AFQ_EHIP; break; }
        AFQ_LAUNCH(h, fullg_trace_kernel, dim3((unsigned)p.batch), dim3(256), 0, h->stream, T, part, M);
        AFQ_LAUNCH(h, fullg_exx_add_kernel, dim3((ng2 + 63) / 64), dim3(64), 0, h->stream, part, exx, ng2, cur);
        if (hipGetLastError() != hipSuccess) { h->err = "full-G energy launch failed"; rc = AFQ_EHIP; }
    }
    if (!rc) {
        AFQ_LAUNCH(h, fullg_finish_kernel, dim3(ng), dim3(256), 0, h->stream, G_dev, h->H1, X, exx, E_dev, M, K,
                           h->ecore);
        if (hipGetLastError() != hipSuccess) { h->err = "full-G energy launch failed"; rc = AFQ_EHIP; }
    }
    hipStreamSynchronize(h->stream);       // the scratch buffers are released on return
    return rc;
}
