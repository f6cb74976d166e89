// Library-owned communicator: comb population control across ranks and the estimator reduction, entirely on
// the device (walkers/handler.py:225-338 with the Allgather :232, bcast :291 and Isend / Recv :313,322;
// estimators/mixed.py:261,273).
//
// One population-control event on every rank, all queued on the handle's stream, nothing read back:
//   1. comm_prep_kernel      sendw = [ |weight_0| .. |weight_{nw-1}|, r ]      (r: rank 0's comb uniform)
//   2. all-gather            gw[rank][nw + 1]                                   (RCCL ncclAllGather over xGMI)
//   3. comb_plan_global      every rank decides the IDENTICAL global comb from gw: total weight, scaling of
//                            its own weights, teeth located by bisection of the cumulative weights,
//                            zip(clone, kill) pairs in global walker order.  Both lists are ascending, so the
//                            pairs are sorted by (source rank, destination rank): the pairs of one rank pair
//                            form one contiguous run and the position inside the run is the exchange slot.
//                            Out: pairs inside this rank, send / receive slot lists per peer, global parent_ix.
//   4. clone_kernel          copies inside the rank
//   5. comm_pack_kernel      walker state (phi, scalars, cached Green's function, back-propagation history)
//                            of the outgoing walkers into the per-peer slot buffers
//   6. exchange              fixed-capacity all-to-all: `cap` slots to and from every peer in one RCCL group
//                            of ncclSend / ncclRecv (the host never learns the counts, so it posts the
//                            capacity; xGMI is a full mesh, so the 7 transfers of a rank run on 7 links)
//   7. comm_unpack_kernel    incoming walkers into the kill slots, then all weights <- 1
// More pairs between two ranks than `cap` slots raise the sticky flag scal[3] (AFQ_EOVERFLOW at the next
// afq_estimates_get); scal[4] keeps the largest run seen so that the caller can size `cap`.
//
// Two transports: RCCL (one process per GPU; the library resolves librccl at afq_comm_init, so single-GPU users
// never load it) and an in-process communicator of several handles (afq_comm_init_local: one host thread driving
// several GPUs, or several handles on one GPU -- which is also how the multi-rank path is tested on a 1-GPU box):
// the same kernels, device-to-device copies ordered by events instead of RCCL calls.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "block_scan.h"

namespace {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

RcclApi *rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.lib ? &api : nullptr;
    tried = true;
    // a process that already holds RCCL (torch.distributed's copy) resolves to that one through the soname
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
    }
    if (!api.lib) { api.err = std::string("cannot load librccl: ") + dlerror(); return nullptr; }
#define SYM_(field, sym)                                                              \
    api.field = (decltype(api.field))dlsym(api.lib, #sym);                            \
    if (!api.field) { api.err = "librccl lacks " #sym; api.lib = nullptr; return nullptr; }
    SYM_(GetUniqueId, ncclGetUniqueId) SYM_(CommInitRank, ncclCommInitRank) SYM_(CommDestroy, ncclCommDestroy)
    SYM_(AllGather, ncclAllGather) SYM_(AllReduce, ncclAllReduce) SYM_(Send, ncclSend) SYM_(Recv, ncclRecv)
    SYM_(GroupStart, ncclGroupStart) SYM_(GroupEnd, ncclGroupEnd) SYM_(GetErrorString, ncclGetErrorString)
#undef SYM_
    return &api;
}

struct afq_comm_state {
    int rank = 0, nranks = 1;
    bool local = false;                     // in-process communicator (afq_comm_init_local)
    std::vector<afq_handle *> peers;        // local: the handle of every rank
    ncclComm_t nccl = nullptr;
    int cap = 0;                            // walker slots per peer and event (0: default on first use)
    // sized by ensure_buffers
    int nw = 0, nbp = 0;
    size_t slot = 0;                        // cplx elements per slot
    double *sendw = nullptr, *gw = nullptr;
    int *pix = nullptr;                     // [nranks * nw] global parent_ix
    int *lists = nullptr;                   // nsend[R] | nrecv[R] | send_idx[R][cap] | recv_idx[R][cap]
    cplx *sbuf = nullptr, *rbuf = nullptr;  // [R][cap][slot]
    hipEvent_t ev = nullptr, ev2 = nullptr;
    long events = 0;
};

afq_comm_state *cs_of(afq_handle *h) { return (afq_comm_state *)h->comm; }

#define AFQ_NCCL(h, api, call)                                                                   \
    do {                                                                                         \
        ncclResult_t r_ = (call);                                                                \
        if (r_ != ncclSuccess) {                                                                 \
            (h)->err = std::string(#call) + ": " + (api)->GetErrorString(r_);                    \
            return AFQ_EHIP;                                                                     \
        }                                                                                        \
    } while (0)

// ---- slot layout: the walker state that has to travel, in 16-byte units ------------------------------------
// phi | [ghalf] | [phi_old | hist] | [G] | ot ehyb phase eloc | (unscaled, detR) (log_detR, 0) | [ovlp_new] | [bp_ph | (bp_cos, bp_n)]
struct SlotLayout {
    long per, hist_per, gsz;
    int with_greens, with_bp, with_rdm;     // with_rdm: walker.G travels (mixed estimator with one_rdm: the accumulated
                                            // G is whatever the walker carries, estimators/mixed.py:226-229)
    __host__ __device__ long size() const {
        long n = per + 6;
        if (with_greens) n += per + 1;
        if (with_bp) n += per + hist_per + 2;
        if (with_rdm) n += gsz;
        return n;
    }
};

struct PackArgs {
    SlotLayout L;
    int cap, nranks, rank;
    const int *count;        // nsend / nrecv [R]
    const int *idx;          // send_idx / recv_idx [R][cap]
    cplx *buf;               // sbuf / rbuf [R][cap][slot]
    long slot;               // elements per slot (>= L.size(), fixed for the buffers)
    cplx *phi, *ot, *ehyb, *phase, *eloc, *ghalf, *ovlp_new, *phi_old, *bp_hist, *bp_ph, *G;
    double *unscaled, *detR, *log_detR, *bp_cos;
    int *bp_n;
};

template <bool PACK>
__global__ __launch_bounds__(256) void comm_pack_kernel(PackArgs a) {
    const int sl = blockIdx.y, peer = blockIdx.z;
    if (peer == a.rank || sl >= a.count[peer]) return;
    const int w = a.idx[peer * a.cap + sl];
    cplx *s = a.buf + ((long)peer * a.cap + sl) * a.slot;
    const long per = a.L.per;
    const long stride = (long)gridDim.x * blockDim.x, t0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    auto mv = [&](cplx *field, long n, long off) {
        for (long i = t0; i < n; i += stride) {
            if (PACK) s[off + i] = field[(long)w * n + i];
            else field[(long)w * n + i] = s[off + i];
        }
    };
    long off = 0;
    mv(a.phi, per, off); off += per;
    if (a.L.with_greens) { mv(a.ghalf, per, off); off += per; }
    if (a.L.with_bp) { mv(a.phi_old, per, off); off += per; mv(a.bp_hist, a.L.hist_per, off); off += a.L.hist_per; }
    if (a.L.with_rdm) { mv(a.G, a.L.gsz, off); off += a.L.gsz; }
    if (t0 == 0) {
        if (PACK) {
            s[off] = a.ot[w]; s[off + 1] = a.ehyb[w]; s[off + 2] = a.phase[w]; s[off + 3] = a.eloc[w];
            s[off + 4] = cmake(a.unscaled[w], a.detR[w]);
            s[off + 5] = cmake(a.log_detR[w], 0.0);
            long o = off + 6;
            if (a.L.with_greens) s[o++] = a.ovlp_new[w];
            if (a.L.with_bp) { s[o++] = a.bp_ph[w]; s[o++] = cmake(a.bp_cos[w], (double)a.bp_n[w]); }
        } else {
            a.ot[w] = s[off]; a.ehyb[w] = s[off + 1]; a.phase[w] = s[off + 2]; a.eloc[w] = s[off + 3];
            a.unscaled[w] = s[off + 4].x; a.detR[w] = s[off + 4].y;
            a.log_detR[w] = s[off + 5].x;
            long o = off + 6;
            if (a.L.with_greens) a.ovlp_new[w] = s[o++];
            if (a.L.with_bp) { a.bp_ph[w] = s[o++]; a.bp_cos[w] = s[o].x; a.bp_n[w] = (int)s[o].y; ++o; }
        }
    }
}

__global__ void comm_prep_kernel(const double *weight, int nw, double r, double *sendw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nw) sendw[i] = fabs(weight[i]);                     // handler.py:230
    if (i == nw) sendw[nw] = r;                                   // handler.py:276 (rank 0's is the one used)
}

// walkers/handler.py:225-301 for the global population, identically on every rank.  One 256-thread work-group;
// thread t owns a contiguous chunk of the N = R * nw global walkers (sequential sums inside the chunk, prefix
// scans across chunks; the same arithmetic as comb_plan_kernel of the single-rank path).
struct PlanArgs {
    const double *gw;        // [R][nw + 1]
    int R, nw, rank, cap;
    double target;
    double *weight, *unscaled;       // this rank's walkers
    int *pix_global;         // [N]
    int *parent_ix;          // [nw] this rank's slice
    int *pairs;              // local (src, dst) pairs
    int *lists;              // nsend[R] | nrecv[R] | send_idx[R][cap] | recv_idx[R][cap]
    double *scal;
};

__global__ __launch_bounds__(256) void comb_plan_global_kernel(PlanArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.R * a.nw, nw = a.nw, R = a.R;
    double *cs = (double *)smem;
    int *pix = (int *)(cs + N);
    int *clone_l = pix + N, *kill_l = clone_l + N;
    __shared__ double wtot_d[4];
    __shared__ int wtot_i[4];
    __shared__ int s_nloc, s_maxrun, s_over;
    const int tid = threadIdx.x;
    int *nsend = a.lists, *nrecv = a.lists + R, *send_idx = a.lists + 2 * R, *recv_idx = send_idx + R * a.cap;
    if (tid < R) { nsend[tid] = 0; nrecv[tid] = 0; }
    if (tid == 0) { s_nloc = 0; s_maxrun = 0; s_over = 0; }
    const int per = (N + 255) / 256;
    const int i0 = tid * per < N ? tid * per : N, i1 = (tid + 1) * per < N ? (tid + 1) * per : N;
    auto gweight = [&](int g) { return a.gw[(g / nw) * (nw + 1) + g % nw]; };
    const double r = a.gw[nw];                                    // rank 0's uniform (handler.py:291)
    double loc = 0.0;
    for (int i = i0; i < i1; ++i) { const double x = gweight(i); cs[i] = x; pix[i] = 0; loc += x; }
    double total;
    (void)block_excl_scan256(loc, wtot_d, &total);                // sum(global_weights), handler.py:233
    if (tid == 0) a.scal[0] = total;
    if (total < 1e-8) {                                           // handler.py:236-241
        if (tid == 0) { a.scal[1] = -1.0; a.scal[2] = 1.0; }
        return;
    }
    const double scale = total / a.target;
    for (int i = tid; i < nw; i += 256) {                         // handler.py:244-246, this rank's walkers
        a.unscaled[i] = a.weight[i];
        a.weight[i] = a.weight[i] / scale;
    }
    loc = 0.0;
    for (int i = i0; i < i1; ++i) { loc += cs[i] / scale; cs[i] = loc; }     // global_weights / scale, :248
    double tot2;
    const double base = block_excl_scan256(loc, wtot_d, &tot2);
    for (int i = i0; i < i1; ++i) cs[i] += base;                  // numpy.cumsum(weights)
    __syncthreads();
    const int ntarget = (int)a.target;
    const double step = tot2 / a.target;
    for (int ic = tid; ic < ntarget; ic += 256) {
        const double tooth = (ic + r) * step;
        int lo = 0, hi = N;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tooth < cs[mid]) hi = mid; else lo = mid + 1;
        }
        if (lo < N) atomicAdd(&pix[lo], 1);
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
        a.pix_global[i] = pix[i];
        if (i / nw == a.rank) a.parent_ix[i % nw] = pix[i];
    }
    __syncthreads();
    const int np = totc < totk ? totc : totk;                     // zip(clone, kill) truncates, :295-301
    // pair j = (clone_l[j], kill_l[j]); key(j) = src_rank * R + dst_rank is non-decreasing in j
    auto key = [&](int j) { return (clone_l[j] / nw) * R + kill_l[j] / nw; };
    for (int j = tid; j < np; j += 256) {
        const int c = clone_l[j], k = kill_l[j], s = c / nw, d = k / nw, kj = s * R + d;
        int lo = 0, hi = j;                                       // first pair of this (s, d) run
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (key(mid) < kj) lo = mid + 1; else hi = mid;
        }
        const int pos = j - lo;
        if (s == d) {
            if (s == a.rank) { a.pairs[2 * pos] = c % nw; a.pairs[2 * pos + 1] = k % nw; atomicMax(&s_nloc, pos + 1); }
            continue;
        }
        atomicMax(&s_maxrun, pos + 1);
        if (pos >= a.cap) { s_over = 1; continue; }               // every rank sees every overflow
        if (s == a.rank) { send_idx[d * a.cap + pos] = c % nw; atomicMax(&nsend[d], pos + 1); }
        if (d == a.rank) { recv_idx[s * a.cap + pos] = k % nw; atomicMax(&nrecv[s], pos + 1); }
    }
    __syncthreads();
    if (tid == 0) {
        a.scal[1] = (double)s_nloc;                               // pairs for clone_kernel
        if (s_over) a.scal[3] = 1.0;
        if ((double)s_maxrun > a.scal[4]) a.scal[4] = (double)s_maxrun;
        a.scal[5] += 1.0;                                         // events
    }
}

int default_cap(int nw) { return std::max(8, (nw + 7) / 8); }

void free_buffers(afq_comm_state *c) {
    for (void *p : {(void *)c->sendw, (void *)c->gw, (void *)c->pix, (void *)c->lists, (void *)c->sbuf, (void *)c->rbuf})
        if (p) hipFree(p);
    c->sendw = c->gw = nullptr; c->pix = c->lists = nullptr; c->sbuf = c->rbuf = nullptr;
    c->nw = 0;
}

SlotLayout max_layout(afq_handle *h) {
    SlotLayout L;
    L.per = (long)h->M * h->nt; L.hist_per = (long)h->nbp * h->K; L.gsz = 2L * h->M * h->M;
    L.with_greens = 1; L.with_bp = h->nbp > 0; L.with_rdm = (h->rdm_on && h->G) ? 1 : 0;
    return L;
}

int ensure_buffers(afq_handle *h) {
    afq_comm_state *c = cs_of(h);
    if (c->cap <= 0) c->cap = default_cap(h->nw);
    const size_t slot = (size_t)max_layout(h).size();
    if (c->nw == h->nw && c->nbp == h->nbp && c->slot == slot && c->sbuf) return AFQ_OK;
    AFQ_HIP(h, hipStreamSynchronize(h->stream));
    free_buffers(c);
    const size_t R = c->nranks, nw = h->nw;
    AFQ_HIP(h, hipMalloc(&c->sendw, sizeof(double) * (nw + 1)));
    AFQ_HIP(h, hipMalloc(&c->gw, sizeof(double) * R * (nw + 1)));
    AFQ_HIP(h, hipMalloc(&c->pix, sizeof(int) * R * nw));
    AFQ_HIP(h, hipMalloc(&c->lists, sizeof(int) * (2 * R + 2 * R * c->cap)));
    AFQ_HIP(h, hipMalloc(&c->sbuf, sizeof(cplx) * R * c->cap * slot));
    AFQ_HIP(h, hipMalloc(&c->rbuf, sizeof(cplx) * R * c->cap * slot));
    AFQ_HIP(h, hipMemset(c->lists, 0, sizeof(int) * (2 * R + 2 * R * c->cap)));
    c->nw = h->nw; c->nbp = h->nbp; c->slot = slot;
    return AFQ_OK;
}

// stages 1 (prep) / 3-5 (plan, local clones, pack) / 7 (unpack, reset) of one event, queued on h->stream
int stage_prep(afq_handle *h, double r) {
    afq_comm_state *c = cs_of(h);
    int rc = ensure_buffers(h);
    if (rc) return rc;
    AFQ_LAUNCH(h, comm_prep_kernel, dim3((h->nw + 1 + 255) / 256), dim3(256), 0, h->stream, h->weight, h->nw, r, c->sendw);
    AFQ_POST(h);
    return AFQ_OK;
}

void fill_pack(afq_handle *h, PackArgs &p, bool with_greens, bool send) {
    afq_comm_state *c = cs_of(h);
    p.L = max_layout(h); p.L.with_greens = with_greens ? 1 : 0;
    p.cap = c->cap; p.nranks = c->nranks; p.rank = c->rank;
    p.count = c->lists + (send ? 0 : c->nranks);
    p.idx = c->lists + 2 * c->nranks + (send ? 0 : c->nranks * c->cap);
    p.buf = send ? c->sbuf : c->rbuf; p.slot = (long)c->slot;
    p.phi = h->phi; p.ot = h->ot; p.ehyb = h->ehyb; p.phase = h->phase; p.eloc = h->eloc; p.ghalf = h->ghalf;
    p.ovlp_new = h->ovlp_new; p.phi_old = h->phi_old; p.bp_hist = h->bp_hist; p.bp_ph = h->bp_ph; p.G = h->G;
    p.unscaled = h->unscaled; p.detR = h->detR; p.log_detR = h->log_detR; p.bp_cos = h->bp_cos; p.bp_n = h->bp_n;
}

int stage_plan_pack(afq_handle *h, double target, bool with_greens) {
    afq_comm_state *c = cs_of(h);
    const long N = (long)c->nranks * h->nw;
    const size_t lds = (sizeof(double) + 3 * sizeof(int)) * (size_t)N;
    if (lds > 160 * 1024 - 256) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "device comb: more than 8000 walkers in total");
    static size_t lds_set[AFQ_MAX_DEVICES] = {0};
    AFQ_HIP(h, afq_raise_lds((const void *)comb_plan_global_kernel, lds, lds_set));
    PlanArgs a;
    a.gw = c->gw; a.R = c->nranks; a.nw = h->nw; a.rank = c->rank; a.cap = c->cap; a.target = target;
    a.weight = h->weight; a.unscaled = h->unscaled; a.pix_global = c->pix; a.parent_ix = h->parent_ix;
    a.pairs = (int *)h->pack_tmp; a.lists = c->lists; a.scal = h->scal;
    AFQ_LAUNCH(h, comb_plan_global_kernel, dim3(1), dim3(256), lds, h->stream, a);
    AFQ_POST(h);
    int rc = k_clone_pairs(h, with_greens);
    if (rc) return rc;
    if (c->nranks > 1) {
        PackArgs p;
        fill_pack(h, p, with_greens, true);
        AFQ_LAUNCH(h, comm_pack_kernel<true>, dim3(4, c->cap, c->nranks), dim3(256), 0, h->stream, p);
        AFQ_POST(h);
    }
    return AFQ_OK;
}

int stage_unpack(afq_handle *h, bool with_greens) {
    ++h->ghalf_version;                 // cloned / received walkers bring their Ghalf along
    afq_comm_state *c = cs_of(h);
    if (c->nranks > 1) {
        PackArgs p;
        fill_pack(h, p, with_greens, false);
        AFQ_LAUNCH(h, comm_pack_kernel<false>, dim3(4, c->cap, c->nranks), dim3(256), 0, h->stream, p);
        AFQ_POST(h);
    }
    c->events += 1;
    return k_reset_weights(h, true);                              // handler.py:337-338
}

int check_group(afq_handle **hs, int n, std::string *err) {
    if (!hs || n < 1) return AFQ_EINVAL;
    for (int i = 0; i < n; ++i) {
        afq_comm_state *c = hs[i] ? cs_of(hs[i]) : nullptr;
        if (!c || !c->local || c->nranks != n || c->rank != i || c->peers.size() != (size_t)n || c->peers[0] != hs[0]) {
            if (err) *err = "the handles are not the ranks 0..n-1 of one afq_comm_init_local communicator";
            return AFQ_ESTATE;
        }
        if (hs[i]->nw != hs[0]->nw || !hs[i]->nw) { if (err) *err = "every rank needs the same number of walkers"; return AFQ_ESTATE; }
    }
    return AFQ_OK;
}

}  // namespace

int k_comm_size(afq_handle *h) { return h->comm ? cs_of(h)->nranks : 1; }

void k_comm_destroy(afq_handle *h) {
    afq_comm_state *c = cs_of(h);
    if (!c) return;
    free_buffers(c);
    if (c->nccl) { RcclApi *api = rccl_api(); if (api) api->CommDestroy(c->nccl); }
    if (c->ev) hipEventDestroy(c->ev);
    if (c->ev2) hipEventDestroy(c->ev2);
    delete c;
    h->comm = nullptr;
}

// one event on an RCCL communicator (stages 1-7 above)
int k_comm_popcontrol(afq_handle *h, double r, double target, bool with_greens) {
    h->scal_cache_valid = false;
    afq_comm_state *c = cs_of(h);
    if (!c) AFQ_FAIL(h, AFQ_ESTATE, "no communicator");
    if (c->local) AFQ_FAIL(h, AFQ_ESTATE, "in-process communicator: use afq_popcontrol_comb_local for all ranks at once");
    RcclApi *api = rccl_api();
    if (!api) AFQ_FAIL(h, AFQ_ESTATE, "RCCL is not loaded");
    int rc = stage_prep(h, r);
    if (rc) return rc;
    afq_note_launch(h, "ncclAllGather(weights)");
    AFQ_NCCL(h, api, api->AllGather(c->sendw, c->gw, (size_t)h->nw + 1, ncclDouble, c->nccl, h->stream));
    if ((rc = stage_plan_pack(h, target, with_greens))) return rc;
    if (c->nranks > 1) {
        const size_t bytes = sizeof(cplx) * (size_t)c->cap * c->slot;
        afq_note_launch(h, "ncclSend/Recv(walker slots)");
        AFQ_NCCL(h, api, api->GroupStart());
        for (int p = 0; p < c->nranks; ++p) {
            if (p == c->rank) continue;
            AFQ_NCCL(h, api, api->Send(c->sbuf + (size_t)p * c->cap * c->slot, bytes, ncclChar, p, c->nccl, h->stream));
            AFQ_NCCL(h, api, api->Recv(c->rbuf + (size_t)p * c->cap * c->slot, bytes, ncclChar, p, c->nccl, h->stream));
        }
        AFQ_NCCL(h, api, api->GroupEnd());
    }
    return stage_unpack(h, with_greens);
}

extern "C" {

int afq_comm_unique_id(void *id_out) {
    if (!id_out) return AFQ_EINVAL;
    RcclApi *api = rccl_api();
    if (!api) return AFQ_EUNSUPPORTED;
    ncclUniqueId id;
    if (api->GetUniqueId(&id) != ncclSuccess) return AFQ_EHIP;
    static_assert(sizeof(ncclUniqueId) == AFQ_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id_out, &id, sizeof(id));
    return AFQ_OK;
}

int afq_comm_init(afq_handle *h, const void *unique_id, int rank, int nranks) {
    if (!h || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) return AFQ_EINVAL;
    hipSetDevice(h->device);
    RcclApi *api = rccl_api();
    if (!api) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "librccl could not be loaded (dlopen librccl.so.1)");
    k_comm_destroy(h);
    afq_comm_state *c = new afq_comm_state();
    c->rank = rank; c->nranks = nranks;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = api->CommInitRank(&c->nccl, nranks, id, rank);
    if (r != ncclSuccess) {
        h->err = std::string("ncclCommInitRank: ") + api->GetErrorString(r);
        delete c;
        return AFQ_EHIP;
    }
    h->comm = c;
    if (h->nw) {    // walker.total_weight starts as the size of the whole population (walkers/handler.py:164)
        const double tw0 = (double)h->nw * nranks;
        AFQ_HIP(h, hipMemcpy(h->scal, &tw0, sizeof(double), hipMemcpyHostToDevice));
    }
    return AFQ_OK;
}

int afq_comm_init_local(afq_handle **handles, int n) {
    if (!handles || n < 1) return AFQ_EINVAL;
    for (int i = 0; i < n; ++i) if (!handles[i]) return AFQ_EINVAL;
    for (int i = 0; i < n; ++i) {
        afq_handle *h = handles[i];
        hipSetDevice(h->device);
        k_comm_destroy(h);
        afq_comm_state *c = new afq_comm_state();
        c->rank = i; c->nranks = n; c->local = true;
        c->peers.assign(handles, handles + n);
        hipEventCreateWithFlags(&c->ev, hipEventDisableTiming);
        hipEventCreateWithFlags(&c->ev2, hipEventDisableTiming);
        h->comm = c;
        if (h->nw) {
            const double tw0 = (double)h->nw * n;
            AFQ_HIP(h, hipMemcpy(h->scal, &tw0, sizeof(double), hipMemcpyHostToDevice));
        }
    }
    return AFQ_OK;
}

int afq_comm_destroy(afq_handle *h) {
    if (!h) return AFQ_EINVAL;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    k_comm_destroy(h);
    return AFQ_OK;
}

int afq_comm_set_capacity(afq_handle *h, int max_walkers_per_peer) {
    if (!h || max_walkers_per_peer < 1) return AFQ_EINVAL;
    afq_comm_state *c = cs_of(h);
    if (!c) AFQ_FAIL(h, AFQ_ESTATE, "no communicator");
    if (c->cap == max_walkers_per_peer) return AFQ_OK;
    hipSetDevice(h->device);
    AFQ_HIP(h, hipStreamSynchronize(h->stream));
    free_buffers(c);
    c->cap = max_walkers_per_peer;
    return AFQ_OK;
}

int afq_comm_stats(afq_handle *h, int64_t *out) {
    if (!h || !out) return AFQ_EINVAL;
    afq_comm_state *c = cs_of(h);
    if (!c) AFQ_FAIL(h, AFQ_ESTATE, "no communicator");
    hipSetDevice(h->device);
    double sc[8];
    if (h->scal_cache_valid) {
        // right behind the block's afq_estimates_get(_end): the scalars came along with the sums, no synchronisation
        // (the driver reads the statistics at block boundaries: with a sync here the host loses its lead over the device
        //  once per block and the first launches of the next step arrive late)
        memcpy(sc, h->scal_cache, sizeof(sc));
    } else {
        AFQ_HIP(h, hipMemcpyAsync(sc, h->scal, sizeof(sc), hipMemcpyDeviceToHost, h->stream));
        AFQ_HIP(h, hipStreamSynchronize(h->stream));
    }
    out[0] = (int64_t)sc[4];                 // largest number of walkers one rank sent to another in one event
    out[1] = (int64_t)sc[5];                 // events
    out[2] = c->cap > 0 ? c->cap : default_cap(h->nw);
    out[3] = (int64_t)sc[3];                 // overflow flag
    out[4] = c->rank; out[5] = c->nranks;
    return AFQ_OK;
}

int afq_popcontrol_comb_local(afq_handle **hs, int n, double r, double target, int32_t *parent_ix, double *total_out) {
    std::string err;
    int rc = check_group(hs, n, &err);
    if (rc) { if (hs && n > 0 && hs[0]) hs[0]->err = err; return rc; }
    const int nw = hs[0]->nw;
    std::vector<char> keep(n);
    for (int i = 0; i < n; ++i) {
        afq_handle *h = hs[i];
        hipSetDevice(h->device);
        h->scal_cache_valid = false;
        keep[i] = h->greens_valid && h->ndet == 1;
        h->greens_valid = false;
    }
    // a cached Green's function travels only if every rank has one (the slots of a pair must agree)
    bool with_greens = true;
    for (int i = 0; i < n; ++i) with_greens = with_greens && keep[i];
    for (int i = 0; i < n; ++i) {
        hipSetDevice(hs[i]->device);
        if ((rc = stage_prep(hs[i], i == 0 ? r : 0.0))) return rc;
        AFQ_HIP(hs[i], hipEventRecord(cs_of(hs[i])->ev, hs[i]->stream));
    }
    for (int i = 0; i < n; ++i) {                                    // all-gather
        afq_handle *h = hs[i];
        hipSetDevice(h->device);
        for (int j = 0; j < n; ++j) {
            AFQ_HIP(h, hipStreamWaitEvent(h->stream, cs_of(hs[j])->ev, 0));
            AFQ_HIP(h, hipMemcpyAsync(cs_of(h)->gw + (size_t)j * (nw + 1), cs_of(hs[j])->sendw, sizeof(double) * (nw + 1),
                                      hipMemcpyDeviceToDevice, h->stream));
        }
    }
    for (int i = 0; i < n; ++i) {
        hipSetDevice(hs[i]->device);
        if ((rc = stage_plan_pack(hs[i], target, with_greens))) return rc;
        AFQ_HIP(hs[i], hipEventRecord(cs_of(hs[i])->ev2, hs[i]->stream));
    }
    for (int i = 0; i < n; ++i) {                                    // all-to-all of the slot buffers
        afq_handle *h = hs[i];
        afq_comm_state *c = cs_of(h);
        hipSetDevice(h->device);
        const size_t chunk = (size_t)c->cap * c->slot;
        for (int j = 0; j < n; ++j) {
            if (j == i) continue;
            if (cs_of(hs[j])->cap != c->cap || cs_of(hs[j])->slot != c->slot) AFQ_FAIL(h, AFQ_ESTATE, "ranks disagree on the slot layout");
            AFQ_HIP(h, hipStreamWaitEvent(h->stream, cs_of(hs[j])->ev2, 0));
            AFQ_HIP(h, hipMemcpyAsync(c->rbuf + j * chunk, cs_of(hs[j])->sbuf + i * chunk, sizeof(cplx) * chunk,
                                      hipMemcpyDeviceToDevice, h->stream));
        }
        AFQ_HIP(h, hipEventRecord(c->ev, h->stream));                // this rank's reads of the others' buffers are done
    }
    for (int i = 0; i < n; ++i) {
        afq_handle *h = hs[i];
        hipSetDevice(h->device);
        for (int j = 0; j < n; ++j)
            if (j != i) AFQ_HIP(h, hipStreamWaitEvent(h->stream, cs_of(hs[j])->ev, 0));
        if ((rc = stage_unpack(h, with_greens))) return rc;
        h->greens_valid = with_greens;
    }
    if (!parent_ix && !total_out) return AFQ_OK;
    afq_handle *h0 = hs[0];
    hipSetDevice(h0->device);
    double sc[4];
    AFQ_HIP(h0, hipMemcpyAsync(sc, h0->scal, sizeof(sc), hipMemcpyDeviceToHost, h0->stream));
    if (parent_ix)
        AFQ_HIP(h0, hipMemcpyAsync(parent_ix, cs_of(h0)->pix, sizeof(int) * (size_t)n * nw, hipMemcpyDeviceToHost, h0->stream));
    for (int i = 0; i < n; ++i) { hipSetDevice(hs[i]->device); AFQ_HIP(hs[i], hipStreamSynchronize(hs[i]->stream)); }
    if (total_out) *total_out = sc[0];
    if (sc[1] < 0) AFQ_FAIL(h0, AFQ_EWEIGHT, "total walker weight below 1e-8");
    if (sc[3] != 0.0) AFQ_FAIL(h0, AFQ_EOVERFLOW, "more walkers moved between two ranks than the exchange slots hold");
    return AFQ_OK;
}

int afq_comm_parent_ix(afq_handle *h, int32_t *parent_ix_global) {
    if (!h || !parent_ix_global) return AFQ_EINVAL;
    afq_comm_state *c = cs_of(h);
    if (!c || !c->pix) AFQ_FAIL(h, AFQ_ESTATE, "no population control on a communicator yet");
    hipSetDevice(h->device);
    AFQ_HIP(h, hipMemcpyAsync(parent_ix_global, c->pix, sizeof(int) * (size_t)c->nranks * h->nw, hipMemcpyDeviceToHost, h->stream));
    AFQ_HIP(h, hipStreamSynchronize(h->stream));
    return AFQ_OK;
}

int afq_estimates_allreduce(afq_handle *h, double *buf, int nest) {
    if (!h || (buf && nest < 1)) return AFQ_EINVAL;
    afq_comm_state *c = cs_of(h);
    if (!c) AFQ_FAIL(h, AFQ_ESTATE, "no communicator");
    if (c->local) AFQ_FAIL(h, AFQ_ESTATE, "in-process communicator: use afq_estimates_allreduce_local");
    RcclApi *api = rccl_api();
    if (!api) AFQ_FAIL(h, AFQ_ESTATE, "RCCL is not loaded");
    hipSetDevice(h->device);
    if (!buf) {     // the device accumulators of afq_estimates_update, in place: no host round trip
        { const int rc = k_estimates(h, 0, true); if (rc) return rc; }     // (sums still in the per-walker accumulators)
        afq_note_launch(h, "ncclAllReduce(estimates)");
        AFQ_NCCL(h, api, api->AllReduce(h->estimates, h->estimates, 2 * AFQ_EST_COUNT_, ncclDouble, ncclSum, c->nccl, h->stream));
        if (h->rdm_on && h->rdm_acc)    // the one-body RDM sums are part of the same reduction in the reference (mixed.py:261)
            AFQ_NCCL(h, api, api->AllReduce(h->rdm_acc, h->rdm_acc, (size_t)2 * h->M * h->M, ncclDouble, ncclSum, c->nccl, h->stream));
        return AFQ_OK;
    }
    double *tmp = nullptr;
    AFQ_HIP(h, hipMalloc(&tmp, sizeof(double) * 2 * (size_t)nest));
    hipError_t e = hipMemcpyAsync(tmp, buf, sizeof(double) * 2 * (size_t)nest, hipMemcpyHostToDevice, h->stream);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = api->AllReduce(tmp, tmp, 2 * (size_t)nest, ncclDouble, ncclSum, c->nccl, h->stream);
    if (e == hipSuccess && r == ncclSuccess)
        e = hipMemcpyAsync(buf, tmp, sizeof(double) * 2 * (size_t)nest, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(tmp);
    if (r != ncclSuccess) { h->err = std::string("ncclAllReduce: ") + api->GetErrorString(r); return AFQ_EHIP; }
    AFQ_HIP(h, e);
    return AFQ_OK;
}

int afq_estimates_allreduce_local(afq_handle **hs, int n) {
    std::string err;
    int rc = check_group(hs, n, &err);
    if (rc) { if (hs && n > 0 && hs[0]) hs[0]->err = err; return rc; }
    std::vector<double> sum(2 * AFQ_EST_COUNT_, 0.0), one(2 * AFQ_EST_COUNT_);
    for (int i = 0; i < n; ++i) {
        hipSetDevice(hs[i]->device);
        AFQ_HIP(hs[i], hipMemcpyAsync(one.data(), hs[i]->estimates, sizeof(double) * one.size(), hipMemcpyDeviceToHost, hs[i]->stream));
        AFQ_HIP(hs[i], hipStreamSynchronize(hs[i]->stream));
        for (size_t k = 0; k < one.size(); ++k) sum[k] += one[k];
    }
    for (int i = 0; i < n; ++i) {
        hipSetDevice(hs[i]->device);
        AFQ_HIP(hs[i], hipMemcpyAsync(hs[i]->estimates, sum.data(), sizeof(double) * sum.size(), hipMemcpyHostToDevice, hs[i]->stream));
        AFQ_HIP(hs[i], hipStreamSynchronize(hs[i]->stream));
    }
    if (hs[0]->rdm_on && hs[0]->rdm_acc) {
        const size_t m = (size_t)2 * hs[0]->M * hs[0]->M;
        std::vector<double> rs(m, 0.0), ro(m);
        for (int i = 0; i < n; ++i) {
            if (!hs[i]->rdm_acc) AFQ_FAIL(hs[0], AFQ_ESTATE, "one_rdm switched on for some ranks only");
            hipSetDevice(hs[i]->device);
            AFQ_HIP(hs[i], hipMemcpy(ro.data(), hs[i]->rdm_acc, sizeof(double) * m, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < m; ++k) rs[k] += ro[k];
        }
        for (int i = 0; i < n; ++i) {
            hipSetDevice(hs[i]->device);
            AFQ_HIP(hs[i], hipMemcpy(hs[i]->rdm_acc, rs.data(), sizeof(double) * m, hipMemcpyHostToDevice));
        }
    }
    return AFQ_OK;
}

}  // extern "C"
