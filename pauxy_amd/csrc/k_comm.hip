// Library-owned communicator: comb population control across ranks and the estimator reduction, entirely on
// the device (walkers/handler.py:225-338 with the Allgather :232, bcast :291 and Isend / Recv :313,322;
// estimators/mixed.py:261,273).
//
// One population-control event on every rank, all queued on the handle's stream, nothing read back:
//   1. comm_prep_kernel      sendw = [ |weight_0| .. |weight_{nw-1}|, r ]      (r: rank 0's comb uniform)
//   2. all-gather            gw[rank][nw + 1]                                   (RCCL ncclAllGather over xGMI, or peer
//                                                                                writes into the mapped windows)
//   3. comb_plan_global      every rank decides the IDENTICAL global comb from gw: total weight, scaling of
//                            its own weights, teeth located by bisection of the cumulative weights,
//                            zip(clone, kill) pairs in global walker order.  Both lists are ascending, so the
//                            pairs are sorted by (source rank, destination rank): the pairs of one rank pair
//                            form one contiguous run and the position inside the run is the exchange slot.
//                            Out: pairs inside this rank, send / receive slot lists per peer, global parent_ix.
//   4. clone_kernel          copies inside the rank
//   5. comm_pack_kernel      walker state (phi, scalars, cached Green's function, back-propagation history)
//                            of the outgoing walkers
//   6. exchange              WINDOW transport (default): every rank maps every peer's receive window once
//                            (hipIpcGetMemHandle / hipIpcOpenMemHandle; xGMI peer access) and the pack kernel
//                            writes ONLY THE LIVE SLOTS straight into the destination rank's window, then raises
//                            that rank's flag for this event (system-scope release); the unpack kernel of the
//                            destination waits for the flags of the peers it expects walkers from.  Bytes moved =
//                            live slots x slot size; the host never learns the counts and posts nothing.
//                            SENDRECV transport (fallback): `cap` slots to and from every peer in one RCCL group of
//                            ncclSend / ncclRecv, fixed size because the host does not know the counts.
//   7. comm_unpack_kernel    incoming walkers into the kill slots, then all weights <- 1
// Ordering without extra barriers: the all-gather of event e completes on a rank only after every rank has
// contributed, and a rank contributes in stream order AFTER its unpack of event e - 1 -- so when a pack kernel of
// event e writes into a peer's window, that peer is done reading the slots of event e - 1.  Flags carry the event
// number (monotonic), a waiting kernel gives up after the wait budget (afq_comm_set_timeout, 300 s by default) and raises the sticky error scal[6] (AFQ_ECOMM at the next
// host synchronisation) instead of hanging the device.
// More pairs between two ranks than `cap` slots raise the sticky flag scal[3] (AFQ_EOVERFLOW at the next
// afq_estimates_get); scal[4] keeps the largest run seen.  Up to 512 walkers per rank the window transport sizes
// cap = nw (a rank owns nw walkers: it can neither send nor receive more) and cannot overflow; above, cap =
// max(512, nw / 4) slots per peer (default_cap) with the overflow flag as the guard, and the driver grows the windows at
// block boundaries when the largest transfer seen comes near the capacity (Walkers.tune_exchange_capacity).
//
// Three communicators over the same kernels:
//   RCCL   (afq_comm_init: one process per GPU) -- ncclAllGather / ncclAllReduce for the two collectives, windows
//          (or ncclSend / ncclRecv) for the walkers; the library resolves librccl at afq_comm_init, so single-GPU
//          users never load it;
//   IPC    (afq_comm_init_ipc: one process per GPU, no RCCL) -- everything through the mapped windows; the caller
//          lends an all-gather of a few hundred bytes (its MPI / torch.distributed communicator) as bootstrap;
//   LOCAL  (afq_comm_init_local: one host thread driving several handles, on several GPUs or on one -- how the
//          multi-rank path is tested on a 1-GPU box) -- the windows are plain device pointers of the same process.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "block_scan.h"

namespace {

constexpr int MAX_RANKS = 16;
// How long a kernel waits for a peer's flag before it gives up and raises the sticky error (never a hung device).  The
// budget has to cover everything that can legitimately delay ONE rank between two collectives -- rank 0 writing HDF5
// output or restart files, a file-system stall, first-use code-object loading -- so it is minutes, not seconds, and the
// caller can set it (afq_comm_set_timeout, AFQ_COMM_TIMEOUT_S).  wall_clock64 runs at 100 MHz.
constexpr double WAIT_SECONDS_DEFAULT = 300.0;
__device__ unsigned long long g_wait_ticks = (unsigned long long)(WAIT_SECONDS_DEFAULT * 1e8);

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

enum CommMode { COMM_LOCAL = 0, COMM_RCCL = 1, COMM_IPC = 2 };

// One rank's window: memory every peer maps and writes into.  All offsets in bytes from the window base.
//   flag_x[R]  walker exchange: flag_x[s] = number of the last event whose slots from rank s have landed
//   flag_g[R]  weights all-gather, flag_e[R]  estimator all-gather (window collectives: IPC / LOCAL communicators)
//   gw[2][R][nw + 1], est[2][R][est_n]  double-buffered by event parity: a rank can run at most one event ahead of a
//   peer (its next plan / sum kernel waits for that peer's flag of the next event), never two
//   rbuf[R][cap][slot]  slot (s, k) = k-th walker rank s sends here in the current event
struct WinLayout {
    long off_flag_x, off_flag_g, off_flag_e, off_gw, off_est, off_rbuf, bytes;
    long gw_n, est_n;          // doubles per rank in gw (nw + 1) and est
    int R, cap;
    long slot;                 // cplx elements per slot
    __host__ __device__ unsigned long long *flag_x(void *w) const { return (unsigned long long *)((char *)w + off_flag_x); }
    __host__ __device__ unsigned long long *flag_g(void *w) const { return (unsigned long long *)((char *)w + off_flag_g); }
    __host__ __device__ unsigned long long *flag_e(void *w) const { return (unsigned long long *)((char *)w + off_flag_e); }
    __host__ __device__ double *gw(void *w, int par) const { return (double *)((char *)w + off_gw) + (long)par * R * gw_n; }
    __host__ __device__ double *est(void *w, int par) const { return (double *)((char *)w + off_est) + (long)par * R * est_n; }
    __host__ __device__ cplx *rbuf(void *w) const { return (cplx *)((char *)w + off_rbuf); }
};

WinLayout make_layout(int R, int nw, int cap, long slot, long est_n) {
    WinLayout L;
    L.R = R; L.cap = cap; L.slot = slot; L.gw_n = nw + 1; L.est_n = est_n;
    long o = 0;
    auto take = [&](long bytes) { const long at = o; o += (bytes + 255) & ~255L; return at; };
    L.off_flag_x = take(8L * R); L.off_flag_g = take(8L * R); L.off_flag_e = take(8L * R);
    L.off_gw = take(8L * 2 * R * L.gw_n);
    L.off_est = take(8L * 2 * R * est_n);
    L.off_rbuf = take(16L * R * cap * slot);
    L.bytes = o;
    return L;
}

struct PeerWindows { void *base[MAX_RANKS]; };      // every rank's window as this rank addresses it (own included)

struct afq_comm_state {
    int rank = 0, nranks = 1;
    CommMode mode = COMM_LOCAL;
    std::vector<afq_handle *> peers;        // LOCAL: the handle of every rank
    ncclComm_t nccl = nullptr;              // RCCL
    afq_allgather_fn boot = nullptr;        // IPC: the caller's all-gather (bootstrap of the window handles)
    void *boot_user = nullptr;
    bool window = true;                     // walkers travel by peer writes into mapped windows (else ncclSend / ncclRecv)
    bool win_collectives = true;            // the two collectives go through the windows as well (IPC, LOCAL)
    int cap = 0;                            // walker slots per peer and event (0: default on first use)
    // sized by ensure_buffers
    int nw = 0, nbp = 0;
    size_t slot = 0;                        // cplx elements per slot
    double *sendw = nullptr, *gw = nullptr; // RCCL all-gather staging / result
    int *pix = nullptr;                     // [nranks * nw] global parent_ix
    int *lists = nullptr;                   // nsend[R] | nrecv[R] | send_idx[R][cap] | recv_idx[R][cap]
    cplx *sbuf = nullptr, *rbuf = nullptr;  // SENDRECV transport: [R][cap][slot]
    // window transport
    void *win = nullptr;
    int win_kind = 0;                       // 1 uncached, 2 fine-grained, 3 plain device memory
    WinLayout wl;
    PeerWindows pw;
    bool peer_opened[MAX_RANKS] = {false};
    int *tickets = nullptr;                 // [R] pack blocks done per peer (device)
    unsigned long long seq_x = 0, seq_g = 0, seq_e = 0;     // events so far (identical on every rank)
    hipEvent_t ev = nullptr, ev2 = nullptr;
    long events = 0;
    unsigned long long sendrecv_bytes = 0;  // SENDRECV transport: bytes posted so far (empty slots included)
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

// ---- flags between GPUs ---------------------------------------------------------------------------------------
__device__ inline void flag_release(unsigned long long *flag, unsigned long long v) {
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// waits until *flag >= v; false after the wait budget (the peer never wrote: error, not a hang)
__device__ inline bool flag_wait(const unsigned long long *flag, unsigned long long v) {
    const unsigned long long t0 = wall_clock64();
    const unsigned long long budget = g_wait_ticks;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > budget) return false;
    }
    return true;
}

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
    cplx *buf;               // SENDRECV: sbuf / rbuf [R][cap][slot]; WINDOW unpack: this rank's rbuf
    long slot;               // elements per slot (>= L.size(), fixed for the buffers)
    // WINDOW transport
    int window;
    PeerWindows pw;          // pack: where peer p's window is mapped
    WinLayout wl;
    unsigned long long seq;  // number of this event
    int *tickets;            // pack: blocks done per peer
    double *scal;            // scal[6]: sticky communication error; scal[7], scal[8]: walkers / bytes sent
    cplx *phi, *ot, *ehyb, *phase, *eloc, *ghalf, *ovlp_new, *phi_old, *bp_hist, *bp_ph, *G;
    double *unscaled, *detR, *log_detR, *bp_cos;
    int *bp_n;
    // unpack: a walker that arrives with a cached Ghalf whose spin blocks differ raises closed_bad to this rank's current
    // epoch (afq_internal.h: the closed-shell verdict of the population must cover what other ranks send); closed_half =
    // na * M elements per spin block, 0 = no check
    unsigned long long *closed_bad;
    unsigned long long closed_epoch;
    long closed_half;
};

// grid (4, min(cap, COMM_GRID_Y), R): block (x, y, peer) moves quarter x of the slots y, y + gridDim.y, ... of that peer.
// (one block row per slot of the capacity would launch thousands of work-groups that find nothing to do: the capacity
//  is what a rank COULD send, a comb moves a handful)
#define COMM_GRID_Y 32
template <bool PACK>
__device__ __forceinline__ void comm_pack_body(const PackArgs &a, const int peer) {
    if (peer == a.rank) return;
    const int count = a.count[peer];
    if ((int)blockIdx.y >= count) return;                         // (work-group uniform)
    if (a.window && !PACK) {
        // the slots of this peer are complete once it has raised its flag for this event
        __shared__ int ok;
        if (threadIdx.x == 0) ok = flag_wait(a.wl.flag_x(a.pw.base[a.rank]) + peer, a.seq) ? 1 : 0;
        __syncthreads();
        if (!ok) { if (threadIdx.x == 0 && blockIdx.x == 0) a.scal[6] = 1.0; return; }
    }
    const long per = a.L.per;
    const long stride = (long)gridDim.x * blockDim.x, t0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (int sl = blockIdx.y; sl < count; sl += gridDim.y) {
        const int w = a.idx[peer * a.cap + sl];
        cplx *s;
        // window pack: straight into the window of the destination rank, row of this rank; window unpack: this rank's window,
        // row of the source; send / receive buffers otherwise
        if (a.window && PACK) s = a.wl.rbuf(a.pw.base[peer]) + ((long)a.rank * a.cap + sl) * a.slot;
        else s = a.buf + ((long)peer * a.cap + sl) * a.slot;
        auto mv = [&](cplx *field, long n, long off) {
            for (long i = t0; i < n; i += stride) {
                if (PACK) s[off + i] = field[(long)w * n + i];
                else field[(long)w * n + i] = s[off + i];
            }
        };
        long off = 0;
        mv(a.phi, per, off); off += per;
        if (a.L.with_greens) {
            mv(a.ghalf, per, off);
            if (!PACK && a.closed_half) {
                bool differ = false;
                for (long i = t0; i < a.closed_half; i += stride) {
                    const cplx x = s[off + i], y = s[off + a.closed_half + i];
                    differ |= __double_as_longlong(x.x) != __double_as_longlong(y.x) || __double_as_longlong(x.y) != __double_as_longlong(y.y);
                }
                if (differ) atomicMax(a.closed_bad, a.closed_epoch);
            }
            off += per;
        }
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
                if (a.window) {     // traffic statistics: walkers / bytes this rank has written into peer windows
                    atomicAdd(a.scal + 7, 1.0);
                    atomicAdd(a.scal + 8, (double)(a.L.size() * (long)sizeof(cplx)));
                }
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
    if (PACK && a.window) {
        // every block that had slots of this peer takes a ticket once its writes are out; the last one raises the peer's flag
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const int rows = count < (int)gridDim.y ? count : (int)gridDim.y;
            if (atomicAdd(a.tickets + peer, 1) == (int)gridDim.x * rows - 1) {
                a.tickets[peer] = 0;
                __threadfence_system();
                flag_release(a.wl.flag_x(a.pw.base[peer]) + a.rank, a.seq);
            }
        }
    }
}
template <bool PACK>
__global__ __launch_bounds__(256) void comm_pack_kernel(PackArgs a) { comm_pack_body<PACK>(a, (int)blockIdx.z); }
// Window transport, one process per rank (round 6): pack and unpack of one event in ONE launch, grid (4, y, 2 R): the blocks
// z < R write this rank's live slots into the peers' windows and raise their flags, the blocks z >= R wait for the peers'
// flags and move what arrived into the kill slots.  The pack blocks never wait and are dispatched first; an unpack block
// waits for ANOTHER rank's pack blocks only.  (The in-process communicator of one host thread keeps the two launches: there
// a rank's pack may sit behind the waiting kernel in the same hardware queue.)
__global__ __launch_bounds__(256) void comm_exchange_kernel(PackArgs ps, PackArgs pr) {
    if ((int)blockIdx.z < ps.nranks) comm_pack_body<true>(ps, (int)blockIdx.z);
    else comm_pack_body<false>(pr, (int)blockIdx.z - pr.nranks);
}

__device__ __forceinline__ void comm_prep_window(const double *weight, int nw, double r, const PeerWindows &pw, const WinLayout &wl,
                                                 int rank, unsigned long long seq, int peer) {
    double *dst = wl.gw(pw.base[peer], (int)(seq & 1)) + (long)rank * wl.gw_n;
    for (int i = threadIdx.x; i <= nw; i += blockDim.x) dst[i] = i < nw ? fabs(weight[i]) : r;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) flag_release(wl.flag_g(pw.base[peer]) + rank, seq);
}

// |weights| and rank 0's uniform: into the staging buffer of the RCCL all-gather, or (window collectives) straight into
// row `rank` of every peer's gw of this event's parity, one block per peer, followed by that peer's flag
__global__ __launch_bounds__(256) void comm_prep_kernel(const double *weight, int nw, double r, double *sendw,
                                                        int window, PeerWindows pw, WinLayout wl, int rank,
                                                        unsigned long long seq) {
    if (!window) {
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i < nw) sendw[i] = fabs(weight[i]);                   // handler.py:230
        if (i == nw) sendw[nw] = r;                                 // handler.py:276 (rank 0's is the one used)
        return;
    }
    comm_prep_window(weight, nw, r, pw, wl, rank, seq, (int)blockIdx.x);
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
    const unsigned long long *gflag;     // window all-gather: flag_g[R] of this rank's window (else null)
    unsigned long long seq;
    // window all-gather with one process per rank (round 6): the launch carries R more blocks, block 1 + p writes this rank's
    // |weights| and uniform into peer p's window (comm_prep_kernel's work); block 0 -- the plan -- waits for every rank's row
    // as before.  Nothing in this launch writes the weights (the division by the scale, handler.py:246, is overwritten by the
    // reset to 1 of handler.py:337-338 before anything reads it), so the prep blocks may read them at any time.
    int prep_R;
    double prep_r;
    PeerWindows pw;
    WinLayout wl;
};

__global__ __launch_bounds__(256) void comb_plan_global_kernel(PlanArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (blockIdx.x > 0) {                                          // (only in launches with prep_R blocks behind the plan)
        comm_prep_window(a.weight, a.nw, a.prep_r, a.pw, a.wl, a.rank, a.seq, (int)blockIdx.x - 1);
        return;
    }
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
    if (a.gflag) {
        // window all-gather: row s of gw is complete once rank s has raised its flag for this event
        __shared__ int s_ok;
        if (tid == 0) s_ok = 1;
        __syncthreads();
        if (tid < R && !flag_wait(a.gflag + tid, a.seq)) s_ok = 0;
        __syncthreads();
        if (!s_ok) {          // a peer never arrived: no clones, no transfers, sticky error for the next host synchronisation
            if (tid == 0) { a.scal[1] = 0.0; a.scal[6] = 1.0; }
            return;
        }
    }
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
    // handler.py:244-246, this rank's walkers.  (w.weight / scale is not stored: handler.py:337-338 sets every weight to 1
    // before anything reads it -- clone and reset follow in the next launch -- and the prep blocks of a fused launch read the
    // weights concurrently)
    for (int i = tid; i < nw; i += 256) a.unscaled[i] = a.weight[i];
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

// the window transport moves only live slots, so its capacity is the bound that cannot overflow: a rank owns nw walkers,
// it can neither send nor receive more in one event.  The fixed-size ncclSend / ncclRecv transport pays for every slot
// of the capacity on every link, so it starts at the same safe bound and Walkers.tune_exchange_capacity shrinks it once
// there is history.
// Above 512 walkers per rank the windows are sized for a quarter of the population per peer (a comb moves a handful of
// walkers; nw slots per peer would be 330 MB per rank at the bench shape and several GB at 2048 walkers per rank), with the
// overflow flag scal[3] as the guard; afq_comm_set_capacity overrides.
int default_cap(int nw) { return nw <= 512 ? std::max(1, nw) : std::max(512, nw / 4); }

void close_peer_windows(afq_comm_state *c) {
    for (int p = 0; p < MAX_RANKS; ++p) {
        if (c->peer_opened[p] && c->pw.base[p]) hipIpcCloseMemHandle(c->pw.base[p]);
        c->peer_opened[p] = false; c->pw.base[p] = nullptr;
    }
}

void free_buffers(afq_comm_state *c) {
    close_peer_windows(c);
    for (void *p : {(void *)c->sendw, (void *)c->gw, (void *)c->pix, (void *)c->lists, (void *)c->sbuf, (void *)c->rbuf,
                    (void *)c->win, (void *)c->tickets})
        if (p) hipFree(p);
    c->sendw = c->gw = nullptr; c->pix = c->lists = c->tickets = nullptr; c->sbuf = c->rbuf = nullptr;
    c->win = nullptr;
    c->nw = 0;
}

SlotLayout max_layout(afq_handle *h) {
    SlotLayout L;
    L.per = (long)h->M * h->nt; L.hist_per = (long)h->nbp * h->K; L.gsz = 2L * h->M * h->M;
    L.with_greens = 1; L.with_bp = h->nbp > 0; L.with_rdm = (h->rdm_on && h->G) ? 1 : 0;
    return L;
}

bool uses_windows(const afq_comm_state *c) { return c->window || c->win_collectives; }

// Window memory must be coherent between GPUs while kernels run: uncached (what RCCL itself allocates for its peer
// buffers on gfx942 / gfx950), else fine-grained, else plain device memory (the system-scope release / acquire pairs
// around every hand-over still apply; afq_comm_probe checks the outcome either way).
int alloc_window(afq_handle *h, afq_comm_state *c, size_t bytes) {
    void *p = nullptr;
    c->win_kind = 0;
#ifdef hipDeviceMallocUncached
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) == hipSuccess) c->win_kind = 1;
#endif
    if (!c->win_kind) {
        (void)hipGetLastError();
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) == hipSuccess) c->win_kind = 2;
    }
    if (!c->win_kind) {
        (void)hipGetLastError();
        AFQ_HIP(h, hipMalloc(&p, bytes));
        c->win_kind = 3;
    }
    AFQ_HIP(h, hipMemset(p, 0, bytes));
    // the zeroes must be in memory before the handle is published: a peer's first flag write is ordered against this
    // memset by nothing but the host bootstrap that follows
    AFQ_HIP(h, hipDeviceSynchronize());
    c->win = p;
    return AFQ_OK;
}

struct WinBlob { int ok; int kind; unsigned long long bytes; hipIpcMemHandle_t handle; };

// all-gather of `bytes` per rank on the host, over whatever the communicator has (collective; every stream idle)
int host_allgather(afq_handle *h, const void *send, void *recv, int bytes) {
    afq_comm_state *c = cs_of(h);
    if (c->mode == COMM_IPC) {
        if (c->boot(send, recv, bytes, c->boot_user) != 0) AFQ_FAIL(h, AFQ_ECOMM, "the caller's bootstrap all-gather failed");
        return AFQ_OK;
    }
    RcclApi *api = rccl_api();
    if (!api) AFQ_FAIL(h, AFQ_ESTATE, "RCCL is not loaded");
    char *d = nullptr;
    AFQ_HIP(h, hipMalloc(&d, (size_t)bytes * (c->nranks + 1)));
    hipError_t e = hipMemcpyAsync(d, send, bytes, hipMemcpyHostToDevice, h->stream);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = api->AllGather(d, d + bytes, (size_t)bytes, ncclChar, c->nccl, h->stream);
    if (e == hipSuccess && r == ncclSuccess)
        e = hipMemcpyAsync(recv, d + bytes, (size_t)bytes * c->nranks, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d);
    if (r != ncclSuccess) { h->err = std::string("ncclAllGather(bootstrap): ") + api->GetErrorString(r); return AFQ_EHIP; }
    AFQ_HIP(h, e);
    return AFQ_OK;
}

// (re)allocates this rank's window and maps every peer's.  Collective for the RCCL / IPC communicators (the handles
// travel through host_allgather, which also is the barrier behind which the old windows may go); the LOCAL group call
// fills the tables itself.  A rank that cannot export or map a window makes EVERY rank fall back (RCCL: fixed-size
// ncclSend / ncclRecv) or fail (IPC) -- decided from the gathered blobs, identically everywhere.
int exchange_windows(afq_handle *h) {
    afq_comm_state *c = cs_of(h);
    const int R = c->nranks;
    int rc = alloc_window(h, c, (size_t)c->wl.bytes);
    WinBlob mine;
    memset(&mine, 0, sizeof(mine));
    mine.ok = rc == AFQ_OK; mine.kind = c->win_kind; mine.bytes = (unsigned long long)c->wl.bytes;
    if (mine.ok && c->mode != COMM_LOCAL && R > 1 && hipIpcGetMemHandle(&mine.handle, c->win) != hipSuccess) {
        (void)hipGetLastError();
        // uncached / fine-grained memory that cannot be exported: try plain device memory once
        hipFree(c->win); c->win = nullptr;
        void *p = nullptr;
        if (hipMalloc(&p, (size_t)c->wl.bytes) == hipSuccess && hipMemset(p, 0, (size_t)c->wl.bytes) == hipSuccess &&
            hipDeviceSynchronize() == hipSuccess && hipIpcGetMemHandle(&mine.handle, p) == hipSuccess) { c->win = p; c->win_kind = 3; mine.kind = 3; }
        else { (void)hipGetLastError(); if (p) hipFree(p); mine.ok = 0; }
    }
    for (int p = 0; p < MAX_RANKS; ++p) { c->pw.base[p] = nullptr; c->peer_opened[p] = false; }
    c->pw.base[c->rank] = c->win;
    if (c->mode == COMM_LOCAL || R == 1) return rc;
    std::vector<WinBlob> all(R);
    if ((rc = host_allgather(h, &mine, all.data(), (int)sizeof(WinBlob)))) return rc;
    bool ok = true;
    for (int p = 0; p < R; ++p) ok = ok && all[p].ok && all[p].bytes == mine.bytes;
    int mapped = 1;
    if (ok) {
        for (int p = 0; p < R && mapped; ++p) {
            if (p == c->rank) continue;
            void *q = nullptr;
            if (hipIpcOpenMemHandle(&q, all[p].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); mapped = 0; break; }
            c->pw.base[p] = q; c->peer_opened[p] = true;
        }
    }
    // second round: did every rank map every window?  (also the barrier after which peers may write here)
    std::vector<int> flags(R);
    const int my = ok && mapped;
    if ((rc = host_allgather(h, &my, flags.data(), (int)sizeof(int)))) return rc;
    for (int p = 0; p < R; ++p) ok = ok && flags[p];
    if (ok) return AFQ_OK;
    close_peer_windows(c);
    c->pw.base[c->rank] = c->win;
    if (c->mode == COMM_RCCL) { c->window = false; return AFQ_OK; }     // every rank takes the ncclSend / ncclRecv transport
    AFQ_FAIL(h, AFQ_EUNSUPPORTED, "peer windows could not be exported / mapped on every rank (hipIpcGetMemHandle / hipIpcOpenMemHandle)");
}

long est_window_doubles(afq_handle *h) { return 2L * AFQ_EST_COUNT_ + 2L * h->M * h->M; }

// buffers of the current (nw, nbp, slot, cap); collective when windows have to be (re)made
int ensure_buffers(afq_handle *h) {
    afq_comm_state *c = cs_of(h);
    if (c->cap <= 0) c->cap = default_cap(h->nw);
    const size_t slot = (size_t)max_layout(h).size();
    if (c->nw == h->nw && c->nbp == h->nbp && c->slot == slot && c->lists) return AFQ_OK;
    AFQ_HIP(h, hipStreamSynchronize(h->stream));
    free_buffers(c);
    const size_t R = c->nranks, nw = h->nw;
    AFQ_HIP(h, hipMalloc(&c->sendw, sizeof(double) * (nw + 1)));
    AFQ_HIP(h, hipMalloc(&c->gw, sizeof(double) * R * (nw + 1)));
    AFQ_HIP(h, hipMalloc(&c->pix, sizeof(int) * R * nw));
    AFQ_HIP(h, hipMalloc(&c->lists, sizeof(int) * (2 * R + 2 * R * c->cap)));
    AFQ_HIP(h, hipMemset(c->lists, 0, sizeof(int) * (2 * R + 2 * R * c->cap)));
    AFQ_HIP(h, hipMalloc(&c->tickets, sizeof(int) * R));
    AFQ_HIP(h, hipMemset(c->tickets, 0, sizeof(int) * R));
    c->nw = h->nw; c->nbp = h->nbp; c->slot = slot;
    if (uses_windows(c)) {
        c->wl = make_layout((int)R, h->nw, c->window ? c->cap : 0, (long)slot, est_window_doubles(h));
        const int rc = exchange_windows(h);           // may switch an RCCL communicator to the ncclSend / ncclRecv transport
        if (rc) { c->nw = 0; return rc; }
    }
    if (!c->window && R > 1) {
        AFQ_HIP(h, hipMalloc(&c->sbuf, sizeof(cplx) * R * c->cap * slot));
        AFQ_HIP(h, hipMalloc(&c->rbuf, sizeof(cplx) * R * c->cap * slot));
    }
    return AFQ_OK;
}

// stages 1 (prep) / 3-5 (plan, local clones, pack) / 7 (unpack, reset) of one event, queued on h->stream
int stage_prep(afq_handle *h, double r) {
    afq_comm_state *c = cs_of(h);
    int rc = ensure_buffers(h);
    if (rc) return rc;
    ++c->seq_g; ++c->seq_x;
    if (c->win_collectives)
        AFQ_LAUNCH(h, comm_prep_kernel, dim3(c->nranks), dim3(256), 0, h->stream, h->weight, h->nw, r, c->sendw, 1, c->pw, c->wl,
                   c->rank, c->seq_g);
    else
        AFQ_LAUNCH(h, comm_prep_kernel, dim3((h->nw + 1 + 255) / 256), dim3(256), 0, h->stream, h->weight, h->nw, r, c->sendw, 0,
                   c->pw, c->wl, c->rank, c->seq_g);
    AFQ_POST(h);
    return AFQ_OK;
}

void fill_pack(afq_handle *h, PackArgs &p, bool with_greens, bool send) {
    afq_comm_state *c = cs_of(h);
    p.L = max_layout(h); p.L.with_greens = with_greens ? 1 : 0;
    p.cap = c->cap; p.nranks = c->nranks; p.rank = c->rank;
    p.count = c->lists + (send ? 0 : c->nranks);
    p.idx = c->lists + 2 * c->nranks + (send ? 0 : c->nranks * c->cap);
    p.slot = (long)c->slot;
    p.window = c->window ? 1 : 0; p.pw = c->pw; p.wl = c->wl; p.seq = c->seq_x; p.tickets = c->tickets;
    p.scal = h->scal;
    p.buf = c->window ? (send ? nullptr : c->wl.rbuf(c->win)) : (send ? c->sbuf : c->rbuf);
    p.phi = h->phi; p.ot = h->ot; p.ehyb = h->ehyb; p.phase = h->phase; p.eloc = h->eloc; p.ghalf = h->ghalf;
    p.ovlp_new = h->ovlp_new; p.phi_old = h->phi_old; p.bp_hist = h->bp_hist; p.bp_ph = h->bp_ph; p.G = h->G;
    p.unscaled = h->unscaled; p.detR = h->detR; p.log_detR = h->log_detR; p.bp_cos = h->bp_cos; p.bp_n = h->bp_n;
    p.closed_bad = h->closed_bad; p.closed_epoch = h->closed_epoch; p.closed_half = 0;
}

// fused_prep: the launch of the plan also writes this rank's weights into the peers' windows (window all-gather, one process per
// rank: k_comm_popcontrol); pack_now = false leaves the pack to the combined exchange launch of stage_unpack
int stage_plan_pack(afq_handle *h, double target, bool with_greens, bool fused_prep = false, double r = 0.0, bool pack_now = true) {
    afq_comm_state *c = cs_of(h);
    const long N = (long)c->nranks * h->nw;
    const size_t lds = (sizeof(double) + 3 * sizeof(int)) * (size_t)N;
    if (lds > 160 * 1024 - 256) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "device comb: more than 8000 walkers in total");
    static size_t lds_set[AFQ_MAX_DEVICES] = {0};
    AFQ_HIP(h, afq_raise_lds((const void *)comb_plan_global_kernel, lds, lds_set));
    PlanArgs a;
    a.gw = c->win_collectives ? c->wl.gw(c->win, (int)(c->seq_g & 1)) : c->gw;
    a.gflag = c->win_collectives ? c->wl.flag_g(c->win) : nullptr; a.seq = c->seq_g;
    a.R = c->nranks; a.nw = h->nw; a.rank = c->rank; a.cap = c->cap; a.target = target;
    a.weight = h->weight; a.unscaled = h->unscaled; a.pix_global = c->pix; a.parent_ix = h->parent_ix;
    a.pairs = (int *)h->pack_tmp; a.lists = c->lists; a.scal = h->scal;
    a.prep_R = fused_prep ? c->nranks : 0; a.prep_r = r; a.pw = c->pw; a.wl = c->wl;
    AFQ_LAUNCH(h, comb_plan_global_kernel, dim3(1 + a.prep_R), dim3(256), lds, h->stream, a);
    AFQ_POST(h);
    int rc = k_clone_pairs(h, with_greens, true);                 // ... and every weight back to 1 (handler.py:337-338)
    if (rc) return rc;
    if (c->nranks > 1 && pack_now) {
        PackArgs p;
        fill_pack(h, p, with_greens, true);
        AFQ_LAUNCH(h, comm_pack_kernel<true>, dim3(4, std::min(c->cap, COMM_GRID_Y), c->nranks), dim3(256), 0, h->stream, p);
        AFQ_POST(h);
    }
    return AFQ_OK;
}

int stage_unpack(afq_handle *h, bool with_greens, bool pack_too = false) {
    // (the closed-shell verdict of this rank's population carries over when the walkers that arrive are checked too)
    const bool closed_too = with_greens && h->closed_bad && h->closed_checked_version == h->ghalf_version && h->na == h->nb;
    ++h->ghalf_version;                 // cloned / received walkers bring their Ghalf along
    if (closed_too) h->closed_checked_version = h->ghalf_version;
    afq_comm_state *c = cs_of(h);
    if (c->nranks > 1) {
        PackArgs p;
        fill_pack(h, p, with_greens, false);
        if (closed_too) p.closed_half = (long)h->na * h->M;
        if (pack_too) {
            PackArgs ps;
            fill_pack(h, ps, with_greens, true);
            AFQ_LAUNCH(h, comm_exchange_kernel, dim3(4, std::min(c->cap, COMM_GRID_Y), 2 * c->nranks), dim3(256), 0, h->stream, ps, p);
        } else {
            AFQ_LAUNCH(h, comm_pack_kernel<false>, dim3(4, std::min(c->cap, COMM_GRID_Y), c->nranks), dim3(256), 0, h->stream, p);
        }
        AFQ_POST(h);
    }
    c->events += 1;
    return AFQ_OK;                                                // (the weights went back to 1 with the clones)
}

int check_group(afq_handle **hs, int n, std::string *err) {
    if (!hs || n < 1) return AFQ_EINVAL;
    for (int i = 0; i < n; ++i) {
        afq_comm_state *c = hs[i] ? cs_of(hs[i]) : nullptr;
        if (!c || c->mode != COMM_LOCAL || c->nranks != n || c->rank != i || c->peers.size() != (size_t)n || c->peers[0] != hs[0]) {
            if (err) *err = "the handles are not the ranks 0..n-1 of one afq_comm_init_local communicator";
            return AFQ_ESTATE;
        }
        if (hs[i]->nw != hs[0]->nw || !hs[i]->nw) { if (err) *err = "every rank needs the same number of walkers"; return AFQ_ESTATE; }
    }
    return AFQ_OK;
}

// LOCAL communicator: (re)make the buffers of every rank and point every rank at every window (same process: the
// windows are ordinary device pointers; ranks on different GPUs need peer access)
int ensure_group(afq_handle **hs, int n) {
    bool fresh = false;
    for (int i = 0; i < n; ++i) {
        afq_comm_state *c = cs_of(hs[i]);
        hipSetDevice(hs[i]->device);
        const bool had = c->lists && c->nw == hs[i]->nw && c->nbp == hs[i]->nbp && c->slot == (size_t)max_layout(hs[i]).size();
        const int rc = ensure_buffers(hs[i]);
        if (rc) return rc;
        fresh = fresh || !had;
    }
    if (!fresh) return AFQ_OK;
    for (int i = 0; i < n; ++i) {
        afq_comm_state *c = cs_of(hs[i]);
        hipSetDevice(hs[i]->device);
        for (int j = 0; j < n; ++j) {
            if (cs_of(hs[j])->cap != c->cap || cs_of(hs[j])->slot != c->slot) AFQ_FAIL(hs[i], AFQ_ESTATE, "ranks disagree on the slot layout");
            c->pw.base[j] = cs_of(hs[j])->win;
            if (hs[j]->device != hs[i]->device) {
                const hipError_t e = hipDeviceEnablePeerAccess(hs[j]->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) AFQ_HIP(hs[i], e);
                (void)hipGetLastError();
            }
        }
    }
    return AFQ_OK;
}

// ---- estimator reduction through the windows (IPC / LOCAL): every rank writes its vector into row `rank` of every
// peer's est buffer and raises the peer's flag; every rank then adds the R rows in rank order -- the same additions in
// the same order everywhere, so all ranks hold bit-identical sums
__global__ __launch_bounds__(256) void est_put_kernel(const double *src, long n, PeerWindows pw, WinLayout wl, int rank,
                                                      unsigned long long seq) {
    const int peer = blockIdx.x;
    double *dst = wl.est(pw.base[peer], (int)(seq & 1)) + (long)rank * wl.est_n;
    for (long i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) flag_release(wl.flag_e(pw.base[peer]) + rank, seq);
}

__global__ __launch_bounds__(256) void est_sum_kernel(double *dst, long n, void *win, WinLayout wl, unsigned long long seq,
                                                      double *scal) {
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_ok = 1;
    __syncthreads();
    if (threadIdx.x < wl.R && !flag_wait(wl.flag_e(win) + threadIdx.x, seq)) s_ok = 0;
    __syncthreads();
    if (!s_ok) { if (threadIdx.x == 0 && blockIdx.x == 0) scal[6] = 1.0; return; }
    const double *rows = wl.est(win, (int)(seq & 1));
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int s = 0; s < wl.R; ++s) acc += rows[(long)s * wl.est_n + i];
        dst[i] = acc;
    }
}

// put and sum of one chunk in ONE launch (one process per rank, round 6): blocks 0 .. R-1 write this rank's row into the
// windows and raise the flags, the blocks behind them wait for every rank's row and sum IN PLACE.  Only the block that writes
// this rank's OWN window reads the vector; the blocks for the other peers copy that row once its flag is up -- so when a sum
// block has seen every flag (its own rank's included) nobody reads the vector any more and it may be overwritten.
__global__ __launch_bounds__(256) void est_reduce_kernel(const double *src, double *dst, long n, PeerWindows pw, void *win, WinLayout wl,
                                                         int rank, unsigned long long seq, double *scal) {
    if ((int)blockIdx.x < wl.R) {
        const int peer = blockIdx.x;
        double *row = wl.est(pw.base[peer], (int)(seq & 1)) + (long)rank * wl.est_n;
        if (peer != rank) {
            __shared__ int s_mine;
            if (threadIdx.x == 0) s_mine = flag_wait(wl.flag_e(win) + rank, seq) ? 1 : 0;
            __syncthreads();
            if (!s_mine) { if (threadIdx.x == 0) scal[6] = 1.0; return; }
            src = wl.est(win, (int)(seq & 1)) + (long)rank * wl.est_n;
        }
        for (long i = threadIdx.x; i < n; i += blockDim.x) row[i] = src[i];
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) flag_release(wl.flag_e(pw.base[peer]) + rank, seq);
        return;
    }
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_ok = 1;
    __syncthreads();
    if (threadIdx.x < wl.R && !flag_wait(wl.flag_e(win) + threadIdx.x, seq)) s_ok = 0;
    __syncthreads();
    if (!s_ok) { if (threadIdx.x == 0 && (int)blockIdx.x == wl.R) scal[6] = 1.0; return; }
    const double *rows = wl.est(win, (int)(seq & 1));
    const long nb = (long)gridDim.x - wl.R;
    for (long i = ((long)blockIdx.x - wl.R) * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) {
        double acc = 0.0;
        for (int s = 0; s < wl.R; ++s) acc += rows[(long)s * wl.est_n + i];
        dst[i] = acc;
    }
}

// sum over ranks of the device vector v[0..n) in place, window collectives (chunks of the window's est rows)
int window_allreduce(afq_handle *h, double *v, long n) {
    afq_comm_state *c = cs_of(h);
    for (long o = 0; o < n; o += c->wl.est_n) {
        const long m = std::min(c->wl.est_n, n - o);
        ++c->seq_e;
        const unsigned nblk = (unsigned)std::min<long>(64, (m + 255) / 256);
        AFQ_LAUNCH(h, est_reduce_kernel, dim3(c->nranks + nblk), dim3(256), 0, h->stream, v + o, v + o, m, c->pw, c->win, c->wl,
                   c->rank, c->seq_e, h->scal);
        AFQ_POST(h);
    }
    return AFQ_OK;
}

// ---- known-answer probe (afq_comm_probe) ----------------------------------------------------------------------
__device__ inline double probe_value(int src, int dst, long i) { return 1000.0 * src + 10.0 * dst + 0.001 * (double)(i % 997); }

__global__ __launch_bounds__(256) void probe_fill_kernel(cplx *sbuf, long slot, long cap, int rank, int R) {
    const int peer = blockIdx.x;
    cplx *s = sbuf + (long)peer * cap * slot;
    for (long i = threadIdx.x; i < slot; i += blockDim.x) s[i] = cmake(probe_value(rank, peer, i), -probe_value(rank, peer, i));
}

__global__ __launch_bounds__(256) void probe_put_kernel(long slot, long cap, PeerWindows pw, WinLayout wl, int rank,
                                                        unsigned long long seq) {
    const int peer = blockIdx.x;
    if (peer == rank) return;
    cplx *s = wl.rbuf(pw.base[peer]) + (long)rank * cap * slot;
    for (long i = threadIdx.x; i < slot; i += blockDim.x) s[i] = cmake(probe_value(rank, peer, i), -probe_value(rank, peer, i));
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) flag_release(wl.flag_x(pw.base[peer]) + rank, seq);
}

// mismatches of slot 0 of every peer row of rbuf against the pattern -> bad[0]; bad[1] = 1 if a flag never came
__global__ __launch_bounds__(256) void probe_check_kernel(const cplx *rbuf, long slot, long cap, int rank, int R,
                                                          const unsigned long long *flags, unsigned long long seq,
                                                          unsigned long long *bad) {
    const int peer = blockIdx.x;
    if (peer == rank) return;
    __shared__ int ok;
    if (threadIdx.x == 0) ok = flags ? (flag_wait(flags + peer, seq) ? 1 : 0) : 1;
    __syncthreads();
    if (!ok) { if (threadIdx.x == 0) atomicAdd(bad + 1, 1ull); return; }
    const cplx *s = rbuf + (long)peer * cap * slot;
    unsigned long long n = 0;
    for (long i = threadIdx.x; i < slot; i += blockDim.x) {
        const double v = probe_value(peer, rank, i);
        if (s[i].x != v || s[i].y != -v) ++n;
    }
    if (n) atomicAdd(bad, n);
}

// the rows of a window all-gather are complete once every rank's flag has arrived; bad[1] counts flags that never came
__global__ void probe_wait_kernel(const unsigned long long *flags, int R, unsigned long long seq, unsigned long long *bad) {
    if ((int)threadIdx.x < R && !flag_wait(flags + threadIdx.x, seq)) atomicAdd(bad + 1, 1ull);
}

int sendrecv_slots(afq_handle *h, RcclApi *api) {
    afq_comm_state *c = cs_of(h);
    const size_t bytes = sizeof(cplx) * (size_t)c->cap * c->slot;
    afq_note_launch(h, "ncclSend/Recv(walker slots)");
    AFQ_NCCL(h, api, api->GroupStart());
    for (int p = 0; p < c->nranks; ++p) {
        if (p == c->rank) continue;
        AFQ_NCCL(h, api, api->Send(c->sbuf + (size_t)p * c->cap * c->slot, bytes, ncclChar, p, c->nccl, h->stream));
        AFQ_NCCL(h, api, api->Recv(c->rbuf + (size_t)p * c->cap * c->slot, bytes, ncclChar, p, c->nccl, h->stream));
    }
    AFQ_NCCL(h, api, api->GroupEnd());
    c->sendrecv_bytes += (unsigned long long)bytes * (c->nranks - 1);
    return AFQ_OK;
}

}  // namespace

int k_comm_size(afq_handle *h) { return h->comm ? cs_of(h)->nranks : 1; }

// scal[3..8] belong to the communicator (overflow flag, largest run, events, sticky error, walkers / bytes sent): a new
// communicator -- or the host-mediated path after a failed candidate -- must not inherit the previous one's sticky flags
static void clear_comm_scalars(afq_handle *h) {
    if (h->scal) { (void)hipMemset(h->scal + 3, 0, 6 * sizeof(double)); h->scal_cache_valid = false; }
}

// g_wait_ticks is one symbol PER DEVICE (one code object instance each): the host mirror is kept per device too, so that
// with several GPUs in one process (afq_comm_init_local) a probe restores the budget of the device it shortened, and a
// budget set through one handle is never reported for -- or "restored" onto -- another device
static double g_wait_seconds_host[AFQ_MAX_DEVICES];
static bool g_wait_seconds_known[AFQ_MAX_DEVICES];
static int current_device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= AFQ_MAX_DEVICES) d = 0;
    return d;
}
static double wait_budget_of_current_device() {
    const int d = current_device_slot();
    return g_wait_seconds_known[d] ? g_wait_seconds_host[d] : WAIT_SECONDS_DEFAULT;
}
static int set_wait_budget(double seconds) {          // for the CURRENT device (callers hipSetDevice(h->device) first)
    if (!(seconds > 0.0)) return AFQ_EINVAL;
    const unsigned long long ticks = (unsigned long long)std::min(seconds * 1e8, 9.0e17);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wait_ticks), &ticks, sizeof(ticks)) != hipSuccess) return AFQ_EHIP;
    const int d = current_device_slot();
    g_wait_seconds_host[d] = seconds; g_wait_seconds_known[d] = true;
    return AFQ_OK;
}

void k_comm_destroy(afq_handle *h) {
    afq_comm_state *c = cs_of(h);
    if (!c) return;
    clear_comm_scalars(h);
    free_buffers(c);
    if (c->nccl) { RcclApi *api = rccl_api(); if (api) api->CommDestroy(c->nccl); }
    if (c->ev) hipEventDestroy(c->ev);
    if (c->ev2) hipEventDestroy(c->ev2);
    delete c;
    h->comm = nullptr;
}

// one event on an RCCL or IPC communicator (stages 1-7 above)
int k_comm_popcontrol(afq_handle *h, double r, double target, bool with_greens) {
    h->scal_cache_valid = false;
    afq_comm_state *c = cs_of(h);
    if (!c) AFQ_FAIL(h, AFQ_ESTATE, "no communicator");
    if (c->mode == COMM_LOCAL) AFQ_FAIL(h, AFQ_ESTATE, "in-process communicator: use afq_popcontrol_comb_local for all ranks at once");
    RcclApi *api = c->mode == COMM_RCCL ? rccl_api() : nullptr;
    if (c->mode == COMM_RCCL && !api) AFQ_FAIL(h, AFQ_ESTATE, "RCCL is not loaded");
    // Launches of one event (round 6; seven before): window collectives -- [prep + plan] [clones + reset] [pack + unpack];
    // RCCL collectives -- [prep] ncclAllGather [plan] [clones + reset] [pack + unpack], or with the ncclSend / ncclRecv
    // transport [pack] ncclSend/Recv [unpack]
    int rc;
    if (c->win_collectives) {
        if ((rc = ensure_buffers(h))) return rc;
        ++c->seq_g; ++c->seq_x;
    } else {
        if ((rc = stage_prep(h, r))) return rc;
        afq_note_launch(h, "ncclAllGather(weights)");
        AFQ_NCCL(h, api, api->AllGather(c->sendw, c->gw, (size_t)h->nw + 1, ncclDouble, c->nccl, h->stream));
    }
    const bool one_launch = c->nranks > 1 && c->window;
    if ((rc = stage_plan_pack(h, target, with_greens, c->win_collectives, r, !one_launch))) return rc;
    if (c->nranks > 1 && !c->window && (rc = sendrecv_slots(h, api))) return rc;
    return stage_unpack(h, with_greens, one_launch);
}

extern "C" {

int afq_comm_available(void) { return rccl_api() ? 1 : 0; }

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

static int comm_attach(afq_handle *h, afq_comm_state *c) {
    h->comm = c;
    clear_comm_scalars(h);
    const char *env = getenv("AFQ_COMM_TIMEOUT_S");
    if (env && atof(env) > 0.0 && set_wait_budget(atof(env)) != AFQ_OK) AFQ_FAIL(h, AFQ_EHIP, "could not set the communicator's wait budget");
    if (h->nw) {    // walker.total_weight starts as the size of the whole population (walkers/handler.py:164)
        const double tw0 = (double)h->nw * c->nranks;
        AFQ_HIP(h, hipMemcpy(h->scal, &tw0, sizeof(double), hipMemcpyHostToDevice));
    }
    return AFQ_OK;
}

int afq_comm_init(afq_handle *h, const void *unique_id, int rank, int nranks) {
    if (!h || !unique_id || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return AFQ_EINVAL;
    hipSetDevice(h->device);
    RcclApi *api = rccl_api();
    if (!api) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "librccl could not be loaded (dlopen librccl.so.1)");
    k_comm_destroy(h);
    afq_comm_state *c = new afq_comm_state();
    c->rank = rank; c->nranks = nranks; c->mode = COMM_RCCL;
    c->window = true; c->win_collectives = false;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = api->CommInitRank(&c->nccl, nranks, id, rank);
    if (r != ncclSuccess) {
        h->err = std::string("ncclCommInitRank: ") + api->GetErrorString(r);
        delete c;
        return AFQ_EHIP;
    }
    return comm_attach(h, c);
}

int afq_comm_init_ipc(afq_handle *h, int rank, int nranks, afq_allgather_fn allgather, void *user) {
    if (!h || !allgather || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return AFQ_EINVAL;
    hipSetDevice(h->device);
    k_comm_destroy(h);
    afq_comm_state *c = new afq_comm_state();
    c->rank = rank; c->nranks = nranks; c->mode = COMM_IPC;
    c->window = true; c->win_collectives = true;
    c->boot = allgather; c->boot_user = user;
    return comm_attach(h, c);
}

int afq_comm_init_local(afq_handle **handles, int n) {
    if (!handles || n < 1 || n > MAX_RANKS) return AFQ_EINVAL;
    for (int i = 0; i < n; ++i) if (!handles[i]) return AFQ_EINVAL;
    for (int i = 0; i < n; ++i) {
        afq_handle *h = handles[i];
        hipSetDevice(h->device);
        k_comm_destroy(h);
        afq_comm_state *c = new afq_comm_state();
        c->rank = i; c->nranks = n; c->mode = COMM_LOCAL;
        c->window = true; c->win_collectives = true;
        c->peers.assign(handles, handles + n);
        hipEventCreateWithFlags(&c->ev, hipEventDisableTiming);
        hipEventCreateWithFlags(&c->ev2, hipEventDisableTiming);
        const int rc = comm_attach(h, c);
        if (rc) return rc;
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

int afq_comm_set_timeout(afq_handle *h, double seconds) {
    if (!h) return AFQ_EINVAL;
    hipSetDevice(h->device);
    AFQ_HIP(h, hipStreamSynchronize(h->stream));
    const int rc = set_wait_budget(seconds);
    if (rc == AFQ_EINVAL) AFQ_FAIL(h, AFQ_EINVAL, "afq_comm_set_timeout: seconds must be positive");
    if (rc) AFQ_FAIL(h, rc, "afq_comm_set_timeout: hipMemcpyToSymbol failed");
    return AFQ_OK;
}

int afq_comm_set_transport(afq_handle *h, int window) {
    if (!h) return AFQ_EINVAL;
    afq_comm_state *c = cs_of(h);
    if (!c) AFQ_FAIL(h, AFQ_ESTATE, "no communicator");
    if (c->mode != COMM_RCCL && !window) AFQ_FAIL(h, AFQ_EUNSUPPORTED, "only an RCCL communicator has the ncclSend / ncclRecv transport");
    if (c->window == (window != 0)) return AFQ_OK;
    hipSetDevice(h->device);
    AFQ_HIP(h, hipStreamSynchronize(h->stream));
    free_buffers(c);
    c->window = window != 0;
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
    double sc[AFQ_NSCAL];
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
    out[6] = c->window ? (int64_t)sc[7] : -1;                                          // walkers this rank has sent
    out[7] = c->window ? (int64_t)sc[8] : (int64_t)c->sendrecv_bytes;                  // bytes this rank has sent
    out[8] = c->window ? 1 : 0;              // transport: 1 peer windows (live slots only), 0 fixed-size ncclSend / ncclRecv
    out[9] = (int64_t)sc[6];                 // sticky communication error (a peer's flag never arrived)
    out[10] = (int64_t)c->mode;              // 0 in-process, 1 RCCL, 2 IPC windows
    out[11] = c->win ? c->win_kind : 0;      // window memory: 1 uncached, 2 fine-grained, 3 plain device memory
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
        h->greens_valid = false; h->gsum_only = false;
    }
    // a cached Green's function travels only if every rank has one (the slots of a pair must agree)
    bool with_greens = true;
    for (int i = 0; i < n; ++i) with_greens = with_greens && keep[i];
    if ((rc = ensure_group(hs, n))) return rc;
    // One host thread queues every rank's kernels, each stage for all ranks before the next: whatever a kernel waits for
    // (a peer's flag) has been queued before it, also when several streams share a hardware queue.  The events order
    // the streams on top of the flags, so that a rank never even starts a stage before its inputs are complete.
    for (int i = 0; i < n; ++i) {
        hipSetDevice(hs[i]->device);
        if ((rc = stage_prep(hs[i], i == 0 ? r : 0.0))) return rc;        // window all-gather: writes row i of every rank's gw
        AFQ_HIP(hs[i], hipEventRecord(cs_of(hs[i])->ev, hs[i]->stream));
    }
    for (int i = 0; i < n; ++i) {
        afq_handle *h = hs[i];
        hipSetDevice(h->device);
        for (int j = 0; j < n; ++j) if (j != i) AFQ_HIP(h, hipStreamWaitEvent(h->stream, cs_of(hs[j])->ev, 0));
        if ((rc = stage_plan_pack(h, target, with_greens))) return rc;     // pack: live slots straight into the peers' windows
        AFQ_HIP(h, hipEventRecord(cs_of(h)->ev2, h->stream));
    }
    for (int i = 0; i < n; ++i) {
        afq_handle *h = hs[i];
        hipSetDevice(h->device);
        for (int j = 0; j < n; ++j) if (j != i) AFQ_HIP(h, hipStreamWaitEvent(h->stream, cs_of(hs[j])->ev2, 0));
        if ((rc = stage_unpack(h, with_greens))) return rc;
        h->greens_valid = with_greens; h->gsum_only = false;
        // a rank's next prep overwrites rows of the OTHER ranks' gw of the other parity only; its next pack writes into
        // their windows only after their next prep (all-gather), which follows their unpack in stream order
    }
    if (!parent_ix && !total_out) return AFQ_OK;
    afq_handle *h0 = hs[0];
    hipSetDevice(h0->device);
    double sc[8];
    AFQ_HIP(h0, hipMemcpyAsync(sc, h0->scal, sizeof(sc), hipMemcpyDeviceToHost, h0->stream));
    if (parent_ix)
        AFQ_HIP(h0, hipMemcpyAsync(parent_ix, cs_of(h0)->pix, sizeof(int) * (size_t)n * nw, hipMemcpyDeviceToHost, h0->stream));
    for (int i = 0; i < n; ++i) { hipSetDevice(hs[i]->device); AFQ_HIP(hs[i], hipStreamSynchronize(hs[i]->stream)); }
    if (total_out) *total_out = sc[0];
    if (sc[1] < 0) AFQ_FAIL(h0, AFQ_EWEIGHT, "total walker weight below 1e-8");
    if (sc[3] != 0.0) AFQ_FAIL(h0, AFQ_EOVERFLOW, "more walkers moved between two ranks than the exchange slots hold");
    if (sc[6] != 0.0) AFQ_FAIL(h0, AFQ_ECOMM, "communicator: a rank never signalled (the wait budget of afq_comm_set_timeout ran out on the device)");
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
    if (c->mode == COMM_LOCAL) AFQ_FAIL(h, AFQ_ESTATE, "in-process communicator: use afq_estimates_allreduce_local");
    RcclApi *api = c->mode == COMM_RCCL ? rccl_api() : nullptr;
    if (c->mode == COMM_RCCL && !api) AFQ_FAIL(h, AFQ_ESTATE, "RCCL is not loaded");
    hipSetDevice(h->device);
    if (c->win_collectives) { const int rc = ensure_buffers(h); if (rc) return rc; }
    if (!buf) {     // the device accumulators of afq_estimates_update, in place: no host round trip
        { const int rc = k_estimates(h, 0, true); if (rc) return rc; }     // (sums still in the per-walker accumulators)
        if (c->win_collectives) {
            int rc = window_allreduce(h, (double *)h->estimates, 2 * AFQ_EST_COUNT_);
            if (!rc && h->rdm_on && h->rdm_acc) rc = window_allreduce(h, h->rdm_acc, 2L * h->M * h->M);
            return rc;
        }
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
    int rcw = AFQ_OK;
    if (e == hipSuccess) {
        if (c->win_collectives) rcw = window_allreduce(h, tmp, 2L * nest);
        else r = api->AllReduce(tmp, tmp, 2 * (size_t)nest, ncclDouble, ncclSum, c->nccl, h->stream);
    }
    if (e == hipSuccess && r == ncclSuccess && rcw == AFQ_OK)
        e = hipMemcpyAsync(buf, tmp, sizeof(double) * 2 * (size_t)nest, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(tmp);
    if (rcw) return rcw;
    if (r != ncclSuccess) { h->err = std::string("ncclAllReduce: ") + api->GetErrorString(r); return AFQ_EHIP; }
    AFQ_HIP(h, e);
    if (c->win_collectives) {       // a flag that never came leaves garbage in buf: report it here, this call synchronises anyway
        double err = 0.0;
        AFQ_HIP(h, hipMemcpy(&err, h->scal + 6, sizeof(double), hipMemcpyDeviceToHost));
        if (err != 0.0) AFQ_FAIL(h, AFQ_ECOMM, "communicator: a peer rank never signalled its estimator row (the wait budget of afq_comm_set_timeout ran out on the device)");
    }
    return AFQ_OK;
}

int afq_estimates_allreduce_local(afq_handle **hs, int n) {
    std::string err;
    int rc = check_group(hs, n, &err);
    if (rc) { if (hs && n > 0 && hs[0]) hs[0]->err = err; return rc; }
    if ((rc = ensure_group(hs, n))) return rc;
    const bool rdm = hs[0]->rdm_on && hs[0]->rdm_acc;
    for (int i = 0; i < n; ++i) if (rdm && !hs[i]->rdm_acc) AFQ_FAIL(hs[0], AFQ_ESTATE, "one_rdm switched on for some ranks only");
    // the window reduction of the IPC communicator, stage by stage for all ranks (see afq_popcontrol_comb_local)
    for (int pass = 0; pass < (rdm ? 2 : 1); ++pass) {
        const long nn = pass == 0 ? 2L * AFQ_EST_COUNT_ : 2L * hs[0]->M * hs[0]->M;
        for (long o = 0; o < nn; o += cs_of(hs[0])->wl.est_n) {
            const long m = std::min(cs_of(hs[0])->wl.est_n, nn - o);
            for (int i = 0; i < n; ++i) {
                afq_handle *h = hs[i];
                afq_comm_state *c = cs_of(h);
                hipSetDevice(h->device);
                if (pass == 0 && o == 0) { rc = k_estimates(h, 0, true); if (rc) return rc; }
                double *v = (pass == 0 ? (double *)h->estimates : h->rdm_acc) + o;
                ++c->seq_e;
                AFQ_LAUNCH(h, est_put_kernel, dim3(n), dim3(256), 0, h->stream, v, m, c->pw, c->wl, c->rank, c->seq_e);
                AFQ_POST(h);
                AFQ_HIP(h, hipEventRecord(c->ev, h->stream));
            }
            for (int i = 0; i < n; ++i) {
                afq_handle *h = hs[i];
                afq_comm_state *c = cs_of(h);
                hipSetDevice(h->device);
                for (int j = 0; j < n; ++j) if (j != i) AFQ_HIP(h, hipStreamWaitEvent(h->stream, cs_of(hs[j])->ev, 0));
                double *v = (pass == 0 ? (double *)h->estimates : h->rdm_acc) + o;
                const unsigned nblk = (unsigned)std::min<long>(64, (m + 255) / 256);
                AFQ_LAUNCH(h, est_sum_kernel, dim3(nblk), dim3(256), 0, h->stream, v, m, c->win, c->wl, c->seq_e, h->scal);
                AFQ_POST(h);
            }
        }
    }
    return AFQ_OK;
}

// Known-answer round through every piece the first population control will use, before any walker depends on it:
// the all-gather (a pattern per rank), one full exchange slot to and from every peer through the configured transport
// (window writes + flags, or the ncclSend / ncclRecv group), and the all-reduce.  Collective; synchronises.
// mismatch_out (may be NULL): [0] wrong all-gather / all-reduce values, [1] wrong slot elements, [2] flags that never came.
int afq_comm_probe(afq_handle *h, int64_t *mismatch_out) {
    if (!h) return AFQ_EINVAL;
    afq_comm_state *c = cs_of(h);
    if (!c) AFQ_FAIL(h, AFQ_ESTATE, "no communicator");
    if (c->mode == COMM_LOCAL) AFQ_FAIL(h, AFQ_ESTATE, "in-process communicator: nothing to probe");
    if (!h->nw) AFQ_FAIL(h, AFQ_ESTATE, "allocate the walkers first (the buffers are sized from them)");
    hipSetDevice(h->device);
    RcclApi *api = c->mode == COMM_RCCL ? rccl_api() : nullptr;
    if (c->mode == COMM_RCCL && !api) AFQ_FAIL(h, AFQ_ESTATE, "RCCL is not loaded");
    int rc = ensure_buffers(h);
    if (rc) return rc;
    // The ranks enter the probe together (the caller agrees on every step of the bring-up first), so nothing legitimate
    // delays a peer here: a transport that does not work must fail the probe within seconds, not after the long budget of
    // the step loop (three candidates x 300 s would outlast a benchmark's time limit).  Restored on every way out.
    struct ProbeBudget {
        double keep; afq_handle *h;
        ProbeBudget(afq_handle *h_) : keep(wait_budget_of_current_device()), h(h_) {
            if (keep > 20.0) { hipStreamSynchronize(h->stream); set_wait_budget(20.0); }
        }
        ~ProbeBudget() { if (keep > 20.0) { hipStreamSynchronize(h->stream); set_wait_budget(keep); } }
    } probe_budget(h);
    const int R = c->nranks, nw = h->nw;
    int64_t bad[3] = {0, 0, 0};
    // ---- all-gather: rank s contributes [1000 s + i] and r = 0.5 + s
    std::vector<double> wts(nw), gathered((size_t)R * (nw + 1));
    for (int i = 0; i < nw; ++i) wts[i] = 1000.0 * c->rank + i;
    double *wdev = nullptr;
    AFQ_HIP(h, hipMalloc(&wdev, sizeof(double) * nw));
    AFQ_HIP(h, hipMemcpyAsync(wdev, wts.data(), sizeof(double) * nw, hipMemcpyHostToDevice, h->stream));
    ++c->seq_g;
    const double *gsrc = nullptr;
    if (c->win_collectives) {
        AFQ_LAUNCH(h, comm_prep_kernel, dim3(R), dim3(256), 0, h->stream, wdev, nw, 0.5 + c->rank, c->sendw, 1, c->pw, c->wl, c->rank, c->seq_g);
        AFQ_POST(h);
        gsrc = c->wl.gw(c->win, (int)(c->seq_g & 1));     // (complete once every rank's flag has arrived: waited for below)
    } else {
        AFQ_LAUNCH(h, comm_prep_kernel, dim3((nw + 1 + 255) / 256), dim3(256), 0, h->stream, wdev, nw, 0.5 + c->rank, c->sendw, 0, c->pw, c->wl, c->rank, c->seq_g);
        AFQ_POST(h);
        AFQ_NCCL(h, api, api->AllGather(c->sendw, c->gw, (size_t)nw + 1, ncclDouble, c->nccl, h->stream));
        gsrc = c->gw;
    }
    // ---- one slot to and from every peer
    ++c->seq_x;
    unsigned long long *badd = nullptr;
    AFQ_HIP(h, hipMalloc(&badd, 16));
    AFQ_HIP(h, hipMemsetAsync(badd, 0, 16, h->stream));
    if (c->win_collectives) {
        AFQ_LAUNCH(h, probe_wait_kernel, dim3(1), dim3(64), 0, h->stream, c->wl.flag_g(c->win), R, c->seq_g, badd);
        AFQ_POST(h);
    }
    if (R > 1) {
        if (c->window) {
            AFQ_LAUNCH(h, probe_put_kernel, dim3(R), dim3(256), 0, h->stream, (long)c->slot, (long)c->cap, c->pw, c->wl, c->rank, c->seq_x);
            AFQ_POST(h);
            AFQ_LAUNCH(h, probe_check_kernel, dim3(R), dim3(256), 0, h->stream, c->wl.rbuf(c->win), (long)c->slot, (long)c->cap,
                       c->rank, R, c->wl.flag_x(c->win), c->seq_x, badd);
            AFQ_POST(h);
        } else {
            AFQ_LAUNCH(h, probe_fill_kernel, dim3(R), dim3(256), 0, h->stream, c->sbuf, (long)c->slot, (long)c->cap, c->rank, R);
            AFQ_POST(h);
            if ((rc = sendrecv_slots(h, api))) { hipFree(wdev); hipFree(badd); return rc; }
            AFQ_LAUNCH(h, probe_check_kernel, dim3(R), dim3(256), 0, h->stream, c->rbuf, (long)c->slot, (long)c->cap, c->rank, R,
                       (const unsigned long long *)nullptr, c->seq_x, badd);
            AFQ_POST(h);
        }
    }
    // ---- all-reduce of [rank + 1, 2 (rank + 1), ...] (host-buffer variant: the device variant is the same call underneath)
    double red[8];
    for (int i = 0; i < 8; ++i) red[i] = (i + 1.0) * (c->rank + 1);
    if ((rc = afq_estimates_allreduce(h, red, 4))) { hipFree(wdev); hipFree(badd); return rc; }
    AFQ_HIP(h, hipMemcpy(gathered.data(), gsrc, sizeof(double) * gathered.size(), hipMemcpyDeviceToHost));
    for (int s = 0; s < R; ++s) {
        for (int i = 0; i < nw; ++i) if (gathered[(size_t)s * (nw + 1) + i] != 1000.0 * s + i) ++bad[0];
        if (gathered[(size_t)s * (nw + 1) + nw] != 0.5 + s) ++bad[0];
    }
    for (int i = 0; i < 8; ++i) if (red[i] != (i + 1.0) * (R * (R + 1) / 2)) ++bad[0];
    unsigned long long bh[2] = {0, 0};
    AFQ_HIP(h, hipMemcpy(bh, badd, 16, hipMemcpyDeviceToHost));
    bad[1] = (int64_t)bh[0]; bad[2] += (int64_t)bh[1];
    hipFree(wdev); hipFree(badd);
    if (mismatch_out) { mismatch_out[0] = bad[0]; mismatch_out[1] = bad[1]; mismatch_out[2] = bad[2]; }
    if (bad[0] || bad[1] || bad[2]) {
        char msg[256];
        snprintf(msg, sizeof(msg), "communicator probe failed: %lld wrong collective values, %lld wrong slot elements, %lld flags "
                 "that never came (transport %s)", (long long)bad[0], (long long)bad[1], (long long)bad[2],
                 c->window ? "peer windows" : "ncclSend/ncclRecv");
        AFQ_FAIL(h, AFQ_ECOMM, msg);
    }
    return AFQ_OK;
}

}  // extern "C"
