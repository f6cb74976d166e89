// Fused per-walker propagator:   phi <- B . [ sum_{n<=order} V^n / n! ] . B . phi
// (propagation/continuous.py:251, :162-171 with :82-111, :258) in ONE launch.
//
// One 512-thread work-group (8 waves) per live walker.  The walker's Slater
// matrix never leaves the CU between the 2 + order + 2 matrix products:
//   * T (the current right-hand operand, M x (na+nb) complex) lives in LDS in
//     MFMA B-fragment order: [k-chunk of 8][column-tile slot a0 a1 b0 b1][sub-step][64 lanes x 16 B]
//   * the running Taylor sum lives in registers (waves 0-3 own 2 x 2 tiles, waves 4-7 own 3 x 1: 7 per SIMD;
//     for M <= 100 the third tile of waves 4-7 is one v_mfma_f64_4x4x4 unit of 4 rows x 16 columns instead)
//   * the left operands (BH1[0], BH1[1], VHS[w] x order, BH1[0], BH1[1]) stream through a
//     3-slot LDS ring as ONE continuous sequence of k-chunks filled by global_load_lds
//     (A-fragment order), so the pipeline never drains between products
//   * complex products use the 3-multiplication form (see mfma_gemm_wg.h); when BH1 has no imaginary part
//     (real trial and real Cholesky vectors: the usual case) the four one-body products need only the two
//     real-by-complex multiplications
//   * shapes whose tile deal has no holes (template FULL) run a half-chunk pipeline in which every non-MFMA
//     instruction -- ring refill, fragment reads, address arithmetic -- sits between two MFMA groups of the
//     same wave; the other shapes run a generic loop with per-tile validity tests
// One raw s_barrier per k-chunk; two more per product to hand the result back into T.
// Replaces 2 x 2 one-body launches + order Taylor launches + 3 copy kernels + the
// phi ping-pong buffers of the unfused path; dead walkers (qmc/afqmc.py:232) are skipped
// in place.
// Limits: M <= 104 (13 chunks of T = 104 KB of LDS), na, nb <= 32, one VHS matrix per walker.
#include <cstdlib>
#include <type_traits>

#include "mfma_gemm_wg.h"

#define PF_D 3
// waves per work-group: 8 (product) or 16 (experiment: -DPF_NW=16; four waves per SIMD, 2 x 1 tile deals)
#ifndef PF_NW
#define PF_NW 8
#endif
#define PF_NT (64 * PF_NW)
// waves that refill the operand ring: all of them (PF_LW == PF_NW: two fragments per wave and chunk), or the last PF_LW.
// Measured NEGATIVE (round 4, C3): PF_LW = 4 -- waves 4-7, the lighter half of the contiguous-column deal, issuing four
// LDS-DMA instructions per chunk each so that the waves with the 2 x 2 tile blocks never stall in DMA issue -- 149.3 us
// against 140.2 us: four back-to-back DMA instructions hold a wave for several hundred cycles and the chunk barrier makes
// everybody wait for it.
#ifndef PF_LW
#define PF_LW PF_NW
#endif
#define PF_FPW (16 / PF_LW)      // operand fragments each refilling wave moves per chunk

// Tuning builds (make TUNING=1 -> libafqmc_hip_tuning.so) carry timing ablations (a.dbg bits: WRONG results, timing only)
// and s_memtime probes.  They all sit behind this one macro family, which expands to NOTHING in the product build, so the
// kernel below reads straight through and its barrier / vmcnt invariants can be audited without mentally compiling
// anything out:
//   PF_UNLESS(bits) stmt;   the statement is skipped when an ablation bit is set
//   PF_TUNE(code)           code that exists in tuning builds only
//   PF_STAGE(i), PF_BND(n, i)   s_memtime stamps of the phases / product boundaries
#ifdef AFQ_TUNING
#define PF_UNLESS(bits) if (!(a.dbg & (bits)))
#define PF_TUNE(...) __VA_ARGS__
#define PF_STAGE(i) stage_stamp(i)
#define PF_BND(n, i) bnd_stamp(n, i)
#else
#define PF_UNLESS(bits)
#define PF_TUNE(...)
#define PF_STAGE(i)
#define PF_BND(n, i)
#endif

struct PropFusedArgs {
    int M, na, nb, nt, order;
    int t4;                     // Taylor products on v_mfma_f64_4x4x4 (see taylor4 below)
    int dbg;                    // tuning builds: bit 0 = no per-chunk barrier (WRONG results; timing ceiling only)
    unsigned long long *ts;     // tuning builds (AFQ_PF_TS=1): s_memtime stamps of work-group 0, waves 0 and 4
    int vhs_upper;              // vhs holds only the upper triangle of the (symmetric) HS potential
    int same_b;                 // BH1[0] == BH1[1]: one one-body pass serves both spins
    int b_real;                 // BH1 is real: one-body products take 2 real multiplications instead of 3
    int rem4;                   // M <= 100 in the full 7-row-tile deal: rows 96.. as one 4x4x4 unit (see taylor)
    int hyb;                    // > 0: a column slot with at most 4 * hyb <= 12 live columns is multiplied as hyb units of
                                // 16 rows x 4 columns on v_mfma_f64_4x4x4 (see taylor_h)
    int symcols;                // contig with na == nb (round 5): T holds the walker's columns in the order [a 0..15 | b 0..15 |
                                // a 16..23, b 16..23 | a 24.., b 24..] so that a column and its twin of the other spin always go
                                // through the same code and MFMA shape (slots 0, 1: taylor; slot 2: taylor_h's tile; slot 3:
                                // 4x4x4 units): the spin blocks of a closed-shell walker stay bitwise equal through the step
    int closed_try;             // one matrix for both spins, na == nb, a deal without holes: a walker whose spin blocks are bitwise
                                // equal (checked in the kernel, every launch) takes the closed-shell deal of the Taylor products
    int contig;                 // the columns of T are the na + nb columns of the walker back to back (slot = column / 16)
                                // instead of two slots per spin: every matrix of the chain acts on both spins alike
    const cplx *BH1;            // [2, M, M]
    const cplx *vhs;            // [nw, M, M]
    cplx *phi;                  // [nw, M, nt], updated in place
    const int *alive;
    const void *zero16;
    unsigned long long *n_closed;   // afq_counters [3]: walkers that took the closed-shell deal
};

__device__ inline d2_t lds_read_c(unsigned addr) { return lds_read_b128(addr); }

// ds_read_b128 of fragment `idx` (1 KB apart) behind one base register: the offset goes into the instruction,
// so a set of fragment reads costs one address register instead of one per fragment.
template <int OFF> __device__ inline d2_t lds_read_off(unsigned addr) {
    d2_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __attribute__((always_inline)) inline d2_t lds_read_frag(unsigned base, int idx) {
    switch (idx) {
    case 0: return lds_read_off<0>(base);
    case 1: return lds_read_off<1024>(base);
    case 2: return lds_read_off<2048>(base);
    case 3: return lds_read_off<3072>(base);
    case 4: return lds_read_off<4096>(base);
    case 5: return lds_read_off<5120>(base);
    case 6: return lds_read_off<6144>(base);
    default: return lds_read_off<7168>(base);
    }
}

// Real part only (ds_read_b64) of fragment `idx`: the operand of a real matrix.  Reading the full 16 bytes and dropping the
// upper half is not equivalent: the register allocator then lets that dead upper half overlap registers an MFMA issued
// shortly before still has to read (measured: wrong rows in the column-deal one-body pass, varying from run to run).
template <int OFF> __device__ inline double lds_read_re_off(unsigned addr) {
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __attribute__((always_inline)) inline double lds_read_frag_re(unsigned base, int idx) {
    switch (idx) {
    case 0: return lds_read_re_off<0>(base);
    case 1: return lds_read_re_off<1024>(base);
    case 2: return lds_read_re_off<2048>(base);
    case 3: return lds_read_re_off<3072>(base);
    case 4: return lds_read_re_off<4096>(base);
    case 5: return lds_read_re_off<5120>(base);
    case 6: return lds_read_re_off<6144>(base);
    default: return lds_read_re_off<7168>(base);
    }
}

// Work-group barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0), i.e. for every
// operand chunk still in flight in the DMA ring -- a full pipeline drain at each product boundary.
__device__ inline void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// NARROW: at most one column tile per spin (na, nb <= 16); a separate instantiation so that each carries only the
// Taylor tile deals it uses (register allocation and code size of one variant do not tax the other)
// FULL = number of row tiles when every tile of the deal exists (wide: 5-7 row tiles, i.e. 64 < M <= 112 capped by the
// M <= 104 of the kernel, and na, nb > 16; narrow: 6 row tiles; host-checked), else 0: the per-tile validity tests,
// which cost a branch per tile and k-step inside the MFMA blocks, are compiled out.
template <bool NARROW, int FULL>
__global__ __launch_bounds__(PF_NT) void prop_fused_kernel(PropFusedArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int w = blockIdx.x;
    if (!a.alive[w]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int M = a.M, nt = a.nt;
    const int NCH = (M + 7) >> 3;
    const int nrt = (M + 15) >> 4;                               // row tiles that exist
    // ---- LDS carve
    unsigned char *Tf = smem;                                   // [NCH][4][2][1024]
    unsigned char *ring = smem + (size_t)NCH * 8192;            // [PF_D][8 row tiles][2][1024]
    const unsigned tf_l = lds_addr(Tf), ring_l = lds_addr(ring);
    cplx *phi = a.phi + (long)w * M * nt;
    const cplx *vhs = a.vhs + (long)w * M * M;

    // (tuning: coarse stage stamps (AFQ_PF_TS=1), work-group 0, wave 0 -> a.ts[128 + i]; product build: nothing)
    PF_TUNE(auto stage_stamp = [&](int i) {
        if (a.ts && w == 0 && wave == 0) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
            if (lane == 0) a.ts[128 + i] = t;
        }
    };
    auto bnd_stamp = [&](int n, int i) {
        if (a.ts && w == 0 && (wave & 3) == 0 && (n == 3 || n == 4)) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
            if (lane == 0) a.ts[144 + (wave >> 2) * 8 + (n == 4 ? 4 : 0) + i] = t;
        }
    };)
    PF_STAGE(0);
    // ---- A stream: global chunk g = phase * NCH + c; phases: B0 B1 V..V B0 B1, or B V..V B when both spins
    // share one propagator matrix (BH1[0] == BH1[1]: every closed-shell-type Hamiltonian) -- the one-body
    // products of the two spins then run as ONE pass over the matrix with four column tiles per wave
    const int nob = a.same_b ? 1 : 2;
    const int nphase = 2 * nob + a.order;
    const int G = nphase * NCH;
    const long MM = (long)M * M;
    int gi = 0, gi_phase = 0, gi_c = 0, gi_slot = 0;            // next chunk to issue
    // The source addresses of a refill are computed ahead of time (prepare), normally right behind an MFMA block
    // where the integer arithmetic issues under running MFMAs; the refill itself (issueA), which sits in the
    // barrier-aligned part of the chunk loop that all eight waves execute at once, is then two DMA instructions.
    // prepare() is straight-line code (selects, no branches): a taken branch costs a wave far more than the handful
    // of integer instructions it would skip, and one basic block lets the scheduler place them between MFMAs.
    const void *nsrc[PF_FPW];
    bool prepared = false;
    const bool refills = wave >= PF_NW - PF_LW;                  // wave-uniform
    const int lw = wave - (PF_NW - PF_LW);
    const int p_row0 = (lw >> 1) * 16 + lr;                      // fragment f = lw + t * PF_LW: row tile f>>1,
    const int p_kl = 2 * lk + (lw & 1);                          // sub-step f&1 (the same for every t: PF_LW is even)
    auto prepare = [&]() __attribute__((always_inline)) {
        if (!refills) { prepared = true; return; }
        PF_TUNE(if ((a.dbg & 1024) && gi > 2) { prepared = true; return; })       // stale addresses (timing only)
        const bool in_v = gi_phase >= nob && gi_phase < nob + a.order;
        const int spin = gi_phase < nob ? gi_phase : gi_phase - nob - a.order;      // 0 or 1 on live chunks
        const cplx *A = in_v ? vhs : a.BH1 + (spin == 1 ? MM : 0L);
        const bool sym_phase = a.vhs_upper && in_v;
        const bool live = gi < G;
        const int k = gi_c * 8 + p_kl;
#pragma unroll
        for (int t = 0; t < PF_FPW; ++t) {
            const int row = p_row0 + t * (PF_LW / 2) * 16;
            // upper-triangle storage of V: element (row, k < row) lives at (k, row); the 16 lanes of one k then
            // read 256 contiguous bytes instead of 16 rows
            // (branch-free on purpose: a global_load_lds inside a divergent region leaves the LDS slots of the
            // masked-off lanes stale)
            const int off = (sym_phase && k < row) ? k * M + row : row * M + k;
            nsrc[t] = (live && row < M && k < M) ? (const void *)(A + off) : a.zero16;
            // materialise here: without a use at this point the optimiser sinks the whole computation down to the
            // refill that consumes it, i.e. out of the MFMA shadow it was placed in and into the barrier-aligned part
            asm volatile("" : "+v"(nsrc[t]));
        }
        ++gi;
        const bool wrap = gi_c + 1 == NCH;
        gi_c = wrap ? 0 : gi_c + 1;
        gi_phase += wrap ? 1 : 0;
        prepared = true;
    };
    auto issueA = [&]() __attribute__((always_inline)) {
        if (!refills) return;
        if (!prepared) prepare();
        unsigned char *dst = ring + (size_t)gi_slot * 16384;
        PF_TUNE(if (a.dbg & 512) { for (int t = 0; t < PF_FPW; ++t) asm volatile("" ::"v"(nsrc[t]), "s"(dst)); } else)   // addresses computed, DMA not issued
#pragma unroll
        for (int t = 0; t < PF_FPW; ++t) glds16(nsrc[t], dst + (lw + t * PF_LW) * 1024);
        if (++gi_slot == PF_D) gi_slot = 0;
        prepared = false;
    };
    issueA(); issueA();                                          // PF_D - 1 chunks in flight
    prepare();
    // ---- Closed-shell walker?  (spin blocks bitwise equal: checked on the walker itself, every launch, no state.)  Every
    // matrix of the chain acts on both spins alike (closed_try: one one-body matrix, as many electrons of either spin), so the
    // beta half would stay the bitwise copy of the alpha half through all the products: the Taylor stage multiplies the
    // alpha half only (closed-shell deals below) and copies it into the beta half ahead of the closing one-body pass.
    bool closed = false;
    if (a.closed_try) {
        bool same = true;
        for (int p = tid >> 5; p < M; p += PF_NT / 32) {
            const int c = tid & 31;
            if (c < a.na) {
                const cplx x = phi[p * nt + c], y = phi[p * nt + a.na + c];
                same = same && __double_as_longlong(x.x) == __double_as_longlong(y.x) &&
                       __double_as_longlong(x.y) == __double_as_longlong(y.y);
            }
        }
        closed = __builtin_amdgcn_readfirstlane(__syncthreads_and(same ? 1 : 0)) != 0;
        if (closed && tid == 0) atomicAdd(a.n_closed, 1ULL);
    }
    // Layout of T for THIS walker: the contiguous-column layout (a.contig) serves open-shell walkers; a closed-shell one takes
    // the two-slots-per-spin layout [a0 a1 b0 b1], whose alpha half is three tiles per SIMD.  (Measured, C3, tuning build:
    // 126.7 us; keeping the contiguous layout and leaving out its one wholly redundant slot [b 0..15] 145.3; neither 162.3.)
    const bool relayout = closed && a.contig;
    const int contig = relayout ? 0 : a.contig, hyb = relayout ? 0 : a.hyb;
    // T column -> walker column (identity unless a.symcols: see PropFusedArgs)
    const int sym_n = (a.symcols && !relayout) ? a.na : 0;
    // (slots 0 and 1 are multiplied by the same code (taylor), slot 2 by taylor_h's full tile, slot 3 as 4x4x4 units -- three
    //  summation orders: a column and its twin must sit in the same KIND of slot: [a 0..15 | b 0..15 | a 16..23, b 16..23 |
    //  a 24.., b 24..])
    auto wcol = [&](const int c) -> int {
        if (!sym_n) return c;
        if (c < 16) return c;
        if (c < 32) return sym_n + (c - 16);
        if (c < 40) return 16 + (c - 32);
        if (c < 48) return sym_n + 16 + (c - 40);
        const int e = c - 48, extra = sym_n - 24;
        return e < extra ? 24 + e : sym_n + 24 + (e - extra);
    };

    // ---- phi[w] -> T (B-fragment order), padding zeroed: one sweep over the ENTRIES of T (16 bytes each; entry index =
    // ((chunk * 4 + slot) * 2 + (p & 1)) * 64 + ((p & 7) >> 1) * 16 + column in the slot, i.e. shifts and masks only), each
    // either a walker element or a zero -- instead of a zero fill, a barrier and a sweep over the walker with a division
    // by the column count per element
    PF_UNLESS((32 | 128))
    for (int e = tid; e < NCH * 512; e += PF_NT) {
        const int j = e & 15, kk = (e >> 4) & 3, pb = (e >> 6) & 1, slot = (e >> 7) & 3, ch = e >> 9;
        const int p = ch * 8 + 2 * kk + pb, sp = contig ? 0 : slot >> 1, col = (contig ? slot : slot & 1) * 16 + j;
        const int ns_ = contig ? nt : sp ? a.nb : a.na, off_ = sp ? a.na : 0;
        const bool ok = p < M && col < ns_;
        const cplx v = phi[ok ? p * nt + off_ + wcol(col) : 0];
        ((d2_t *)Tf)[e] = ok ? (d2_t){v.x, v.y} : (d2_t){0.0, 0.0};
    }
    __syncthreads();

    // accumulator-layout address of element (row tile ti, reg r) of column slot cs for this lane
    auto t_ok = [&](int ti, int r) -> bool { return 2 * ti + (r >> 1) < NCH; };   // rows past the last chunk do not exist
    // = lane part + wave-uniform tile part + small immediates in r.  The lane part is laundered through an empty
    // asm at every use: otherwise the compiler keeps one address register per tile alive across the whole MFMA
    // loop, spills them, and the reload's s_waitcnt vmcnt(0) drains the operand DMA ring once per product.
    const unsigned t_lane = (unsigned)((lk & 1) * 1024 + ((lk >> 1) * 16 + lr) * 16);
    auto t_addr = [&](int ti, int r, int cs) -> unsigned {
        unsigned base = t_lane;
        asm volatile("" : "+v"(base));
        return base + (unsigned)(((2 * ti * 4 + cs) * 2) * 1024) + (unsigned)((r >> 1) * 8192 + (r & 1) * 512);
    };

    PF_STAGE(1);
    int ring_slot = 0;                                           // slot of the chunk being consumed
    // one k-chunk of a product: wait own DMA, barrier, refill the ring, hand back the slot base
    auto next_chunk = [&]() __attribute__((always_inline)) -> unsigned {
        PF_UNLESS(18)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF_D - 2) * PF_FPW) : "memory");
        PF_UNLESS(1)
        __builtin_amdgcn_s_barrier();
        PF_UNLESS(2)
        issueA();
        const unsigned sl = ring_l + ring_slot * 16384;
        if (++ring_slot == PF_D) ring_slot = 0;
        return sl;
    };
    // the same without the refill: the caller issues issueA() itself, behind its first MFMAs of the chunk
    auto next_chunk_sync = [&]() __attribute__((always_inline)) -> unsigned {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF_D - 2) * PF_FPW) : "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned sl = ring_l + ring_slot * 16384;
        ring_slot = ring_slot + 1 == PF_D ? 0 : ring_slot + 1;
        return sl;
    };

    // ------------------------------------------------------------------ one-body product
    // wave v owns row tile v and NSL column-tile slots starting at slot0: the (up to) two tiles of one spin, or
    // all four when both spins share the matrix
    auto one_body = [&](auto nsl_tag, int slot0, bool to_global, auto real_tag, const int rt) __attribute__((always_inline)) {
        constexpr int NSL = decltype(nsl_tag)::value;
        constexpr bool BR = decltype(real_tag)::value;
        constexpr int SS = (NARROW && FULL) ? 2 : 1;             // narrow + full: only the first column slot of a spin exists
        d4_t P1[NSL], P2[NSL], P3[NSL];
        bool cv[NSL];
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            P1[j] = (d4_t){0, 0, 0, 0}; P2[j] = (d4_t){0, 0, 0, 0}; P3[j] = (d4_t){0, 0, 0, 0};
            const int cs = slot0 + j * SS;
            cv[j] = rt < nrt && (cs & 1) < ((((cs >> 1) ? a.nb : a.na) + 15) >> 4);
        }
        auto load_frags = [&](unsigned sl, int c, d2_t (&av)[2], d2_t (&bv)[NSL][2]) {
            const unsigned abase = sl + rt * 2048 + lane * 16;
            const unsigned bbase = tf_l + (c * 4 + slot0) * 2048 + lane * 16;
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                if (BR) av[ss][0] = lds_read_frag_re(abase, ss);
                else av[ss] = lds_read_frag(abase, ss);
#pragma unroll
                for (int j = 0; j < NSL; ++j) bv[j][ss] = lds_read_frag(bbase, j * SS * 2 + ss);
            }
        };
        auto mfmas = [&](d2_t (&av)[2], d2_t (&bv)[NSL][2]) {
            __builtin_amdgcn_sched_barrier(0);
            PF_UNLESS(64)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                for (int j = 0; j < NSL; ++j)
                    if (FULL || cv[j]) {        // FULL: a wave without a row tile multiplies the zero fragments the DMA put there
                        if (BR) {
                            P1[j] = mfma16(av[ss][0], bv[j][ss][0], P1[j]);
                            P2[j] = mfma16(av[ss][0], bv[j][ss][1], P2[j]);
                        } else {
                            P1[j] = mfma16(av[ss][0], bv[j][ss][0], P1[j]);
                            P2[j] = mfma16(av[ss][1], bv[j][ss][1], P2[j]);
                            P3[j] = mfma16(av[ss][0] + av[ss][1], bv[j][ss][0] + bv[j][ss][1], P3[j]);
                        }
                    }
            __builtin_amdgcn_sched_barrier(0);
            if (!prepared) prepare();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
#if PF_NW == 8
        if (FULL) {
            // the half-chunk pipeline of the Taylor products below (see there): refill, fragment reads and address
            // arithmetic between the MFMA groups of the sub-step in flight
            constexpr int NR = 1 + NSL;
            constexpr int RPG = NSL > 1 ? (NR + NSL - 2) / (NSL > 1 ? NSL - 1 : 1) : NR;
            d2_t a0, a1, b0[NSL], b1[NSL];
            auto half = [&](d2_t &ax, d2_t (&bx)[NSL], d2_t &ay, d2_t (&by)[NSL], const unsigned abase, const unsigned bbase,
                            const int ys, const bool fetch, const bool refill) __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NSL; ++j) {
                    if (BR) {
                        P1[j] = mfma16(ax[0], bx[j][0], P1[j]);
                        P2[j] = mfma16(ax[0], bx[j][1], P2[j]);
                    } else {
                        P1[j] = mfma16(ax[0], bx[j][0], P1[j]);
                        P2[j] = mfma16(ax[1], bx[j][1], P2[j]);
                        P3[j] = mfma16(ax[0] + ax[1], bx[j][0] + bx[j][1], P3[j]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (j == 0 && refill) issueA();
                    if (j == 0 && !refill && !prepared) prepare();
                    if (fetch && (j < NSL - 1 || NSL == 1)) {
#pragma unroll
                        for (int q = 0; q < RPG; ++q) {
                            const int r = j * RPG + q;
                            if (r < NR) {
                                if (r == 0) {
                                    if (BR) ay[0] = lds_read_frag_re(abase, ys);
                                    else ay = lds_read_frag(abase, ys);
                                }
                                else by[r - 1] = lds_read_frag(bbase, (r - 1) * SS * 2 + ys);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            };
            unsigned sl = next_chunk_sync();                     // reads first, the refill issues under their latency
            {
                const unsigned abase = sl + rt * 2048 + lane * 16, bbase = tf_l + slot0 * 2048 + lane * 16;
                if (BR) a0[0] = lds_read_frag_re(abase, 0);
                else a0 = lds_read_frag(abase, 0);
#pragma unroll
                for (int j = 0; j < NSL; ++j) b0[j] = lds_read_frag(bbase, j * SS * 2);
                __builtin_amdgcn_sched_barrier(0);
                issueA();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            for (int c = 0; c < NCH; ++c) {
                const bool more = c + 1 < NCH;
                half(a0, b0, a1, b1, sl + rt * 2048 + lane * 16, tf_l + (c * 4 + slot0) * 2048 + lane * 16, 1, true, false);
                if (more) sl = next_chunk_sync();
                half(a1, b1, a0, b0, sl + rt * 2048 + lane * 16, tf_l + ((c + 1) * 4 + slot0) * 2048 + lane * 16, 0, more, more);
            }
        } else
#endif
        {
#if PF_NW == 16
        // four waves per SIMD: the other waves' MFMAs cover this wave's LDS reads, one register set is enough
        d2_t avA[2], bvA[NSL][2];
        for (int c = 0; c < NCH; ++c) {
            load_frags(next_chunk(), c, avA, bvA);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mfmas(avA, bvA);
        }
#else
        // fragments of chunk c+1 are read while the MFMAs of chunk c run (two register sets)
        d2_t avA[2], bvA[NSL][2], avB[2], bvB[NSL][2];
        load_frags(next_chunk(), 0, avA, bvA);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int c = 0; c < NCH; c += 2) {
            if (c + 1 < NCH) load_frags(next_chunk(), c + 1, avB, bvB);
            mfmas(avA, bvA);
            if (c + 1 < NCH) {
                if (c + 2 < NCH) load_frags(next_chunk(), c + 2, avA, bvA);
                mfmas(avB, bvB);
            }
        }
#endif
        }
        if (!to_global) __builtin_amdgcn_s_barrier();            // everyone finished reading these T columns
        int lk_e = lk, lr_e = lr;                                // laundered: keeps the store addresses from being
        asm volatile("" : "+v"(lk_e), "+v"(lr_e));               // computed (and kept alive) ahead of the MFMA loop
#pragma unroll
        for (int j = 0; j < NSL; ++j)
            if (cv[j]) {
                const int cs = slot0 + j * SS, sp = contig ? 0 : cs >> 1;
                const int ns_ = contig ? nt : sp ? a.nb : a.na, off_ = sp ? a.na : 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double re = BR ? P1[j][r] : P1[j][r] - P2[j][r];
                    const double im = BR ? P2[j][r] : P3[j][r] - P1[j][r] - P2[j][r];
                    if (to_global) {
                        const int row = rt * 16 + lk_e + 4 * r, col = (contig ? cs : cs & 1) * 16 + lr_e;
                        PF_UNLESS(256)
                        if (row < M && col < ns_) phi[(long)row * nt + off_ + wcol(col)] = cmake(re, im);
                    } else if (t_ok(rt, r)) {
                        *(d2_t *)(Tf + t_addr(rt, r, cs)) = (d2_t){re, im};
                    }
                }
            }
    };
    // Column deal of a one-body product for the full 7-row-tile shape with M <= 100 and one matrix for both spins
    // (FULL == 7, rem4, same_b): SIMD s = waves s, s + 4 owns column slot s; wave s the row tiles 0-2, wave s + 4 the row
    // tiles 3-5 plus the rows 96 .. M-1 as one 4x4x4 unit (see REM in taylor) -- 6.25 tile equivalents per SIMD instead of
    // the 8 of the row deal above (row tile 6 padded to 16 rows, wave 7 multiplying zeros).  Same half-chunk pipeline.
    auto one_body_col = [&](bool to_global, auto real_tag, auto rem_tag) __attribute__((always_inline)) {
        constexpr bool BR = decltype(real_tag)::value;
        constexpr bool REM = decltype(rem_tag)::value;
        constexpr int NI = 3;
        constexpr int NG = NI + (REM ? 1 : 0), NR = NI + 1 + (REM ? 1 : 0), RPG = (NR + NG - 2) / (NG - 1);
        const int c0 = wave & 3, r0 = (wave >> 2) * 3;
        const unsigned rem_t = (unsigned)((((12 * 4 + c0) * 2 + ((lane >> 4) & 1)) * 1024) + (((lane >> 5) * 16) + (lane & 15)) * 16);
        const unsigned rem_a = (unsigned)(6 * 2048 + ((lane >> 4) * 16 + (lane & 3)) * 16);
        d4_t P1[NI], P2[NI], P3[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) { P1[i] = (d4_t){0, 0, 0, 0}; P2[i] = (d4_t){0, 0, 0, 0}; P3[i] = (d4_t){0, 0, 0, 0}; }
        double Q1 = 0.0, Q2 = 0.0, Q3 = 0.0;
        d2_t a0[NI], a1[NI], b0, b1, q0 = (d2_t){0.0, 0.0}, q1 = (d2_t){0.0, 0.0};
        auto half = [&](d2_t (&ax)[NI], d2_t &bx, d2_t &qx, d2_t (&ay)[NI], d2_t &by, d2_t &qy, const unsigned abase,
                        const unsigned bbase, const unsigned qbase, const int ys, const bool fetch, const bool refill)
            __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gt = 0; gt < NG; ++gt) {
                if (REM && gt == 0) {
                    if (BR) {
                        Q1 = __builtin_amdgcn_mfma_f64_4x4x4f64(qx[0], bx[0], Q1, 0, 0, 0);
                        Q2 = __builtin_amdgcn_mfma_f64_4x4x4f64(qx[0], bx[1], Q2, 0, 0, 0);
                    } else {
                        Q1 = __builtin_amdgcn_mfma_f64_4x4x4f64(qx[0], bx[0], Q1, 0, 0, 0);
                        Q2 = __builtin_amdgcn_mfma_f64_4x4x4f64(qx[1], bx[1], Q2, 0, 0, 0);
                        Q3 = __builtin_amdgcn_mfma_f64_4x4x4f64(qx[0] + qx[1], bx[0] + bx[1], Q3, 0, 0, 0);
                    }
                } else {
                    const int i = REM ? gt - 1 : gt;
                    if (BR) {
                        P1[i] = mfma16(ax[i][0], bx[0], P1[i]);
                        P2[i] = mfma16(ax[i][0], bx[1], P2[i]);
                    } else {
                        P1[i] = mfma16(ax[i][0], bx[0], P1[i]);
                        P2[i] = mfma16(ax[i][1], bx[1], P2[i]);
                        P3[i] = mfma16(ax[i][0] + ax[i][1], bx[0] + bx[1], P3[i]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (gt == 0 && refill) issueA();
                if (gt == 0 && !refill && !prepared) prepare();
                if (fetch && gt < NG - 1) {
#pragma unroll
                    for (int q = 0; q < RPG; ++q) {
                        const int r = gt * RPG + q;
                        if (r < NR) {
                            if (r < NI) {
                                if (BR) ay[r][0] = lds_read_frag_re(abase, r * 2 + ys);
                                else ay[r] = lds_read_frag(abase, r * 2 + ys);
                            } else if (r == NI) {
                                by = lds_read_frag(bbase, ys);
                            } else {
                                if (BR) qy[0] = lds_read_frag_re(qbase, ys);
                                else qy = lds_read_frag(qbase, ys);
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        unsigned sl = next_chunk_sync();                         // reads first, the refill issues under their latency
        {
            const unsigned abase = sl + r0 * 2048 + lane * 16, bbase = tf_l + c0 * 2048 + lane * 16;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (BR) a0[i][0] = lds_read_frag_re(abase, i * 2);
                else a0[i] = lds_read_frag(abase, i * 2);
            }
            b0 = lds_read_frag(bbase, 0);
            if (REM) {
                if (BR) q0[0] = lds_read_frag_re(sl + rem_a, 0);
                else q0 = lds_read_frag(sl + rem_a, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            issueA();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        for (int c = 0; c < NCH; ++c) {
            const bool more = c + 1 < NCH;
            half(a0, b0, q0, a1, b1, q1, sl + r0 * 2048 + lane * 16, tf_l + (c * 4 + c0) * 2048 + lane * 16, sl + rem_a, 1, true, false);
            if (more) sl = next_chunk_sync();
            half(a1, b1, q1, a0, b0, q0, sl + r0 * 2048 + lane * 16, tf_l + ((c + 1) * 4 + c0) * 2048 + lane * 16, sl + rem_a, 0, more, more);
        }
        // everyone finished reading these T columns -- see the Taylor epilogue: the full tiles (rows below the last chunk)
        // are stored first, the barrier sits ahead of the remainder unit's store; the pass that writes phi does not touch T
        // and needs no barrier at all
        int lk_e = lk, lr_e = lr, ln_e = lane;                   // laundered (see one_body)
        asm volatile("" : "+v"(lk_e), "+v"(lr_e), "+v"(ln_e));
        const int sp = contig ? 0 : c0 >> 1, ns_ = contig ? nt : sp ? a.nb : a.na, off_ = sp ? a.na : 0;
        const int cb = (contig ? c0 : c0 & 1) * 16;            // first walker column of the slot (within the spin block)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double re = BR ? P1[i][r] : P1[i][r] - P2[i][r];
                const double im = BR ? P2[i][r] : P3[i][r] - P1[i][r] - P2[i][r];
                if (to_global) {
                    const int row = (r0 + i) * 16 + lk_e + 4 * r, col = cb + lr_e;
                    if (row < M && col < ns_) phi[(long)row * nt + off_ + wcol(col)] = cmake(re, im);
                } else if (t_ok(r0 + i, r)) {
                    *(d2_t *)(Tf + t_addr(r0 + i, r, c0)) = (d2_t){re, im};
                }
            }
        if (!to_global) __builtin_amdgcn_s_barrier();
        if (REM) {
            const double re = BR ? Q1 : Q1 - Q2, im = BR ? Q2 : Q3 - Q1 - Q2;
            if (to_global) {
                const int row = 96 + (ln_e >> 4), col = cb + (ln_e & 15);
                if (row < M && col < ns_) phi[(long)row * nt + off_ + wcol(col)] = cmake(re, im);
            } else {
                *(d2_t *)(Tf + rem_t) = (d2_t){re, im};
            }
        }
    };
    auto one_body_stage = [&](bool to_global) __attribute__((always_inline)) {
        using I2 = std::integral_constant<int, 2>;
#if PF_NW == 8
        if constexpr (FULL == 7) {
            bool col_deal = a.rem4 && a.same_b;
            PF_TUNE(if (a.dbg & (to_global ? 16384 : 8192)) col_deal = false;)
            if (col_deal) {
                if (a.b_real) {
                    if (wave < 4) one_body_col(to_global, std::true_type{}, std::false_type{});
                    else one_body_col(to_global, std::true_type{}, std::true_type{});
                } else {
                    if (wave < 4) one_body_col(to_global, std::false_type{}, std::false_type{});
                    else one_body_col(to_global, std::false_type{}, std::true_type{});
                }
                return;
            }
        }
#endif
#if PF_NW == 16
        // two waves per row tile: each takes half of the column slots
        using I1 = std::integral_constant<int, 1>;
        const int rt = wave >> 1, h2 = wave & 1;
        if (a.same_b) {
            if (a.b_real) one_body(I2{}, 2 * h2, to_global, std::true_type{}, rt);
            else one_body(I2{}, 2 * h2, to_global, std::false_type{}, rt);
        } else if (a.b_real) {
            one_body(I1{}, h2, to_global, std::true_type{}, rt); one_body(I1{}, 2 + h2, to_global, std::true_type{}, rt);
        } else {
            one_body(I1{}, h2, to_global, std::false_type{}, rt); one_body(I1{}, 2 + h2, to_global, std::false_type{}, rt);
        }
#else
        if (NARROW && FULL) {                // column slots 0 and 2 only (see SS in one_body)
            using I1 = std::integral_constant<int, 1>;
            if (a.same_b) {
                if (a.b_real) one_body(I2{}, 0, to_global, std::true_type{}, wave);
                else one_body(I2{}, 0, to_global, std::false_type{}, wave);
            } else if (a.b_real) {
                one_body(I1{}, 0, to_global, std::true_type{}, wave); one_body(I1{}, 2, to_global, std::true_type{}, wave);
            } else {
                one_body(I1{}, 0, to_global, std::false_type{}, wave); one_body(I1{}, 2, to_global, std::false_type{}, wave);
            }
            return;
        }
        using I4 = std::integral_constant<int, 4>;
        if (a.same_b) {
            if (a.b_real) one_body(I4{}, 0, to_global, std::true_type{}, wave);
            else one_body(I4{}, 0, to_global, std::false_type{}, wave);
        } else if (a.b_real) {
            one_body(I2{}, 0, to_global, std::true_type{}, wave); one_body(I2{}, 2, to_global, std::true_type{}, wave);
        } else {
            one_body(I2{}, 0, to_global, std::false_type{}, wave); one_body(I2{}, 2, to_global, std::false_type{}, wave);
        }
#endif
    };

    // (s_setprio for either half of the waves only moves the barrier wait from one SIMD partner to the other:
    //  measured, same chunk time)
    one_body_stage(false);
    lds_barrier();                                               // T = B phi complete
    PF_STAGE(2);

    // ------------------------------------------------------------------ Taylor series
    // Tile deal: the 7 x 4 grid of 16x16 output tiles (M <= 104 rows, two column tiles per spin) is split so that
    // every SIMD carries the same MFMA load.  Waves 0-3 own a 2 x 2 block of the first four row tiles, waves 4-7
    // a 3 x 1 block (row tiles 4-6 of one column tile); waves w and w+4 share a SIMD: 4 + 3 = 7 tiles each
    // instead of the 8 (one of them pure padding) of an 8-row-tile deal.
    // STAG (waves 4-7): the wave crosses the chunk barrier in the MIDDLE of a chunk's MFMAs -- sub-step 1 of chunk c is
    // multiplied right behind barrier c + 1 from fragments that are already in registers, then the fragments of chunk
    // c + 1 are read and its sub-step 0 multiplied.  Its SIMD partner (wave - 4) reads its fragments right behind the
    // same barrier and multiplies afterwards, so the partner's LDS reads / ring refill run under this wave's MFMAs and
    // vice versa, instead of both waves doing the same thing at the same time.
    // REM (full 7-row-tile deal, M <= 100, waves 4-7): the rows 96 .. M-1 of the wave's column slot are not a padded
    // seventh 16x16x4 tile but ONE v_mfma_f64_4x4x4 unit of 4 rows x 16 columns (16 cycles instead of 64 per MFMA):
    // blk = 4-column group, so the B operand IS the T fragment the full tiles use and the A operand is the first four
    // rows of the ring fragment of row tile 6, every 4-lane group reading the same 4 rows (lane layouts decoded with
    // tools/mfma4x4_probe: A lane = 16 k + 4 blk + i, B lane = 16 k + 4 blk + j, D lane = 16 i + 4 blk + j).
    auto taylor = [&](auto ni_tag, auto nj_tag, auto stag_tag, const int r0, const int c0, const int rcount,
                      auto rem_tag) __attribute__((always_inline)) {
        constexpr int NI = decltype(ni_tag)::value, NJ = decltype(nj_tag)::value;
        constexpr bool STAG = decltype(stag_tag)::value;
        constexpr bool REM = decltype(rem_tag)::value;
        static_assert(!REM || (FULL == 7 && NJ == 1), "row-remainder unit: waves 4-7 of the full wide deal");
        // D lane (i, blk, j) of the remainder unit = element (row 96 + i, column 4 blk + j = lane & 15) of slot c0
        const unsigned rem_t = (unsigned)((((12 * 4 + c0) * 2 + ((lane >> 4) & 1)) * 1024) + (((lane >> 5) * 16) + (lane & 15)) * 16);
        const unsigned rem_a = (unsigned)(6 * 2048 + ((lane >> 4) * 16 + (lane & 3)) * 16);   // A(96 + i, k) for every blk
        double RR = 0.0, RI = 0.0;
        if (REM) { const d2_t v = *(const d2_t *)(Tf + rem_t); RR = v[0]; RI = v[1]; }
        bool cv[NJ], rv[NI];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int cs = c0 + j;
            cv[j] = (cs & 1) < ((((cs >> 1) ? a.nb : a.na) + 15) >> 4);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) rv[i] = i < rcount && r0 + i < nrt;
        d4_t SR[NI][NJ], SI[NI][NJ];                              // running sum, this wave's tiles
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d2_t v = (d2_t){0.0, 0.0};
                    if (cv[j] && rv[i] && t_ok(r0 + i, r)) v = *(const d2_t *)(Tf + t_addr(r0 + i, r, c0 + j));
                    SR[i][j][r] = v[0]; SI[i][j][r] = v[1];
                }
        // Software pipeline over the k-chunks of one product: the fragments of chunk c+1 travel LDS -> registers
        // (second register set) while the MFMAs of chunk c run, so the LDS pipe and the MFMA pipe overlap instead
        // of alternating in lock step behind the per-chunk barrier.
        auto load_frags = [&](unsigned sl, int c, d2_t (&av)[NI][2], d2_t (&bv)[NJ][2]) {
            PF_TUNE(if (a.dbg & 4) return;)
            const unsigned abase = sl + r0 * 2048 + lane * 16;
            const unsigned bbase = tf_l + (c * 4 + c0) * 2048 + lane * 16;
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
                for (int i = 0; i < NI; ++i) av[i][ss] = lds_read_frag(abase, i * 2 + ss);
#pragma unroll
                for (int j = 0; j < NJ; ++j) bv[j][ss] = lds_read_frag(bbase, j * 2 + ss);
            }
        };
        for (int n = 1; n <= a.order; ++n) {
            double inv_n = 1.0 / n;                              // here, not in the epilogue: the division runs under the first MFMAs
            asm volatile("" : "+v"(inv_n));
            d4_t P1[NI][NJ], P2[NI][NJ], P3[NI][NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    P1[i][j] = (d4_t){0, 0, 0, 0}; P2[i][j] = (d4_t){0, 0, 0, 0}; P3[i][j] = (d4_t){0, 0, 0, 0};
                }
            double Q1 = 0.0, Q2 = 0.0, Q3 = 0.0;                   // remainder unit (REM)
            auto mfma_ss = [&](d2_t (&av)[NI][2], d2_t (&bv)[NJ][2], const int ss) __attribute__((always_inline)) {
                PF_TUNE(if (a.dbg & 8) return;)
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        if (FULL || (cv[j] && rv[i])) {
                            P1[i][j] = mfma16(av[i][ss][0], bv[j][ss][0], P1[i][j]);
                            P2[i][j] = mfma16(av[i][ss][1], bv[j][ss][1], P2[i][j]);
                            P3[i][j] = mfma16(av[i][ss][0] + av[i][ss][1], bv[j][ss][0] + bv[j][ss][1], P3[i][j]);
                        }
            };
            auto mfmas = [&](d2_t (&av)[NI][2], d2_t (&bv)[NJ][2]) {
                __builtin_amdgcn_sched_barrier(0);
                mfma_ss(av, bv, 0);
                mfma_ss(av, bv, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (!prepared) prepare();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            };
#if PF_NW == 8
            if (FULL) {
                // Every instruction that is not an MFMA sits BETWEEN two MFMA groups of the same wave (program order is
                // pinned with sched_barrier), in the 64-cycle shadows of the MFMAs: the refill of the operand ring, the
                // LDS reads of the fragments needed next, the address arithmetic of the next refill.  Measured before:
                // with that work placed before / behind the MFMA block of a chunk it ran while the matrix pipe of the
                // SIMD idled (the partner wave already waiting at the chunk barrier) -- the kernel took the SUM of its
                // MFMA time (107 us) and its bookkeeping (70 us).
                // The pipeline is half a chunk deep: the sub-step 0 MFMAs of chunk c run while the sub-step 1 fragments of
                // chunk c are read, the wave crosses the barrier of chunk c + 1, and the sub-step 1 MFMAs run while the
                // ring is refilled and the sub-step 0 fragments of chunk c + 1 are read -- one register set per sub-step
                // instead of two per chunk (the 2 x 2 deal needs every register it can get: 96 product + 64 sum).
                constexpr int NG = NI * NJ + (REM ? 1 : 0);       // MFMA groups (one tile or unit: 3 MFMAs) per sub-step
                constexpr int NR = NI + NJ + (REM ? 1 : 0);       // fragment reads per sub-step
                constexpr int RPG = NG > 1 ? (NR + NG - 2) / (NG > 1 ? NG - 1 : 1) : NR;   // reads behind each group but the last
                d2_t a0[NI], b0[NJ], a1[NI], b1[NJ];              // sub-step 0 / sub-step 1 fragments
                d2_t q0 = (d2_t){0.0, 0.0}, q1 = (d2_t){0.0, 0.0};   // remainder rows of the A operand, sub-step 0 / 1
                auto half = [&](d2_t (&ax)[NI], d2_t (&bx)[NJ], d2_t &qx, d2_t (&ay)[NI], d2_t (&by)[NJ], d2_t &qy,
                                const unsigned abase, const unsigned bbase, const unsigned qbase, const int ys,
                                const bool fetch, const bool refill) __attribute__((always_inline)) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int gt = 0; gt < NG; ++gt) {
                        if (REM && gt == 0) {                     // the short unit first: the long groups cover the reads
                            Q1 = __builtin_amdgcn_mfma_f64_4x4x4f64(qx[0], bx[0][0], Q1, 0, 0, 0);
                            Q2 = __builtin_amdgcn_mfma_f64_4x4x4f64(qx[1], bx[0][1], Q2, 0, 0, 0);
                            Q3 = __builtin_amdgcn_mfma_f64_4x4x4f64(qx[0] + qx[1], bx[0][0] + bx[0][1], Q3, 0, 0, 0);
                        }
                        const int g = REM ? (gt == 0 ? 0 : gt - 1) : gt;
                        const int i = g / NJ, j = g % NJ;
                        if (!(REM && gt == 0))
                        // (tuning: timing ablation with the MFMA load a hybrid 16x16x4 / 4x4x4 tiling would leave at most)
                        PF_TUNE(if (!((a.dbg & 4096) && ((NI == 2 && NJ == 2 && g == 3) || (NI == 3 && i == 2)))))
                        {
                        P1[i][j] = mfma16(ax[i][0], bx[j][0], P1[i][j]);
                        P2[i][j] = mfma16(ax[i][1], bx[j][1], P2[i][j]);
                        P3[i][j] = mfma16(ax[i][0] + ax[i][1], bx[j][0] + bx[j][1], P3[i][j]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (gt == 0 && refill) issueA();
                        if (gt == 0 && !refill && !prepared) prepare();
                        if (fetch && (gt < NG - 1 || NG == 1)) {
#pragma unroll
                            for (int q = 0; q < RPG; ++q) {
                                const int r = gt * RPG + q;       // read r: the A tiles, the B tiles, the remainder rows
                                if (r < NR) {
                                    if (r < NI) ay[r] = lds_read_frag(abase, r * 2 + ys);
                                    else if (r < NI + NJ) by[r - NI] = lds_read_frag(bbase, (r - NI) * 2 + ys);
                                    else qy = lds_read_frag(qbase, ys);
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                };
                // first chunk of the product: the fragment reads go out first, the ring refill issues under their latency
                unsigned sl = next_chunk_sync();
                if (n == 4) PF_BND(4, 0);
                {
                    const unsigned abase = sl + r0 * 2048 + lane * 16, bbase = tf_l + c0 * 2048 + lane * 16;
#pragma unroll
                    for (int i = 0; i < NI; ++i) a0[i] = lds_read_frag(abase, i * 2);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) b0[j] = lds_read_frag(bbase, j * 2);
                    if (REM) q0 = lds_read_frag(sl + rem_a, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    issueA();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if (n == 4) PF_BND(4, 1);
                for (int c = 0; c < NCH; ++c) {
                    const bool more = c + 1 < NCH;
                    // (tuning: s_memtime lands in SGPRs; the values are read behind the lgkmcnt(0) wait that closes a half)
                    PF_TUNE(unsigned long long t0, t1, t2, t3; asm volatile("s_memtime %0" : "=s"(t0));)
                    // sub-step 0 of chunk c; fetch its sub-step 1 fragments
                    half(a0, b0, q0, a1, b1, q1, sl + r0 * 2048 + lane * 16, tf_l + (c * 4 + c0) * 2048 + lane * 16, sl + rem_a, 1, true, false);
                    PF_TUNE(asm volatile("s_memtime %0" : "=s"(t1));)
                    if (more) sl = next_chunk_sync();             // chunk c + 1 has landed
                    PF_TUNE(asm volatile("s_memtime %0" : "=s"(t2));)
                    // sub-step 1 of chunk c; refill the ring, fetch the sub-step 0 fragments of chunk c + 1
                    half(a1, b1, q1, a0, b0, q0, sl + r0 * 2048 + lane * 16, tf_l + ((c + 1) * 4 + c0) * 2048 + lane * 16, sl + rem_a, 0, more, more);
                    PF_TUNE(asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3));
                            if (a.ts && w == 0 && n == 3 && (wave & 3) == 0 && lane == 0) {
                                unsigned long long *o = a.ts + ((wave >> 2) * 16 + c) * 4;
                                o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
                            }
                            if (a.ts && w == 0 && n == 3 && lane == 0) {
                                unsigned long long *o = a.ts + 256 + (wave * 16 + c) * 4;
                                o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
                            })
                }
            } else
#endif
            {
#if PF_NW == 16
            d2_t avA[NI][2], bvA[NJ][2];
            for (int c = 0; c < NCH; ++c) {
                load_frags(next_chunk(), c, avA, bvA);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                mfmas(avA, bvA);
            }
#else
            d2_t avA[NI][2], bvA[NJ][2], avB[NI][2], bvB[NJ][2];
            if (STAG) {
                // X = fragments of the chunk whose sub-step 1 is still owed, Y = the set being filled
                auto half = [&](d2_t (&ax)[NI][2], d2_t (&bx)[NJ][2], d2_t (&ay)[NI][2], d2_t (&by)[NJ][2], const int c)
                    __attribute__((always_inline)) {
                    unsigned sl = 0;
                    if (c + 1 < NCH) sl = next_chunk();                     // barrier c + 1
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_ss(ax, bx, 1);                                      // (c, sub-step 1): operands in registers
                    __builtin_amdgcn_sched_barrier(0);
                    if (!prepared) prepare();
                    if (c + 1 < NCH) {
                        load_frags(sl, c + 1, ay, by);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_ss(ay, by, 0);                                  // (c + 1, sub-step 0)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                load_frags(next_chunk(), 0, avA, bvA);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                mfma_ss(avA, bvA, 0);
                __builtin_amdgcn_sched_barrier(0);
                for (int c = 0; c < NCH; c += 2) {
                    half(avA, bvA, avB, bvB, c);
                    if (c + 1 < NCH) half(avB, bvB, avA, bvA, c + 1);
                }
            } else {
                load_frags(next_chunk(), 0, avA, bvA);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int c = 0; c < NCH; c += 2) {
                    if (c + 1 < NCH) load_frags(next_chunk(), c + 1, avB, bvB);
                    mfmas(avA, bvA);
                    if (c + 1 < NCH) {
                        if (c + 2 < NCH) load_frags(next_chunk(), c + 2, avA, bvA);
                        mfmas(avB, bvB);
                    }
                }
            }
#endif
            }
            if (n == 3) PF_BND(3, 0);
            // Everyone has to be through with T_{n-1} before it is overwritten -- but behind the last chunk barrier of the
            // half-chunk pipeline only the LAST chunk of T is still being read.  A wave whose tiles end below that chunk
            // (every wave of the full M <= 100 deal: the last chunk belongs to the remainder unit) stores first and takes the
            // barrier afterwards, just ahead of the remainder unit's store: the epilogue arithmetic of a wave then runs
            // while its SIMD partner still multiplies, instead of behind it.
            const bool store_first = PF_NW == 8 && FULL != 0 && 2 * (r0 + NI) <= NCH - 1;
            if (!store_first) __builtin_amdgcn_s_barrier();
            if (n == 3) PF_BND(3, 1);
            // T_n = product / n goes back to T as the next right-hand operand; after the last term T receives the SUM instead
            // (wave-uniform branch: two straight-line store sequences rather than a select per element)
            const bool last = n == a.order;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (FULL || (cv[j] && rv[i])) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double re = (P1[i][j][r] - P2[i][j][r]) * inv_n;
                            const double im = (P3[i][j][r] - P1[i][j][r] - P2[i][j][r]) * inv_n;
                            SR[i][j][r] += re; SI[i][j][r] += im;
                            if (!last && t_ok(r0 + i, r)) *(d2_t *)(Tf + t_addr(r0 + i, r, c0 + j)) = (d2_t){re, im};
                        }
                    }
            if (last) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        if (FULL || (cv[j] && rv[i])) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (t_ok(r0 + i, r))
                                    *(d2_t *)(Tf + t_addr(r0 + i, r, c0 + j)) = (d2_t){SR[i][j][r], SI[i][j][r]};
                        }
            }
            if (store_first) __builtin_amdgcn_s_barrier();
            if (REM) {
                const double re = (Q1 - Q2) * inv_n, im = (Q3 - Q1 - Q2) * inv_n;
                RR += re; RI += im;
                *(d2_t *)(Tf + rem_t) = last ? (d2_t){RR, RI} : (d2_t){re, im};
            }
            if (n == 3) PF_BND(3, 2);
            // T_n visible: the barrier is the one the first chunk of the next product (or of the closing one-body pass)
            // starts with -- only the wave's own LDS writes have to be done before it gets there
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (n == 3) PF_BND(3, 3);
            PF_STAGE(2 + n);
        }
    };

    // ------------------------------------------------------------------ Taylor series, hybrid column tiling (round 4)
    // With 17 .. 28 electrons per spin the second column slot of a spin is mostly padding (C3: 9 live columns of 16).  A
    // 16x16x4 MFMA on it costs 64 cycles whatever it holds; the same rows as NU = ceil((n - 16) / 4) units of 16 rows x 4
    // columns on v_mfma_f64_4x4x4 (blk = 4-row group: the A operand IS the ring fragment of the row tile, lane 16 k + row;
    // the B operand repeats T[k][4 u + j] in every blk; D lane 16 i + 4 blk + j = element (row 4 blk + i, column 4 u + j))
    // cost 16 NU <= 48.  Deal of the full M <= 100 shape (per k-step, in MFMA cycles, three products per tile / unit):
    //   waves 0-3   row tiles (0,1) or (2,3) x [full slot | unit slot] of one spin     2 x (192 + 48 NU)
    //   waves 4, 5  row tiles 4, 5 of the full slot of a spin (the plain taylor() deal)  384
    //   waves 6, 7  row tiles 4, 5 of the unit slot of a spin + the two remainder units
    //               (rows 96 .. M-1, see REM in taylor) of BOTH slots of that spin       96 NU + 96
    // NU = 3: every SIMD (waves w, w + 4) carries 1056 instead of 1200 cycles per k-step.  Same half-chunk pipeline, same
    // barriers per chunk and per product as taylor(); the short groups go first so that the long ones cover the reads.
    auto taylor_h = [&](auto hf_tag, auto rem_tag, auto nu_tag, const int r0, const int cf, const int cu, const int cr)
        __attribute__((always_inline)) {
        constexpr bool HF = decltype(hf_tag)::value;             // the wave also owns the full slot cf of its row tiles
        constexpr bool HR = decltype(rem_tag)::value;            // the wave owns the remainder units of slots cr, cr + 1
        constexpr int NU = decltype(nu_tag)::value;
        constexpr int NI = 2, NRU = HR ? 2 : 0;
        static_assert(FULL == 7, "hybrid tiling: the full wide deal");
        // B operand of unit u: lane (k = lane >> 4, blk, j = lane & 3) reads T[k][4 u + j] of the fragment (+ 64 u bytes)
        const unsigned u_rd = (unsigned)((lane >> 4) * 256 + (lane & 3) * 16);
        // D lane of a unit -> entry of T: row 4 blk + i of the tile = chunk half blk >> 1, sub-step i & 1, k 2 (blk & 1) + (i >> 1)
        const int ui = lane >> 4, ub = (lane >> 2) & 3;
        const unsigned u_wr = (unsigned)((ub >> 1) * 8192 + (ui & 1) * 1024 + ((2 * (ub & 1) + (ui >> 1)) * 16 + (lane & 3)) * 16);
        auto u_addr = [&](int ti, int u) -> unsigned {
            unsigned base = u_wr;
            asm volatile("" : "+v"(base));                       // (laundered like t_addr: no address register per unit kept alive)
            return base + (unsigned)((2 * ti * 4 + cu) * 2048 + u * 64);
        };
        const unsigned rem_a = (unsigned)(6 * 2048 + ((lane >> 4) * 16 + (lane & 3)) * 16);
        auto rem_t = [&](int t) -> unsigned {
            return (unsigned)((((12 * 4 + cr + t) * 2 + ((lane >> 4) & 1)) * 1024) + (((lane >> 5) * 16) + (lane & 15)) * 16);
        };
        // running sums: full tiles (accumulator layout), units (one element per lane), remainder units
        d4_t SR[NI], SI[NI];
        double UR[NI][NU], UI[NI][NU], RR[2] = {0.0, 0.0}, RI[2] = {0.0, 0.0};
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d2_t v = (d2_t){0.0, 0.0};
                if (HF) v = *(const d2_t *)(Tf + t_addr(r0 + i, r, cf));
                SR[i][r] = v[0]; SI[i][r] = v[1];
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) { const d2_t v = *(const d2_t *)(Tf + u_addr(r0 + i, u)); UR[i][u] = v[0]; UI[i][u] = v[1]; }
        }
        if (HR) {
#pragma unroll
            for (int t = 0; t < 2; ++t) { const d2_t v = *(const d2_t *)(Tf + rem_t(t)); RR[t] = v[0]; RI[t] = v[1]; }
        }
        struct Frag { d2_t a[NI]; d2_t bf; d2_t bu[NU]; d2_t q; d2_t br[2]; };
        constexpr int NG = NI * NU + NRU + (HF ? NI : 0);        // MFMA groups per sub-step, short ones first
        constexpr int NR = NI + NU + (HF ? 1 : 0) + (HR ? 3 : 0);   // fragment reads per sub-step
        constexpr int NGAP = HF ? NG - 1 : NG - 3;               // gaps that take reads (all-short waves: keep the tail free)
        constexpr int RPG = (NR + NGAP - 1) / NGAP;
        for (int n = 1; n <= a.order; ++n) {
            double inv_n = 1.0 / n;
            asm volatile("" : "+v"(inv_n));
            d4_t P1[NI], P2[NI], P3[NI];
            double U1[NI][NU], U2[NI][NU], U3[NI][NU], Q1[2] = {0.0, 0.0}, Q2[2] = {0.0, 0.0}, Q3[2] = {0.0, 0.0};
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                P1[i] = (d4_t){0, 0, 0, 0}; P2[i] = (d4_t){0, 0, 0, 0}; P3[i] = (d4_t){0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < NU; ++u) { U1[i][u] = 0.0; U2[i][u] = 0.0; U3[i][u] = 0.0; }
            }
            // fragment r of sub-step ys of the chunk at ring slot `sl`, T chunk c
            auto read_one = [&](Frag &f, const int r, const unsigned sl, const int c, const int ys) __attribute__((always_inline)) {
                const unsigned abase = sl + r0 * 2048 + lane * 16;
                if (r < NI) { f.a[r] = lds_read_frag(abase, r * 2 + ys); return; }
                if (r < NI + NU) {
                    const int u = r - NI;
                    const unsigned ubase = tf_l + (c * 4 + cu) * 2048 + u_rd + ys * 1024;
                    f.bu[u] = u == 0 ? lds_read_off<0>(ubase) : u == 1 ? lds_read_off<64>(ubase) : lds_read_off<128>(ubase);
                    return;
                }
                int k = r - NI - NU;
                if (HF) {
                    if (k == 0) { f.bf = lds_read_frag(tf_l + (c * 4 + cf) * 2048 + lane * 16, ys); return; }
                    --k;
                }
                if (HR) {
                    if (k == 0) f.q = lds_read_frag(sl + rem_a, ys);
                    else f.br[k - 1] = lds_read_frag(tf_l + (c * 4 + cr + k - 1) * 2048 + lane * 16, ys);
                }
            };
            // MFMA groups of the fragments in X; the reads of sub-step ys of (slot sl, chunk c) into Y in between
            auto half = [&](Frag &x, Frag &y, const unsigned sl, const int c, const int ys, const bool fetch, const bool refill)
                __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g < NI * NU) {                            // a unit of 16 rows x 4 columns
                        const int i = g / NU, u = g % NU;
                        U1[i][u] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.a[i][0], x.bu[u][0], U1[i][u], 0, 0, 0);
                        U2[i][u] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.a[i][1], x.bu[u][1], U2[i][u], 0, 0, 0);
                        U3[i][u] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.a[i][0] + x.a[i][1], x.bu[u][0] + x.bu[u][1], U3[i][u], 0, 0, 0);
                    } else if (g < NI * NU + NRU) {               // a remainder unit of 4 rows x 16 columns
                        const int t = g - NI * NU;
                        Q1[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.q[0], x.br[t][0], Q1[t], 0, 0, 0);
                        Q2[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.q[1], x.br[t][1], Q2[t], 0, 0, 0);
                        Q3[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.q[0] + x.q[1], x.br[t][0] + x.br[t][1], Q3[t], 0, 0, 0);
                    } else {                                      // a full 16 x 16 tile
                        const int i = g - NI * NU - NRU;
                        P1[i] = mfma16(x.a[i][0], x.bf[0], P1[i]);
                        P2[i] = mfma16(x.a[i][1], x.bf[1], P2[i]);
                        P3[i] = mfma16(x.a[i][0] + x.a[i][1], x.bf[0] + x.bf[1], P3[i]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (g == 0 && refill) issueA();
                    if (g == 0 && !refill && !prepared) prepare();
                    if (fetch && g < NGAP) {
#pragma unroll
                        for (int q = 0; q < RPG; ++q)
                            if (g * RPG + q < NR) read_one(y, g * RPG + q, sl, c, ys);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            };
            Frag f0, f1;
            f0.q = (d2_t){0.0, 0.0}; f1.q = (d2_t){0.0, 0.0};
            unsigned sl = next_chunk_sync();
#pragma unroll
            for (int r = 0; r < NR; ++r) read_one(f0, r, sl, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            issueA();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int c = 0; c < NCH; ++c) {
                const bool more = c + 1 < NCH;
                PF_TUNE(unsigned long long t0, t1, t2, t3; asm volatile("s_memtime %0" : "=s"(t0));)
                half(f0, f1, sl, c, 1, true, false);             // sub-step 0 of chunk c; fetch its sub-step 1
                PF_TUNE(asm volatile("s_memtime %0" : "=s"(t1));)
                if (more) sl = next_chunk_sync();                 // chunk c + 1 has landed
                PF_TUNE(asm volatile("s_memtime %0" : "=s"(t2));)
                half(f1, f0, sl, c + 1, 0, more, more);          // sub-step 1 of chunk c; refill, fetch sub-step 0 of chunk c + 1
                PF_TUNE(asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3));
                        if (a.ts && w == 0 && n == 3 && lane == 0) {
                            unsigned long long *o = a.ts + 256 + (wave * 16 + c) * 4;
                            o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
                        })
            }
            // T_n = product / n back into T (rows below the last chunk: nobody reads them any more behind the last chunk
            // barrier), the barrier, then the remainder rows -- exactly the sequence of taylor() with store_first
            const bool last = n == a.order;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (HF) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double re = (P1[i][r] - P2[i][r]) * inv_n, im = (P3[i][r] - P1[i][r] - P2[i][r]) * inv_n;
                        SR[i][r] += re; SI[i][r] += im;
                        *(d2_t *)(Tf + t_addr(r0 + i, r, cf)) = last ? (d2_t){SR[i][r], SI[i][r]} : (d2_t){re, im};
                    }
                }
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const double re = (U1[i][u] - U2[i][u]) * inv_n, im = (U3[i][u] - U1[i][u] - U2[i][u]) * inv_n;
                    UR[i][u] += re; UI[i][u] += im;
                    *(d2_t *)(Tf + u_addr(r0 + i, u)) = last ? (d2_t){UR[i][u], UI[i][u]} : (d2_t){re, im};
                }
            }
            __builtin_amdgcn_s_barrier();
            if (HR) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const double re = (Q1[t] - Q2[t]) * inv_n, im = (Q3[t] - Q1[t] - Q2[t]) * inv_n;
                    RR[t] += re; RI[t] += im;
                    *(d2_t *)(Tf + rem_t(t)) = last ? (d2_t){RR[t], RI[t]} : (d2_t){re, im};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };

    // Measured negative result (round 2, profiles/archive/r02_pmc_prop_fused_t4_vs_t16.txt): the same products on
    // v_mfma_f64_4x4x4 (16 x 4 and 4 x 16 units, 79 % of the MFMA cycles of the padded 16x16x4 grid) are CORRECT but
    // not faster -- 183 us against 177 us: MFMA-busy cycles drop 18 %, wave-parked cycles (s_waitcnt / barrier) rise
    // from 23 % to 35 % of the wave cycles, 1.6 x the instructions, and the broadcast read of a 4-column group of T
    // is a 2-way bank conflict in the spin-padded fragment layout.  The kernel is bound by the per-chunk
    // synchronisation skeleton, not by MFMA issue.  The variant is compiled only into tuning builds (AFQ_T4=1).
#ifdef AFQ_TUNING
#include "k_fused_t4.inc"      // taylor4(): the same products on v_mfma_f64_4x4x4 (tuning builds only)
#endif
    // closed-shell walker in the two-slots-per-spin layout: T(b0, b1) = T(a0, a1) behind the Taylor stage, for the closing
    // one-body pass (which multiplies every slot)
    auto mirror_spin = [&]() __attribute__((always_inline)) {
        lds_barrier();
        for (int e = tid; e < NCH * 256; e += PF_NT) {
            const int ch = e >> 8, i = e & 255;
            ((d2_t *)Tf)[ch * 512 + 256 + i] = ((const d2_t *)Tf)[ch * 512 + i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
#if PF_NW == 16
    {
        // SIMD s (waves s, s + 4, s + 8, s + 12) owns column tile s: row tiles (0,1) (2,3) (4,5) (6); 7 tiles per SIMD
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        const int g = wave >> 2;
        if (g & 1) taylor(I2{}, I1{}, std::true_type{}, 2 * g, wave & 3, g == 3 ? 1 : 2, std::false_type{});
        else taylor(I2{}, I1{}, std::false_type{}, 2 * g, wave & 3, 2, std::false_type{});
    }
#else
    PF_TUNE(if (a.t4) taylor4(); else)
    if (NARROW && FULL) {
        // six row tiles, one column tile per spin: waves 0-3 a pair of the row tiles 0-3, waves 4-7 one of the tiles 4, 5
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if (closed) {
            // closed-shell walker: the six tiles of slot a0, one per wave (waves 6, 7 repeat the row tiles 0, 1 -- the same
            // bits to the same addresses -- so that every wave runs the chunk loop with its barriers and refills): two tiles
            // per SIMD instead of three
            taylor(I1{}, I1{}, std::false_type{}, wave < 6 ? wave : wave - 6, 0, 1, std::false_type{});
            mirror_spin();
        }
        else if (wave < 4) taylor(I2{}, I1{}, std::false_type{}, 2 * (wave & 1), 2 * (wave >> 1), 2, std::false_type{});
        else taylor(I1{}, I1{}, std::false_type{}, 4 + (wave & 1), 2 * ((wave - 4) >> 1), 1, std::false_type{});
    }
    else if (NARROW) {
        // one column tile per spin (slots 0 and 2): nrt x 2 tiles.  Waves 0-3 take a pair of row tiles of rows 0-3,
        // waves 4-7 the row tiles from 4 on: singly when there are six (3 tiles on every SIMD), as a pair + a single
        // per spin when there are seven (4, 3, 4, 3) -- instead of the 5, 2, 5, 2 the wide deal below would give
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if (wave < 4) taylor(I2{}, I1{}, std::false_type{}, 2 * (wave & 1), 2 * (wave >> 1), 2, std::false_type{});
        else if (nrt <= 6) taylor(I2{}, I1{}, std::true_type{}, 4 + (wave & 1), 2 * ((wave - 4) >> 1), 1, std::false_type{});
        else taylor(I2{}, I1{}, std::true_type{}, (wave & 1) ? 6 : 4, 2 * ((wave - 4) >> 1), (wave & 1) ? 1 : 2, std::false_type{});
    }
    else if (FULL == 7 && contig) {
        // Contiguous columns (one matrix for both spins, 48 < na + nb <= 56): three full column slots and a fourth with
        // na + nb - 48 live columns, multiplied as NU = hyb units of 16 rows x 4 columns (taylor_h).  The 18 full tiles,
        // 6 NU units and 4 remainder units are dealt so that no SIMD (waves w, w + 4) carries more than 1056 MFMA cycles
        // per k-step at NU = 1 (1200 in the two-slots-per-spin layout):
        //   waves 0, 2   row tiles (0,1) / (2,3) x slots 0, 1                                  768
        //   waves 1, 3   row tiles (0,1) / (2,3) x [slot 2 | unit slot 3]                      384 + 96 NU
        //   wave 4       row tile 4 x slot 0 + remainder unit of slot 0                        240
        //   wave 5       row tile 5 x slots 0, 1                                               384
        //   wave 6       row tile 4 x slot 1 + remainder unit of slot 1                        240
        //   wave 7       row tiles 4, 5 x [slot 2 | unit slot 3] + remainder units of 2, 3     384 + 96 NU + 96
        if constexpr (FULL == 7) {
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            auto deal = [&](auto nu) __attribute__((always_inline)) {
                if (wave == 0 || wave == 2) taylor(I2{}, I2{}, std::false_type{}, wave, 0, 2, std::false_type{});
                else if (wave == 1 || wave == 3) taylor_h(std::true_type{}, std::false_type{}, nu, wave - 1, 2, 3, 0);
                else if (wave == 4 || wave == 6) taylor(I1{}, I1{}, std::true_type{}, 4, (wave - 4) >> 1, 1, std::true_type{});
                else if (wave == 5) taylor(I1{}, I2{}, std::true_type{}, 5, 0, 1, std::false_type{});
                else taylor_h(std::true_type{}, std::true_type{}, nu, 4, 2, 3, 2);
            };
            if (hyb == 2) deal(I2{});
            else deal(I1{});
        }
    }
    else if (FULL == 7 && hyb) {
        if constexpr (FULL == 7) {
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            using I3 = std::integral_constant<int, 3>;
            auto deal = [&](auto nu) __attribute__((always_inline)) {
                if (wave < 4) taylor_h(std::true_type{}, std::false_type{}, nu, 2 * (wave >> 1), 2 * (wave & 1), 2 * (wave & 1) + 1, 0);
                else if (wave < 6) taylor(I2{}, I1{}, std::true_type{}, 4, 2 * (wave - 4), 2, std::false_type{});
                else taylor_h(std::false_type{}, std::true_type{}, nu, 4, 0, 2 * (wave - 6) + 1, 2 * (wave - 6));
            };
            if (hyb == 3) deal(I3{});
            else if (hyb == 2) deal(I2{});
            else deal(I1{});
        }
    }
    else if (FULL >= 5 && closed) {
        // Closed-shell walker, two slots per spin: the alpha half only -- waves 0-3 two row tiles of one slot each (rows (0,1) /
        // (2,3) x slot a0 / a1), waves 4-7 one row tile each (row 4 and, where it exists, row 5 x a0 / a1; with seven full row
        // tiles waves 4, 5 take the rows (4,5) and waves 6, 7 row 6; with five, waves 6, 7 repeat waves 4, 5), the remainder
        // units of the M <= 100 deal with row tile 4: three tiles per SIMD instead of seven.  Same code per tile as in the
        // deals below (taylor), so the alpha results are theirs bit for bit.
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        const int cs = wave & 1;
        if (wave < 4) taylor(I2{}, I1{}, std::false_type{}, 2 * (wave >> 1), cs, 2, std::false_type{});
        else if constexpr (FULL == 7) {
            if (a.rem4) {
                if (wave < 6) taylor(I1{}, I1{}, std::true_type{}, 4, cs, 1, std::true_type{});
                else taylor(I1{}, I1{}, std::true_type{}, 5, cs, 1, std::false_type{});
            } else {
                if (wave < 6) taylor(I2{}, I1{}, std::true_type{}, 4, cs, 2, std::false_type{});
                else taylor(I1{}, I1{}, std::true_type{}, 6, cs, 1, std::false_type{});
            }
        }
        else taylor(I1{}, I1{}, std::true_type{}, (FULL == 6 && wave >= 6) ? 5 : 4, cs, 1, std::false_type{});
        mirror_spin();
    }
    else if (wave < 4) taylor(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, std::false_type{}, 2 * (wave >> 1), 2 * (wave & 1), 2, std::false_type{});
    else if constexpr (FULL == 7) {
        // rows 96 .. M-1 as one 4x4x4 unit when there are at most four of them (see REM in taylor)
        if (a.rem4) taylor(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, std::true_type{}, 4, wave - 4, 2, std::true_type{});
        else taylor(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, std::true_type{}, 4, wave - 4, 3, std::false_type{});
    }
    else if (FULL >= 5) taylor(std::integral_constant<int, FULL >= 5 ? FULL - 4 : 1>{}, std::integral_constant<int, 1>{}, std::true_type{}, 4, wave - 4, FULL - 4, std::false_type{});
    else taylor(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, std::true_type{}, 4, wave - 4, 3, std::false_type{});
#endif
    if (a.order == 0) lds_barrier();

    one_body_stage(true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PF_STAGE(9);
}

int k_prop_fused_supported(afq_handle *h) {
    return !h->no_fused && !h->vhs_diag && h->nv == 1 && h->M <= 104 && h->na <= 32 && h->nb <= 32 && (h->nb > 0 || AFQ_KNOB_SET("AFQ_PF_NB0"));
}

int k_prop_fused(afq_handle *h) {
    PropFusedArgs a;
    a.M = h->M; a.na = h->na; a.nb = h->nb; a.nt = h->nt; a.order = h->exp_order;
    a.vhs_upper = h->vhs_upper ? 1 : 0;
    // 4x4x4 Taylor products (tuning builds only): M <= 100 (six full row tiles + at most four remainder rows)
    a.t4 = (h->M <= 100 && AFQ_KNOB_SET("AFQ_T4")) ? 1 : 0;
    a.dbg = AFQ_KNOB_INT("AFQ_PF_DBG", 0);
    a.ts = nullptr;
#ifdef AFQ_TUNING
    static unsigned long long *ts_dev = nullptr;
    static int ts_launch = 0;
    if (AFQ_KNOB_SET("AFQ_PF_TS")) {
        if (!ts_dev) { hipMalloc(&ts_dev, (256 + 512) * 8); hipMemset(ts_dev, 0, (256 + 512) * 8); }
        a.ts = ts_dev;
    }
#endif
    a.same_b = (h->bh1_same && !AFQ_KNOB_SET("AFQ_NO_SAME_B")) ? 1 : 0;
    a.b_real = (h->bh1_real && !AFQ_KNOB_SET("AFQ_NO_REAL_B")) ? 1 : 0;
    a.rem4 = (h->M > 96 && h->M <= 100 && !AFQ_KNOB_SET("AFQ_PF_NOREM")) ? 1 : 0;
    // hybrid column tiling: both spins with 17 .. 28 electrons and the same number of 4-column units in their second slot
    // (measured NEGATIVE at C3, round 4: 148.6 us against 145.5 us -- three 4x4x4 units need 9 MFMA + 3 add instructions where
    //  the padded 16x16x4 tile needs 3 + 1, and a wave that multiplies mostly units is bound by instruction issue, not by the
    //  matrix pipe; tuning builds only, AFQ_PF_HYB=1)
    a.hyb = 0;
    if (a.rem4 && h->na > 16 && h->nb > 16 && h->na <= 28 && h->nb <= 28 && (h->na - 13) / 4 == (h->nb - 13) / 4 &&
        PF_NW == 8 && AFQ_KNOB_SET("AFQ_PF_HYB"))
        a.hyb = (h->na - 13) / 4;
    // contiguous columns: one one-body matrix for both spins (the HS potential never depends on the spin), 48 < na + nb <= 56
    // (a third unit per row tile would unbalance the deal and push wave 7 over 256 registers)
    a.contig = 0; a.symcols = 0;
    if (a.rem4 && a.same_b && h->na > 16 && h->nb > 16 && h->nt > 48 && h->nt <= 56 && PF_NW == 8 && !a.hyb &&
        !AFQ_KNOB_SET("AFQ_PF_NOCONTIG")) {
        a.contig = 1;
        a.hyb = (h->nt - 48 + 3) / 4;
    }
    a.symcols = (a.contig && h->na == h->nb && !AFQ_KNOB_SET("AFQ_PF_NOSYM")) ? 1 : 0;
    a.closed_try = 0;           // (set below, once the deal is known)
    a.BH1 = h->BH1; a.vhs = h->vhs; a.phi = h->phi; a.alive = h->alive; a.zero16 = h->zero_page;
    a.n_closed = h->counters + 3;
    const int NCH = (h->M + 7) / 8;
    const size_t lds = (size_t)NCH * 8192 + (size_t)PF_D * 16384;
    static size_t lds_set[6][AFQ_MAX_DEVICES] = {{0}, {0}, {0}, {0}, {0}, {0}};
    const bool narrow = h->na <= 16 && h->nb <= 16;
    KernelTrace kt(h, AFQ_K_PROPAGATOR);
    {   // matrix-pipe flops of this launch: (order complex products by 3 multiplications + 2 one-body applications by 2 or
        // 3) x k-steps of 4 x existing 16 x 16 tiles (2048 flop per v_mfma_f64_16x16x4, 512 per 4x4x4 remainder unit)
        const int nrt_ = (h->M + 15) / 16, ct = (h->na + 15) / 16 + (h->nb + 15) / 16;
        const double ksteps = 2.0 * NCH;
        const double per_pass = ksteps * (2048.0 * (a.rem4 ? nrt_ - 1 : nrt_) * ct + 512.0 * (a.rem4 ? ct : 0));
        // hybrid tiling of the Taylor products: per spin one full slot + hyb units per row tile, four remainder units in all
        const double taylor_pass = a.contig ? ksteps * (2048.0 * (nrt_ - 1) * 3 + 512.0 * (nrt_ - 1) * a.hyb + 512.0 * 4)
                                   : a.hyb ? ksteps * (2048.0 * (nrt_ - 1) * 2 + 512.0 * (nrt_ - 1) * 2 * a.hyb + 512.0 * 4) : per_pass;
        h->issued_flops[AFQ_K_PROPAGATOR] = (3.0 * h->exp_order * taylor_pass + 2.0 * (a.b_real ? 2.0 : 3.0) * per_pass) * h->nw;
        h->prop_issued_open = h->issued_flops[AFQ_K_PROPAGATOR] / h->nw;
        // a closed-shell walker (afq_counters [3]) issues the alpha slots only in its Taylor products: two column slots x the
        // row tiles (+ two remainder units in the M <= 100 deal), one slot when narrow; its one-body passes multiply every slot.
        // Waves that repeat a tile (five row tiles, narrow) and row tiles that do not exist are not counted, as above
        const int cta = (h->na + 15) / 16;
        const double closed_pass = ksteps * (2048.0 * (a.rem4 ? nrt_ - 1 : nrt_) * cta + 512.0 * (a.rem4 ? cta : 0));
        h->prop_issued_closed = 3.0 * h->exp_order * closed_pass + 2.0 * (a.b_real ? 2.0 : 3.0) * per_pass;
    }
    // every tile of the deal present: wide with 5-7 row tiles (waves 4-7 own the tiles from 4 on) and two column tiles
    // per spin, or narrow with six row tiles
    const int nrt = (h->M + 15) / 16;
    const bool nofull = PF_NW != 8 || AFQ_KNOB_SET("AFQ_PF_NOFULL");
    const int full = nofull ? 0 : narrow ? (nrt == 6 ? 6 : 0) : (nrt >= 5 && h->na > 16 && h->nb > 16 ? nrt : 0);
    // closed-shell deals: one matrix for both spins (the chain then acts on the spin blocks alike), as many electrons of
    // either spin, and a deal without holes: the contiguous-column deal with twins in like slots (symcols), or two slots per spin
    a.closed_try = (a.same_b && h->na == h->nb && h->exp_order > 0 && full != 0 && PF_NW == 8 && !a.t4 &&
                    (a.contig ? a.symcols != 0 : a.hyb == 0) && !AFQ_KNOB_SET("AFQ_PF_NOCLOSED")) ? 1 : 0;
#define PF_LAUNCH_(NARROW_, FULL_, SLOT_)                                                                     \
    do {                                                                                                      \
        AFQ_HIP(h, afq_raise_lds((const void *)prop_fused_kernel<NARROW_, FULL_>, lds, lds_set[SLOT_]));      \
        AFQ_LAUNCH(h, (prop_fused_kernel<NARROW_, FULL_>), dim3(h->nw), dim3(PF_NT), lds, h->stream, a);      \
    } while (0)
    if (narrow && full) PF_LAUNCH_(true, 6, 0);
    else if (narrow) PF_LAUNCH_(true, 0, 1);
    else if (full == 7) PF_LAUNCH_(false, 7, 2);
    else if (full == 6) PF_LAUNCH_(false, 6, 3);
    else if (full == 5) PF_LAUNCH_(false, 5, 4);
    else PF_LAUNCH_(false, 0, 5);
#undef PF_LAUNCH_
    AFQ_POST(h);
#ifdef AFQ_TUNING
    if (a.ts && ++ts_launch == 30) {
        unsigned long long t[256 + 512];
        hipStreamSynchronize(h->stream);
        hipMemcpy(t, a.ts, sizeof(t), hipMemcpyDeviceToHost);
        fprintf(stderr, "PF_STAGE ticks: phi->T %lld | one-body %lld | Taylor", (long long)(t[129] - t[128]), (long long)(t[130] - t[129]));
        for (int n = 1; n <= h->exp_order && n <= 6; ++n) fprintf(stderr, " %lld", (long long)(t[130 + n] - t[129 + n]));
        for (int wv = 0; wv < 2; ++wv) {
            const unsigned long long *o = t + 144 + wv * 8;
            fprintf(stderr, "\nPF_BND wave %d: loop end -> barrier %lld | epilogue + T writes %lld | barrier %lld | next_chunk %lld | fragment reads %lld",
                    wv * 4, (long long)(o[1] - o[0]), (long long)(o[2] - o[1]), (long long)(o[3] - o[2]), (long long)(o[4] - o[3]), (long long)(o[5] - o[4]));
        }
        fprintf(stderr, "\n");
        fprintf(stderr, " | one-body + store %lld | total %lld\n", (long long)(t[137] - t[130 + h->exp_order]), (long long)(t[137] - t[128]));
        for (int wv = 0; wv < 8; ++wv) {
            const unsigned long long *o = t + 256 + (wv * 16 + 5) * 4, *o0 = t + 256 + 5 * 4, *o6 = t + 256 + (wv * 16 + 6) * 4;
            fprintf(stderr, "PF_ALL wave %d (product 3, chunk 5): start %+5lld  halfA %5lld  sync %5lld (released at %+5lld)  halfB %5lld  | chunk period %lld\n",
                    wv, (long long)(o[0] - o0[0]), (long long)(o[1] - o[0]), (long long)(o[2] - o[1]), (long long)(o[2] - o0[0]),
                    (long long)(o[3] - o[2]), (long long)(o6[0] - o[0]));
        }
        for (int wv = 0; wv < 2; ++wv)
            for (int c = 0; c < (h->M + 7) / 8; ++c) {
                const unsigned long long *o = t + (wv * 16 + c) * 4;
                fprintf(stderr, "PF_TS wave %d chunk %2d: halfA %5lld  sync %5lld  halfB %5lld  | since chunk start of wave 0: %lld\n",
                        wv * 4, c, (long long)(o[1] - o[0]), (long long)(o[2] - o[1]), (long long)(o[3] - o[2]),
                        (long long)(o[0] - t[c * 4]));
            }
    }
#endif
    return AFQ_OK;
}
