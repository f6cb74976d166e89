#!/usr/bin/env python3
"""Headline benchmark: whole-step phaseless-AFQMC throughput in walker-steps/s
on the BASELINE configs[2] workload (generic Hamiltonian, Nbasis=100, Nchol=500,
RHF trial with 25+25 electrons, 256 walkers per GPU), plus the fp64-MFMA roofline
fraction of the Cholesky exchange-energy kernel and the CPU-oracle baseline.

  python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: bench.py starts its N ranks itself,
                                                            as child processes, before anything touches the GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full AFQMC step over the rank's walker batch in the reference's
cadence (qmc/afqmc.py:223-255): re-orthogonalisation every 10 steps, comb
population control every 5, Green's function + local energy every 10, hybrid
weight update and estimator accumulation every step.  Auxiliary fields are drawn
on the device (Philox) inside the step.  Inputs are synthetic (SURVEY section 8d
generator, seed 7) and resident in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time

import numpy

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M, K, N = 100, 500, 25
NW_PER_GPU = 256
STRONG_TOTAL = 2048                  # strong scaling at 2048 walkers in total
DT = 0.005
NSTBLZ, NPOP, NSTEPS_BLOCK = 10, 5, 10
PEAK_F64_MFMA_TFLOPS = 78.6          # MI355X fp64 matrix peak


def exchange_flops_per_walker(M, K, na, nb):
    # 4 flops per real x complex MAC; T_s[x,i,j] costs K*N_s^2*M MACs per spin
    return 4.0 * K * M * (na * na + nb * nb)


def build_inputs():
    from pauxy_amd.systems import synthetic_generic
    from pauxy_amd.trial import rhf_trial_generic
    system = synthetic_generic(M, K, (N, N), seed=7)
    trial = rhf_trial_generic(system)
    return system, trial


# ---- the other BASELINE configurations (--config): one GPU's shard each, through the same step loop -----------------
CONFIGS = {
    # name: sizes (M, na, nb, K, walkers), time step, default steps / warm-up, what BASELINE.json calls it
    "C1": dict(M=16, na=8, nb=8, K=16, nw=10, dt=0.01, steps=400, warmup=50, kind="hubbard",
               label="4x4 Hubbard U=4 half filling, 10 walkers, phaseless continuous propagator (BASELINE configs[0])"),
    "C2": dict(M=93, na=7, nb=7, K=1500, nw=256, dt=0.005, steps=100, warmup=20, kind="ueg", nq=750,
               label="UEG rs=2, 14 electrons, ecut=4 (M=93 plane waves, 750 momentum transfers), 256 walkers "
                     "(BASELINE configs[1])"),
    "C4": dict(M=256, na=128, nb=128, K=256, nw=256, dt=0.005, steps=20, warmup=10, kind="hubbard",
               label="16x16 Hubbard U=8 half filling, UHF trial, 256 walkers = one GPU's shard of 2048 "
                     "(BASELINE configs[3])"),
    "C5sd": dict(M=400, na=50, nb=50, K=2000, nw=256, dt=0.005, steps=10, warmup=5, kind="generic",
                 label="generic Nbasis=400, Nchol=2000, 50+50 electrons, single-determinant RHF trial, 256 walkers "
                       "(BASELINE configs[4] sizes without the multi-determinant trial)"),
    "C5": dict(M=400, na=50, nb=50, K=2000, nw=256, dt=0.005, steps=10, warmup=5, kind="generic", ndet=4,
               label="generic Nbasis=400, Nchol=2000, 50+50 electrons, 4-determinant NOMSD trial, 256 walkers = one GPU's "
                     "shard (BASELINE configs[4])"),
}


def build_config(name):
    from pauxy_amd import systems, trial as trial_mod
    c = CONFIGS[name]
    if name == "C1":
        s = systems.Hubbard(4, 4, 8, 8, 4.0)
        return s, trial_mod.uhf_trial_hubbard(s)
    if name == "C2":
        s = systems.UEG(2.0, 7, 7, 4.0)
        return s, trial_mod.hartree_fock_ueg(s)
    if name == "C4":
        s = systems.Hubbard(16, 16, 128, 128, 8.0)
        return s, trial_mod.uhf_trial_hubbard(s)
    s = systems.synthetic_generic(c["M"], c["K"], (c["na"], c["nb"]), seed=7)
    t0 = trial_mod.rhf_trial_generic(s)
    if name == "C5sd":
        return s, t0
    rng = numpy.random.RandomState(3)
    dets = numpy.array([t0.psi + (0.0 if d == 0 else 0.05) * (rng.rand(c["M"], 100) + 1j * rng.rand(c["M"], 100))
                        for d in range(4)])
    return s, trial_mod.MultiDetTrial(s, (numpy.array([0.8, 0.3, 0.2, 0.1], dtype=complex), dets), init=t0.psi)


def work_table(c, b_real=False, psi_real=False, rchol_same=False, closed_large=0.0):
    """The ONE work model of both bench modes (the headline line and --config): for every launch afq_launch_trace can
    name (kernel names for plain launches, the launching function for the GEMM engines) the algorithmic work of ONE
    launch, SURVEY 8d conventions -- 8 flops per complex MAC, 4 per real-by-complex MAC, no padding.
      survey         SURVEY 8d's own count for the operation (operands priced as complex even where they happen to be real)
      real_operands  the same with 4 flops per MAC wherever one operand is real in THIS run (real BH1, real trial)
      cmac           flops one complex-by-complex MAC is priced at in both: 8, or 6 for a launch that executes
                     3-multiplication products AND would otherwise be priced above the peak (stated in its note)
    ``closed_large``: share of the walker steps that went through the large-system GEMM chain as closed-shell walkers (the
    device's count, afq_counters_ext [7]): their one-body and Taylor products multiply the alpha columns only, and that is the
    work priced (the survey count stays what the reference does for the same walkers).
    Launches that are bookkeeping have no entry."""
    M, na, nb, K, nw = c["M"], c["na"], c["nb"], c["K"], c["nw"]
    nt, ndet = na + nb, c.get("ndet", 1)
    cx = ndet > 1                                       # perturbed determinants make the half-rotated vectors complex
    ob = 4.0 if b_real else 8.0
    ov = 4.0 if psi_real else 8.0
    nn2 = na * na + nb * nb
    n3 = float(na ** 3 + nb ** 3)
    P = M * (M + 1) // 2
    fb_len = na if (ndet == 1 and na == nb and rchol_same) else nt
    exq_pairs = na * M * (na * M + 1) / 2.0 + nb * M * (nb * M + 1) / 2.0
    cl_note = ("; closed-shell walkers (%.0f %% of the walker steps): the alpha columns only, copied over the beta block behind "
               "the chain -- the executed columns are priced" % (100.0 * closed_large)) if closed_large > 0.0 else ""
    t = {
        "prop_fused_kernel": ("mfma", 8.0 * M * M * nt * 8 * nw, M * M * nt * (8.0 * 6 + ob * 2) * nw,
                              "B exp(V) B: 2 one-body + 6 Taylor products of M x M by M x (na+nb) per walker"),
        "prop_ueg_kernel": ("mfma", 8.0 * M * M * nt * 6 * nw, None, "exp(V) phi from per-walker coefficients: 6 products"),
        "ueg_step_kernel": ("mfma", 8.0 * M * M * nt * 6 * nw, None,
                            "force bias + fields + coefficients + B exp(V) B in one launch: the 6 products are the priced work, the "
                            "field part (latency bound) is in the time"),
        "k_apply_exponential": ("mfma", 8.0 * M * M * (nt - closed_large * nb) * nw, None,
                                "one Taylor product V T, both spins" + cl_note),
        # (one launch for both spins when they share one real matrix, else one per spin: the caller divides by the count)
        "onebody_spin": ("mfma", 8.0 * M * M * nt * nw, ob * M * M * (nt - closed_large * nb) * nw,
                         "BH1 phi: one application to both spins" + cl_note),
        "k_vhs_generic": ("mfma", 4.0 * P * K * nw, None, "HS potential, packed symmetric columns"),
        "force_bias_generic_impl": ("mfma", 4.0 * K * fb_len * M * nw, None,
                                    "force bias / Coulomb vectors, one real-B pass over %s" %
                                    ("Ghalf_a + Ghalf_b" if fb_len == na and na == nb else "both spins")),
        # multi-determinant force bias through the determinant-averaged G (k_force_bias_msd_gbar)
        "msd_gbar_fold GEMM": ("mfma", 8.0 * M * M * ndet * nt * nw, None,
                               "Gbar + Gbar^T on the upper triangle = M^2 ndet (na+nb) complex MACs per walker"),
        "msd_gbar_contract GEMM": ("mfma", 4.0 * K * P * nw, None, "packed hs_pot times the folded averaged G"),
        "msd_scale_ghalf_kernel": ("hbm", 2.0 * 16 * ndet * nt * M * nw, None, "reads every Ghalf_d, writes it scaled"),
        # complex Atil (complex half-rotated vectors) and a complex Ghalf: the kernel multiplies with 3-multiplication
        # products, priced at 6 flops per complex MAC (8 would put the launch above the peak)
        "launch_exx_quadratic": ("mfma", (6.0 if cx else 4.0) * exq_pairs * nw, None,
                                 "exchange energy as the quadratic form on the upper triangle of Atil (one determinant)%s" %
                                 ("; complex Atil: 3-multiplication products priced at 6 flops per complex MAC" if cx else "")),
        "exx_kernel": ("mfma", (8.0 if cx else 4.0) * K * M * nn2 * nw, None, "exchange energy, T intermediate"),
        "OvlpProb GEMM": ("mfma", 8.0 * nn2 * M * nw, ov * nn2 * M * nw, "phi^T conj(psi), both spins"),
        "GhalfProb GEMM": ("mfma", 8.0 * nn2 * M * nw, None, "O^-1 phi^T, both spins"),
        "GdiagProb GEMM": ("mfma", 8.0 * nn2 * M * nw, ov * nn2 * M * nw, "(psi O^-1) rowdot phi = diag G, both spins"),
        "GramProb GEMM": ("mfma", 8.0 * nn2 * M * nw, None, "Cholesky-QR Gram matrix, both spins"),
        "QProb GEMM": ("mfma", 8.0 * nn2 * M * nw, None, "Cholesky-QR Q = phi T, both spins"),
        "gj_mfma_kernel": ("mfma", 8.0 * n3 * nw, None, "blocked Gauss-Jordan inverse + determinant, 8 N^3 flops per matrix"),
        "gj_big_kernel (fallback pass)": ("valu", 8.0 * n3 * nw, None,
                                          "step-by-step Gauss-Jordan of the matrices the blocked kernel flagged"),
        "gj_big_kernel": ("valu", 8.0 * n3 * nw, None, "register-resident Gauss-Jordan, 8 N^3 flops per matrix"),
        "chol_mfma_kernel": ("mfma", 8.0 / 3.0 * n3 * nw, None,
                             "blocked Cholesky + inverse of the factor, (4/3 + 4/3) N^3 flops per Gram matrix"),
        "chol_linv_kernel": ("valu", 8.0 / 3.0 * n3 * nw, None, "Cholesky + inverse of the factor, one pivot at a time"),
        "greens_small_kernel": ("hbm", 2.0 * M * nt * 16 * nw, None, "reads phi, writes Ghalf (latency bound: Gauss-Jordan chain)"),
        "ueg_fields_kernel": ("hbm", (nt * M * 16 + 2 * K * 16 + 600 * 16) * nw, None,
                              "reads Ghalf, writes xbar, xs and the HS coefficients"),
    }
    return t


def launch_work(name, c, b_real=False, psi_real=False, rchol_same=False, closed_large=0.0):
    """(bound, survey work, real-operand work or None, note) of the launch `name`, or None (bookkeeping)."""
    table = work_table(c, b_real, psi_real, rchol_same, closed_large)
    for key in sorted(table, key=len, reverse=True):      # longest key first: "gj_big_kernel (fallback pass)" before "gj_big_kernel"
        if key in name:
            return table[key]
    return None


PEAK_F64_VALU_TFLOPS = 78.6          # MI355X_MICROARCH.md: fp64 vector = fp64 matrix peak on CDNA4
PEAK_HBM_TBS = 8.0


def config_cpu_baseline(name, system, trial, budget_s=15.0):
    """The oracle (per-walker numpy/scipy loop of oracle/afqmc_ref.py) on this host for the same configuration and
    cadence, one process with the BLAS threads of one socket, over a bounded sample: blocks of 10 steps for a few
    walkers until about `budget_s` seconds have passed (one warm-up block first)."""
    from oracle import afqmc_ref as ref
    from pauxy_amd.propagation import setup
    c = CONFIGS[name]
    M, na, nb, dt = c["M"], c["na"], c["nb"], c["dt"]
    if c.get("ndet", 1) > 1:
        return {"value": None, "unit": "walker-steps/s", "cores": 0, "kind": "port",
                "sample": "not timed: see the single-determinant configuration C5sd (the oracle's multi-determinant step "
                          "is ndet^2 single-determinant evaluations)"}
    if c["kind"] == "generic":
        BH1, mf = setup.generic_propagator_arrays(system, trial, dt)
        model = ref.RefModel('generic', M, na, nb, trial.psi, BH1, mf, dt, hs_pot=system.hs_pot, rchol=trial._rchol,
                             H1=system.H1.astype(complex), ecore=system.ecore)
    elif c["kind"] == "hubbard":
        if name == "C4":
            # this lattice has E ~ +700: with the reference's zero shift in the first block the weights fall by e^-35 per
            # block and a two-walker sample drops below the comb's 1e-8 threshold; a ten times smaller time step keeps the
            # sample alive and costs exactly the same arithmetic per step
            dt = dt / 10.0
        BH1, mf = setup.hubbard_propagator_arrays(system, trial, dt, True)
        model = ref.RefModel('hubbard', M, na, nb, trial.psi, BH1, mf, dt, U=system.U, H1=system.T.astype(complex))
    else:
        BH1, mf = setup.ueg_propagator_arrays(system, trial, dt)
        H1diag = numpy.array([numpy.diag(system.H1[0]).real, numpy.diag(system.H1[1]).real])
        model = ref.RefModel('ueg', M, na, nb, trial.psi, BH1, mf, dt, iA=system.iA, iB=system.iB, H1diag=H1diag,
                             vqvec=system.vqvec, vol=system.vol, ikpq_i=system.ikpq_i, ikpq_kpq=system.ikpq_kpq,
                             ipmq_i=system.ipmq_i, ipmq_pmq=system.ipmq_pmq, ecore=system.ecore)
    cpu_model, socket_cores, sockets = host_cpu()
    rng = numpy.random.RandomState(11)
    K = model.nfields

    def block(nw_):
        walkers = [ref.new_walker(model, trial.psi) for _ in range(nw_)]
        ref.run_afqmc(model, walkers, lambda s_, w: rng.normal(size=K), lambda s_: rng.random_sample(), NSTEPS_BLOCK, 1,
                      nstblz=NSTBLZ, npop_control=NPOP, energy_eval_freq=NSTEPS_BLOCK)
        return nw_ * NSTEPS_BLOCK
    import contextlib
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=socket_cores)
    except Exception:                                   # noqa: BLE001
        limit = contextlib.nullcontext()
    with limit:
        t0 = time.time()
        block(1)
        warm = time.time() - t0
        nw_ = 2 if warm * 2 < budget_s / 3 else 1
        done, t0 = 0, time.time()
        while time.time() - t0 < budget_s:
            done += block(nw_)
        el = time.time() - t0
    out = {"value": done / el, "unit": "walker-steps/s", "cores": socket_cores, "kind": "port",
           "cpu": "%s, %d cores/socket x %d sockets" % (cpu_model, socket_cores, sockets), "dt": dt,
           "sample": "%d walker-steps (blocks of %d walkers x %d steps in the bench cadence incl. the per-block energy "
                     "evaluation), numpy/scipy per-walker loop of oracle/afqmc_ref.py, %.1f s wall, BLAS threads of one socket"
                     % (done, nw_, NSTEPS_BLOCK, el)}
    if dt != c["dt"]:
        out["sample"] += ("; time step %g instead of the configured %g (same arithmetic per step; at the configured step the "
                          "weights of a two-walker sample fall below the comb's 1e-8 threshold within the first block)" % (dt, c["dt"]))
    return out


def run_config(args, name, state, comm=None, world=1, rank=0, backend="nccl"):
    """`--config C1|C2|C4|C5sd|C5`: the same harness on another BASELINE configuration: W warm-up steps, then `repeats`
    timed regions of EXACTLY --steps steps, each bracketed by barrier + synchronisation, MAX over the ranks, median
    reported; an extra pass under afq_launch_trace names every launch of the step and prices the dominant one against its
    roof.  `--gpus N`: weak scaling -- the configuration's population on EVERY GPU (BASELINE configs[3], [4] are 8-GPU
    configurations: 256 walkers per GPU = 2048), population control on the library-owned communicator as in the C3 line,
    with the device-timed cost of the exchange steps and every rank's traffic in the line."""
    import torch
    from pauxy_amd.qmc.afqmc import AFQMC
    from pauxy_amd.context import release_context
    c = CONFIGS[name]
    steps = args.steps if args.steps_given else c["steps"]
    warmup = args.warmup if args.warmup_given else c["warmup"]
    state["phase"] = "set-up"
    system, trial = build_config(name)
    options = {'qmc': {'timestep': c["dt"], 'num_steps': NSTEPS_BLOCK, 'blocks': 10 ** 6, 'stabilise_freq': NSTBLZ,
                       'pop_control_freq': NPOP, 'num_walkers': c["nw"] * world, 'rng_seed': 7},
               'propagator': {'device_rng': True, 'rng_seed': 7, 'rng_stream': rank},
               'estimators': {'mixed': {'verbose': False}, 'write_file': False}}
    if os.environ.get("AFQ_BENCH_DEVICE_COMM"):
        v = os.environ["AFQ_BENCH_DEVICE_COMM"]
        options['walkers'] = {'device_comm': False if v == '0' else True if v == 'auto' else v}
    afqmc = AFQMC(comm=comm, options=options, system=system, trial=trial)
    dev = afqmc.psi.dev
    state["dev"] = dev
    device_comm = bool(getattr(afqmc.psi, 'device_comm', False))
    comm_kind = getattr(afqmc.psi, 'device_comm_kind', '')
    comm_note = getattr(afqmc.psi, 'device_comm_error', '') or ''
    pc_label = population_control_label(afqmc, comm_kind, comm_note, world, backend)
    if world > 1 and backend == "nccl" and comm_kind != 'rccl':
        sys.stderr.write("bench.py: rank %d: population control runs as %r, not the RCCL + peer-window path (%s)\n"
                         % (rank, comm_kind or 'host-mediated', comm_note or 'no reason recorded'))

    def sync():
        dev.sync()
        torch.cuda.synchronize()
        if comm is not None:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()
    state["phase"] = "warm-up"
    eshift = afqmc.run_batched(warmup, first_step=1, eshift=0.0)
    sync()
    state["phase"] = "timed region"
    repeats = args.repeats if args.repeats else (5 if steps < 100 else 3 if steps < 500 else 1)
    regions, own, first = [], [], warmup + 1
    for _ in range(repeats):
        t0 = time.perf_counter()
        eshift = afqmc.run_batched(steps, first_step=first, eshift=eshift)
        sync()
        el_ = time.perf_counter() - t0
        own.append(el_)
        if comm is not None:
            import torch.distributed as dist
            t = torch.tensor([el_], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_ = float(t.item())
        regions.append(el_)
        first += steps
    order = sorted(range(repeats), key=lambda i: regions[i])
    imed = order[(repeats - 1) // 2]
    elapsed = regions[imed]
    rank_ms = [1e3 * own[imed] / steps]
    if comm is not None:
        gathered = numpy.zeros(world)
        comm.Allgather(numpy.array(rank_ms), gathered)
        rank_ms = gathered.tolist()
    state["phase"] = "launch trace"
    extra = 2 * NSTEPS_BLOCK
    flagged0 = int(dev.counters()[2])
    closed_large0 = int(dev.counters(n=8)[7])
    one_spin0 = int(dev.counters(n=8)[4])
    dev.launch_trace(True)
    eshift = afqmc.run_batched(extra, first_step=first, eshift=eshift)
    dev.sync()
    dev.launch_trace(False)
    trace = dev.launch_trace_get()
    flagged = int(dev.counters()[2]) - flagged0     # matrices the blocked Gauss-Jordan handed to the step-by-step kernel
    # walker steps that went through the large-system GEMM chain as closed-shell walkers (alpha columns only)
    closed_large = (int(dev.counters(n=8)[7]) - closed_large0) / float(extra * c["nw"])
    # exchange-energy evaluations that found the population closed-shell on the device and evaluated one spin (afq_counters_ext
    # [4] counts their walkers): two launches each, the second one returns at once
    one_spin_evals = (int(dev.counters(n=8)[4]) - one_spin0) // c["nw"]
    b_real = bool(numpy.abs(numpy.imag(afqmc.propagators.propagator.BH1)).max() == 0.0)
    psi_real = bool(numpy.abs(numpy.imag(numpy.asarray(trial.psi))).max() == 0.0)
    rc_ = numpy.asarray(getattr(trial, '_rchol', numpy.zeros((0, 1))))
    half = c["na"] * c["M"]
    rchol_same = bool(c["kind"] == "generic" and c.get("ndet", 1) == 1 and c["na"] == c["nb"] and
                      numpy.array_equal(rc_[:half], rc_[half:2 * half]))
    traffic_table, traffic_source = committed_traffic(name)
    rows = []
    for lname, (count, ms) in sorted(trace.items(), key=lambda kv: -kv[1][1]):
        row = {"launch": lname, "launches": count, "avg_ms": ms / count, "ms_per_step": ms / extra}
        w = launch_work(lname, c, b_real=b_real, psi_real=psi_real, rchol_same=rchol_same, closed_large=closed_large)
        if w and "fallback pass" in lname:
            # priced by the matrices it actually processed: every other work-group of the launch returns at once
            nmat = 2 * c["nw"] * count
            if flagged == 0:
                row["note"] = "second pass behind the blocked Gauss-Jordan: 0 of %d matrices flagged, nothing to price" % nmat
                w = None
            else:
                w = (w[0], w[1] * flagged / nmat, None, w[3] + " (%d of %d matrices)" % (flagged, nmat))
        if w:
            bound, work, work_real, note = w
            if lname == "onebody_spin":          # two applications per step, in `count / extra` launches per step
                work = work * 2.0 * extra / count
                work_real = work_real * 2.0 * extra / count if work_real else None
            t = ms / count * 1e-3
            if lname == "launch_exx_quadratic" and one_spin_evals > 0:
                # priced per EVALUATION: an evaluation of a closed-shell population is the alpha launch (one spin's work, counted
                # twice by energy_finish_kernel) plus the beta launch whose work-groups return at once
                evals = count - one_spin_evals
                share = 1.0 - 0.5 * one_spin_evals / evals
                work, work_real = work * share, (work_real * share if work_real else None)
                t = ms / evals * 1e-3
                row.update(evaluations=evals, avg_ms=ms / evals)
                note += ("; closed-shell population in %d of %d evaluations: one spin evaluated, counted twice (avg_ms = the alpha "
                         "launch + the beta launch that returns at once)" % (one_spin_evals, evals))
            if bound == "hbm":
                row.update(bound="hbm", achieved=work / t / 1e9, peak=PEAK_HBM_TBS * 1e3, unit="GB/s",
                           frac=work / t / 1e12 / PEAK_HBM_TBS, bytes_per_launch=work, note=note)
            else:
                # frac: the work THIS run needs -- 4 flops per MAC wherever an operand is real here (a real one-body matrix, a
                # real trial), 8 per complex MAC: a utilisation, never above 1.  frac_survey: SURVEY 8d's own count for the
                # operation, which prices those operands as complex -- an effective rate against that formulation (it may
                # exceed 1 where the kernel uses the real operand)
                need = work_real if work_real is not None else work
                row.update(bound=bound, achieved=need / t / 1e12, peak=PEAK_F64_MFMA_TFLOPS, unit="TFLOP/s",
                           frac=need / t / 1e12 / PEAK_F64_MFMA_TFLOPS, flops_per_launch=need, note=note)
                if need != work:
                    row.update(frac_survey=work / t / 1e12 / PEAK_F64_MFMA_TFLOPS, flops_per_launch_survey=work)
            tr = traffic_for(traffic_table, lname)
            if tr is not None:
                row["traffic"] = tr
            if bound != "hbm" and work_real is None and any(k in lname for k in K3M_LAUNCHES):
                # a kernel that multiplies complex by complex with 3 real products executes 6 flops per complex MAC.  Decided
                # from the kernel variant (both operands complex in this run), never from the measured value: `frac` prices what
                # the matrix pipe executes, the SURVEY convention (8 flops) is kept beside it
                row["frac_at_8_flops_per_complex_mac"] = row["frac"]
                for key in ("frac", "achieved", "flops_per_launch"):
                    row[key] *= 0.75
                row["complex_mac_flops"] = 6
                row["note"] += "; 3-multiplication complex products: priced at the 6 flops per complex MAC they execute"
            if not row["frac"] <= 1.0:
                raise RuntimeError("launch %r priced above its peak (frac %.3f): the timed launch cannot be doing the counted "
                                   "work -- fix its work model" % (lname, row["frac"]))
        rows.append(row)
    priced = [r for r in rows if "frac" in r]
    if not priced:
        raise RuntimeError("no launch of this configuration has a work model: %r" % [r["launch"] for r in rows[:6]])
    dom = max(priced, key=lambda r: r["ms_per_step"])
    mixed = afqmc.estimators.estimators['mixed']
    exchange_timing = comm_stats = comm_per_rank = None
    if world > 1:
        exchange_timing, comm_stats, comm_per_rank = measure_exchange(state, afqmc, dev, comm, world, device_comm,
                                                                      first + extra, eshift,
                                                                      stress=getattr(args, "exchange_stress", 0.0), rank=rank)
    out = {
        "metric": "walker_steps_per_sec", "value": c["nw"] * world * steps / elapsed, "unit": "walker-steps/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
        "ms_per_step_min": 1e3 * min(regions) / steps, "ms_per_step_max": 1e3 * max(regions) / steps,
        "repeats": repeats, "timed_regions_ms": [1e3 * r for r in regions],
        "statistic": "median of %d timed regions of %d steps each%s" % (repeats, steps, " (max over ranks per region)" if world > 1 else ""),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": c["label"] + ", dt=%g, reortho/10, comb/5, energy/10" % c["dt"] +
                               (" -- %d walkers on each of %d GPUs" % (c["nw"], world) if world > 1 else ""), "name": name,
                   "walkers_total": c["nw"] * world, "rng": "device-philox",
                   "backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
                   "population_control": pc_label,
                   "sizes": {k: c[k] for k in ("M", "na", "nb", "K", "nw")}},
        "population_control": pc_label,
        "comm_stats": comm_stats, "comm_stats_per_rank": comm_per_rank,
        "comm_probe": ("passed on every rank (%s)" % comm_kind) if device_comm else
                      (None if world == 1 else "no device communicator: " + (comm_note or "not requested")),
        "exchange_timing": exchange_timing,
        "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "per_rank": rank_ms},
        "last_block_ETotal": float(numpy.real(mixed.blocks[-1][5])) if mixed.blocks else None,
        # what the device counted in the traced pass (afq_counters_ext [7], [4]): tools/roofline_configs.py reads it
        "closed_shell": {"large_chain_walker_step_share": closed_large, "exchange_one_spin": one_spin_evals > 0},
        "roofline": {"bound": "hbm" if dom["bound"] == "hbm" else "mfma", "kernel": dom["launch"] + " (" + dom["note"] + ")",
                     "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"],
                     "frac_survey": dom.get("frac_survey"),
                     "traffic": dom.get("traffic"), "traffic_source": traffic_source if dom.get("traffic") is not None else None,
                     "kernel_ms": dom["avg_ms"], "launches": dom["launches"],
                     "measured": "extra pass of %d steps under afq_launch_trace right behind the timed regions" % extra,
                     "frac_is": "algorithmic work this run needs (8 flops per complex MAC, 4 where an operand is real in this "
                                "run, no padding) over the time; frac_survey: SURVEY 8d's count with every operand priced as complex",
                     "pipe": "fp64 vector ALU (same peak as the matrix pipe on gfx950)" if dom["bound"] == "valu" else None},
        "roofline_all": rows[:16],
        "step_ms_in_traced_launches": sum(r["ms_per_step"] for r in rows),
    }
    if not args.no_cpu_baseline and world == 1:
        state["phase"] = "cpu baseline"
        out["cpu_baseline"] = config_cpu_baseline(name, system, trial)
        if out["cpu_baseline"]["value"]:
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    state["dev"] = None
    release_context(system, trial)
    return out


# rocprofv3 names of the launches afq_launch_trace names by their launching function (GEMM engines)
# launches whose complex products are 3-multiplication (Karatsuba) products
K3M_LAUNCHES = ("k_apply_exponential", "msd_gbar_fold GEMM", "OvlpProb GEMM", "GhalfProb GEMM", "GramProb GEMM", "QProb GEMM",
                "onebody_spin", "prop_fused_kernel", "prop_ueg_kernel", "ueg_step_kernel", "launch_exx_quadratic")
# (tools/publish_profiles.py: short() keeps the problem's name up to "Prob")
ROCPROF_NAME = {"onebody_spin": "mfma_gemm_wg_kernel<OneBodyProb>", "k_apply_exponential": "mfma_gemm_wg_kernel<TaylorProb>",
                "k_vhs_generic": "mfma_gemm_wg_kernel<VhsProb>", "launch_exx_quadratic": "mfma_gemm_wg_kernel<ExxQProb>",
                "msd_gbar_fold GEMM": "mfma_gemm_wg_kernel<GbarSymProb>", "OvlpProb GEMM": "mfma_gemm_wg_kernel<OvlpProb>",
                "GhalfProb GEMM": "mfma_gemm_wg_kernel<GhalfProb>", "GdiagProb GEMM": "mfma_gemm_wg_kernel<GdiagProb>",
                "GramProb GEMM": "mfma_gemm_wg_kernel<GramProb>", "QProb GEMM": "mfma_gemm_wg_kernel<QProb>"}


def csrc_sha16():
    """Fingerprint of the kernel sources (pauxy_amd/csrc/*.hip, *.h, *.inc): tools/publish_profiles.py stores it with the PMC
    passes, so that a bench line can say whether the traffic it quotes was measured on THIS library's kernels."""
    import hashlib
    d = os.path.join(ROOT, "pauxy_amd", "csrc")
    hsh = hashlib.sha256()
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".h", ".inc")):
            with open(os.path.join(d, n), "rb") as f:
                hsh.update(n.encode() + b"\0" + f.read())
    return hsh.hexdigest()[:16]


def committed_traffic(config_name):
    """HBM-side bytes per launch of this configuration's kernels from the newest committed rocprofv3 --pmc passes
    (profiles/rNN_pmc_traffic.json, tools/profile_round.sh + tools/publish_profiles.py: separate FETCH_SIZE / WRITE_SIZE
    passes of this same command, FETCH_SIZE doubled on gfx950) -- ({kernel: bytes}, source) or ({}, None)."""
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", rnd + "_pmc_traffic.json")
        if not os.path.exists(path):
            continue
        with open(path) as f:
            d = json.load(f)
        table = d if config_name == "C3" else d.get("configs", {}).get(config_name)
        borrowed = ""
        if not table and config_name == "C5sd":
            # the single-determinant run of the C5 sizes has no passes of its own; its Taylor / HS-potential GEMMs are the
            # launches of the C5 run (same kernels, same operand sizes: the trial does not enter them)
            table = {k: v for k, v in d.get("configs", {}).get("C5", {}).items() if "TaylorProb" in k or "VhsProb" in k}
            borrowed = " of the C5 configuration: same kernel, same operand sizes"
        if table:
            sha = d.get("csrc_sha16")
            state = ("kernel sources unchanged since those passes" if sha == csrc_sha16() else
                     "STALE: the kernel sources have changed since those passes" if sha else "no source fingerprint recorded")
            return ({k: v["traffic_bytes_per_launch"] for k, v in table.items() if isinstance(v, dict) and
                     "traffic_bytes_per_launch" in v},
                    "%s: rocprofv3 --pmc passes of this command%s (not collected in this run; %s)"
                    % (os.path.relpath(path, ROOT), borrowed, state))
    return {}, None


def traffic_for(table, launch):
    if not table:
        return None
    for key, kname in ROCPROF_NAME.items():
        if key in launch:
            return table.get(kname)
    base = launch.split("<")[0].split(" ")[0]
    return table.get(launch, table.get(base))


def host_cpu():
    """(model name, physical cores of one socket, sockets) from /proc/cpuinfo."""
    model, cores, sockets = "unknown", None, set()
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("cpu cores") and cores is None:
                    cores = int(line.split(":", 1)[1])
                elif line.startswith("physical id"):
                    sockets.add(line.split(":", 1)[1].strip())
    except OSError:
        pass
    return model, cores or (os.cpu_count() or 1), max(1, len(sockets))


def cpu_worker(variant, window):
    """One single-threaded process of the 'socket-processes' CPU variant: oracle blocks of 2 walkers x 20 steps
    in the bench cadence until `window` seconds have passed; prints {"walker_steps": n, "seconds": t}."""
    from oracle import afqmc_ref as ref
    from pauxy_amd.propagation.setup import generic_propagator_arrays
    system, trial = build_inputs()
    BH1, mf = generic_propagator_arrays(system, trial, DT)
    hs = system.hs_pot.astype(numpy.complex128) if variant == "novcopy" else system.hs_pot
    model = ref.RefModel('generic', M, N, N, trial.psi, BH1, mf, DT, hs_pot=hs, rchol=trial._rchol,
                         H1=system.H1.astype(complex), ecore=system.ecore)
    rng = numpy.random.RandomState(11 + os.getpid() % 1000)

    def block(nw_, nblocks):
        walkers = [ref.new_walker(model, trial.psi) for _ in range(nw_)]
        ref.run_afqmc(model, walkers, lambda s, w: rng.normal(size=K), lambda s: rng.random_sample(), NSTEPS_BLOCK,
                      nblocks, nstblz=NSTBLZ, npop_control=NPOP, energy_eval_freq=NSTEPS_BLOCK)
        return nw_ * nblocks * NSTEPS_BLOCK
    block(1, 1)                                             # warm-up
    t0 = time.time()
    done = 0
    while time.time() - t0 < window:
        done += block(2, 2)
    print(json.dumps({"walker_steps": done, "seconds": time.time() - t0}))


def cpu_socket_processes(variant, nproc, window=8.0):
    """`nproc` concurrent single-threaded oracle processes (the reference's own parallelism is MPI over walkers,
    qmc/afqmc.py:167-176): aggregate walker-steps/s = sum over processes of steps / own window."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", variant, "--cpu-window", str(window)]
    t0 = time.time()
    procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for _ in range(nproc)]
    rate, steps, ok = 0.0, 0, 0
    for p in procs:
        out, _ = p.communicate()
        try:
            r = json.loads(out.strip().splitlines()[-1])
            rate += r["walker_steps"] / r["seconds"]
            steps += r["walker_steps"]
            ok += 1
        except Exception:                                   # noqa: BLE001
            pass
    return rate, steps, ok, time.time() - t0


def cpu_baseline(system, trial):
    """the oracle ("port": per-walker numpy/scipy loop of oracle/afqmc_ref.py, the
    reference's algorithm and call sequence) timed on this host over bounded samples of the same workload and
    cadence (reortho / 10, comb / 5, energy / 10; one warm-up block first):
      * socket-novcopy : BLAS threads = physical cores of ONE socket, hs_pot pre-cast to complex128 once (the
                         reference promotes the real 40 MB matrix to complex on every construct_VHS call,
                         propagation/generic.py:177; this variant removes that accident) -- 32 walkers x 20 steps
      * socket-faithful: same threads, the reference's behaviour                     -- 16 walkers x 10 steps
      * one-thread     : 1 BLAS thread, faithful                                     --  4 walkers x 10 steps
      * socket-processes(-novcopy): one single-threaded oracle process per physical core of one socket, all running
                         at once (the reference parallelises over walkers with MPI ranks, not with BLAS threads:
                         qmc/afqmc.py:167-176), each timing its own 8 s window of 2-walker blocks
    The value reported is the FASTEST variant, so the speed-up is neither won on the promotion copy nor on BLAS
    threads that small per-walker matrices cannot use."""
    from oracle import afqmc_ref as ref
    from pauxy_amd.propagation.setup import generic_propagator_arrays
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    BH1, mf = generic_propagator_arrays(system, trial, DT)
    cpu_model, socket_cores, sockets = host_cpu()

    def timed(hs_pot, nthreads, nw_cpu, nsteps):
        model = ref.RefModel('generic', M, N, N, trial.psi, BH1, mf, DT, hs_pot=hs_pot,
                             rchol=trial._rchol, H1=system.H1.astype(complex), ecore=system.ecore)
        rng = numpy.random.RandomState(11)

        def go(steps, nw_):
            walkers = [ref.new_walker(model, trial.psi) for _ in range(nw_)]
            t0 = time.time()
            ref.run_afqmc(model, walkers, lambda s, w: rng.normal(size=K), lambda s: rng.random_sample(),
                          NSTEPS_BLOCK, steps // NSTEPS_BLOCK, nstblz=NSTBLZ, npop_control=NPOP,
                          energy_eval_freq=NSTEPS_BLOCK)
            return time.time() - t0

        def both():
            go(NSTEPS_BLOCK, 2)                              # warm-up: one block of two walkers
            return go(nsteps, nw_cpu)
        if threadpool_limits is not None:
            with threadpool_limits(limits=nthreads):
                dt = both()
        else:
            dt = both()
        return nw_cpu * nsteps / dt, dt

    variants = []
    for name, hs, thr, nwc, nst in (
            ("socket-novcopy", system.hs_pot.astype(numpy.complex128), socket_cores, 32, 20),
            ("socket-faithful", system.hs_pot, socket_cores, 16, 10),
            ("one-thread", system.hs_pot, 1, 4, 10)):
        v, dt = timed(hs, thr, nwc, nst)
        variants.append({"variant": name, "value": v, "threads": thr, "walkers": nwc, "steps": nst, "seconds": dt})
    for name, var in (("socket-processes-novcopy", "novcopy"), ("socket-processes", "faithful")):
        rate, steps, okp, wall = cpu_socket_processes(var, socket_cores)
        variants.append({"variant": name, "value": rate, "threads": okp, "walkers": 2 * okp, "steps": steps // max(1, 2 * okp),
                         "seconds": wall, "processes": okp})
    best = max(variants, key=lambda r: r["value"])
    return {"value": best["value"], "unit": "walker-steps/s", "cores": int(best["threads"]), "kind": "port",
            "cpu": "%s, %d cores/socket x %d sockets" % (cpu_model, socket_cores, sockets),
            "sample": "%s: %d walkers x %d steps in the bench cadence (incl. step-0 and per-block energy evaluations), "
                      "numpy/scipy per-walker loop of oracle/afqmc_ref.py, %.1f s wall, %d thread(s) or process(es) on one socket"
                      % (best["variant"], best["walkers"], best["steps"], best["seconds"], best["threads"]),
            "variants": variants}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(ngpus, argv):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as CHILD processes (one per GPU, the
    environment contract of torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT,
    rendezvous on 127.0.0.1) before anything in this process touches the GPU -- the parent imports neither torch nor
    the library -- relay rank 0's JSON line(s) on stdout, everything else on stderr, and return the first non-zero
    exit status of a rank (the other ranks are then ended by their exact PIDs)."""
    import subprocess
    import threading
    port = str(free_port())
    procs = []
    for r in range(ngpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ngpus), LOCAL_WORLD_SIZE=str(ngpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // ngpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE, text=True))
    lines = [0]

    def relay(r, proc):
        for line in proc.stdout:
            if r == 0 and line.startswith("{"):
                sys.stdout.write(line)
                sys.stdout.flush()
                lines[0] += 1
            else:
                sys.stderr.write(line if r == 0 else "[rank %d] %s" % (r, line))
    threads = [threading.Thread(target=relay, args=(r, p), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    rc = 0
    live = set(range(ngpus))
    while live:
        for r in sorted(live):
            st = procs[r].poll()
            if st is None:
                continue
            live.discard(r)
            if st != 0 and rc == 0:
                rc = st if st > 0 else 128 - st
                sys.stderr.write("bench.py: rank %d exited with status %d; ending the other ranks\n" % (r, st))
                # a rank that is failing for the same reason gets a moment to say so itself before it is ended
                grace = time.time() + 10.0
                while time.time() < grace and any(procs[q].poll() is None for q in live):
                    time.sleep(0.05)
                for q in live:
                    if procs[q].poll() is None:
                        procs[q].terminate()
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5.0)
    if rc == 0 and lines[0] == 0:
        sys.stderr.write("bench.py: the ranks exited 0 but printed no result line\n")
        rc = 4
    return rc


def start_watchdog(state):
    # a hang anywhere (driver, runtime, collective) ends the run with every thread's traceback instead of
    # waiting for the caller's timeout
    import faulthandler
    import threading
    limit = float(os.environ.get("AFQ_BENCH_WATCHDOG_S", "900"))
    faulthandler.dump_traceback_later(limit + 10.0, exit=True)          # backstop if the thread below cannot run

    def watchdog():
        time.sleep(limit)
        dev = state["dev"]
        where = "no device yet"
        if dev is not None:
            try:
                where = dev.last_launch()[0]          # host memory only: safe while the main thread is stuck in a sync
            except Exception as e:                    # noqa: BLE001
                where = "afq_last_launch failed: %r" % (e,)
        sys.stderr.write("bench.py WATCHDOG after %.0f s in phase %r; library: %s\n" % (limit, state["phase"], where))
        sys.stderr.flush()
        faulthandler.dump_traceback(all_threads=True)
        os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["C3"] + sorted(CONFIGS), default="C3",
                    help="BASELINE configuration: C3 (default; the one the metric is quoted on: generic M=100 K=500, 256 "
                         "walkers per GPU, 1..8 GPUs) or one GPU's shard of C1 / C2 / C4 / C5 / C5sd")
    ap.add_argument("--walkers-per-gpu", type=int, default=None)
    ap.add_argument("--scaling", choices=["weak", "strong", "both"], default="weak",
                    help="weak: 256 walkers per GPU (BASELINE configs[2]); strong: 2048 walkers in total (SURVEY 8e); "
                         "both: one JSON line each, weak first")
    ap.add_argument("--repeats", type=int, default=None,
                    help="timed regions of --steps steps each (default: 5 below 100 steps, 3 below 500, else 1); "
                         "the median is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-rng", action="store_true", help="draw fields with numpy on the host (parity mode)")
    ap.add_argument("--exchange-stress", type=float, default=0.0,
                    help="N > 1: in the exchange-timing pass behind the timed regions, skew the weights between even and odd "
                         "ranks before every comb so that about this fraction of the walkers crosses ranks (0.1 = 10 %%)")
    ap.add_argument("--open-shell", action="store_true",
                    help="C3: start every walker with its beta block perturbed by 1e-3 (open-shell walkers: none of the "
                         "closed-shell paths of the Green's function, the exchange energy and the fused propagator applies)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-window", type=float, default=8.0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    args.steps_given, args.warmup_given = args.steps is not None, args.warmup is not None
    if args.steps is None:
        args.steps = 100
    if args.warmup is None:
        args.warmup = 20
    if args.cpu_worker:                 # child of cpu_socket_processes: numpy only, never touches the GPU
        cpu_worker(args.cpu_worker, args.cpu_window)
        return
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us (the driver's `python bench.py --gpus N`): this process becomes the launcher.
        # Nothing above has imported torch or the library, so no GPU state exists in the parent.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    state = {"dev": None, "phase": "start-up"}
    start_watchdog(state)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    # AFQ_BENCH_BACKEND=gloo runs the multi-rank code path with every rank on the visible GPUs modulo
    # their count (functional check of the N > 1 logic on a 1-GPU box; RCCL needs one GPU per rank)
    backend = os.environ.get("AFQ_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py: no GPU visible; the HIP library is the product and there is no CPU path")
    if backend == "nccl" and world > ndev:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible (RCCL needs one GPU per rank; "
                         "AFQ_BENCH_BACKEND=gloo shares GPUs for a functional check)" % (world, ndev))
    if backend != "nccl":
        local_rank = local_rank % ndev
        os.environ["LOCAL_RANK"] = str(local_rank)        # pauxy_amd.context picks the GPU from it
    torch.cuda.set_device(local_rank)
    comm = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from pauxy_amd.comm import TorchComm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            comm = TorchComm(device=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
            comm = TorchComm(device=torch.device("cpu"))
    if args.config != "C3":
        out = run_config(args, args.config, state, comm, world, rank, backend)
        if rank == 0:
            print(json.dumps(out))
            sys.stdout.flush()
        if comm is not None:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return
    system, trial = build_inputs()
    scalings = ["weak", "strong"] if args.scaling == "both" else [args.scaling]
    for i, scaling in enumerate(scalings):
        out = run_bench(args, scaling, comm, world, rank, backend, system, trial, state,
                        with_cpu_baseline=(i == 0 and not args.no_cpu_baseline and world == 1))
        if rank == 0:
            print(json.dumps(out))
            sys.stdout.flush()
    if comm is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def last_block_energy(afqmc):
    try:
        blocks = afqmc.estimators.estimators['mixed'].blocks
        # block row = [step, WeightFactor, Weight, ENumer, EDenom, ETotal, E1Body, E2Body, EHybrid, Overlap, Time]
        # (rounds 1-5 printed column 6 under this name: E1Body, which is the same number for every walker of a trial
        #  built from the eigenvectors of h1)
        return float(numpy.real(blocks[-1][5])) if blocks else None
    except Exception:
        return None


def population_control_label(afqmc, comm_kind, comm_note, world, backend):
    label = {'rccl': "device comb over the library's RCCL communicator (all-gather / all-reduce by RCCL, walkers written "
                     "straight into the destination GPU's mapped window over xGMI, live slots only)",
             'sendrecv': "device comb over the library's RCCL communicator (fixed-capacity ncclSend/ncclRecv slots)",
             'ipc': "device comb over mapped peer windows (hipIpc, no RCCL; bootstrap over torch.distributed)"}.get(
        comm_kind, "device comb (one rank)" if world == 1 else
        "host-mediated (torch.distributed, backend %s)" % backend)
    if comm_note and world > 1:
        label += " [fell through: " + comm_note + "]"
    return label


def measure_exchange(state, afqmc, dev, comm, world, device_comm, first_step, eshift, stress=0.0, rank=0):
    """Device-timed cost of the exchange steps (what the strong-scaling leg pays): two more blocks under
    afq_launch_trace (an event pair around every launch, on the library's stream), every launch that belongs to a
    population-control event or to the block reduction, averaged per event, on every rank.  A kernel that waits for a
    peer's flag (the plan kernel on the window collectives, the unpack kernel) is timed WITH that wait.
    ``stress`` > 0: before every comb of this pass the weights of the walkers of the even ranks are scaled up and those of
    the odd ranks down so that about that fraction of the population has to cross ranks (the benchmark's own weights
    hardly ever clone across ranks: an idle exchange is all the plain pass times).
    Returns (exchange_timing, comm_stats, comm_stats_per_rank)."""
    from pauxy_amd import _lib as L
    state["phase"] = "exchange-step timing"
    pc_steps = 2 * NSTEPS_BLOCK
    stats0 = dev.comm_stats() if device_comm else None
    dev.launch_trace(True)
    if stress > 0.0:
        # step by step: skew the weights right before each comb (host round trip: this pass is not a timed region)
        step, e_ = first_step, eshift
        for _ in range(pc_steps // NPOP):
            e_ = afqmc.run_batched(NPOP - 1, first_step=step, eshift=e_)
            wts = dev.get(L.F_WEIGHT)
            # fraction f of the population must move: the giving ranks hold (1 + f') of the mean, the others (1 - f')
            f_ = min(0.9, 2.0 * stress)
            wts *= (1.0 + f_) if rank % 2 == 0 else (1.0 - f_)
            dev.set(L.F_WEIGHT, wts)
            afqmc.psi._invalidate('weight')
            e_ = afqmc.run_batched(1, first_step=step + NPOP - 1, eshift=e_)
            step += NPOP
    else:
        afqmc.run_batched(pc_steps, first_step=first_step, eshift=eshift)
    dev.sync()
    dev.launch_trace(False)
    ltrace = dev.launch_trace_get()
    pc_names = ("comm_prep_kernel", "ncclAllGather", "comb_plan_global_kernel", "comb_plan_kernel", "clone_kernel",
                "comm_pack_kernel<true>", "ncclSend/Recv", "comm_pack_kernel<false>", "reset_kernel", "comm_exchange_kernel")
    red_names = ("ncclAllReduce", "est_put_kernel", "est_sum_kernel", "est_reduce_kernel")
    n_events, n_blocks = pc_steps // NPOP, pc_steps // NSTEPS_BLOCK
    pc_us = {k: 1e3 * ms / n_events for k, (cnt, ms) in ltrace.items() if any(k.startswith(p_) for p_ in pc_names)}
    pc_launches = {k: cnt / n_events for k, (cnt, ms) in ltrace.items() if any(k.startswith(p_) for p_ in pc_names)}
    red_us = {k: 1e3 * ms / n_blocks for k, (cnt, ms) in ltrace.items() if any(k.startswith(p_) for p_ in red_names)}
    exchange_timing = {"per_popcontrol_event_us": pc_us, "popcontrol_event_us": sum(pc_us.values()),
                       "launches_per_popcontrol_event": sum(pc_launches.values()),
                       "per_block_reduction_us": red_us, "block_reduction_us": sum(red_us.values()),
                       "ms_per_step_share": (sum(pc_us.values()) / NPOP + sum(red_us.values()) / NSTEPS_BLOCK) * 1e-3,
                       "measured": "extra pass of %d steps under afq_launch_trace after the timed regions; device time of the "
                                   "launches of one comb event (every %d steps) / one block reduction (every %d steps), "
                                   "waits for peers' flags included" % (pc_steps, NPOP, NSTEPS_BLOCK)}
    if stress > 0.0:
        exchange_timing["stress"] = ("weights skewed by +-%.0f %% between even and odd ranks before every comb of this pass"
                                     % (100.0 * min(0.9, 2.0 * stress)))
    if comm is not None:
        tot = numpy.zeros(2 * world)
        comm.Allgather(numpy.array([exchange_timing["popcontrol_event_us"], exchange_timing["block_reduction_us"]]), tot)
        exchange_timing["popcontrol_event_us_per_rank"] = tot[0::2].tolist()
        exchange_timing["block_reduction_us_per_rank"] = tot[1::2].tolist()
    comm_stats = dev.comm_stats() if device_comm else None
    comm_per_rank = None
    if device_comm and comm is not None:
        # every rank's traffic of the run: walkers it wrote into peers' windows, events, largest per-peer transfer, flags
        keys = ('walkers_sent', 'bytes_sent', 'events', 'max_transfer', 'overflow', 'error')
        mine = numpy.array([float(comm_stats[k]) for k in keys])
        allr = numpy.zeros(world * len(keys))
        comm.Allgather(mine, allr)
        allr = allr.reshape(world, len(keys))
        comm_per_rank = {k: [int(x) for x in allr[:, i]] for i, k in enumerate(keys)}
        ev = max(1, int(allr[0, 2]))
        comm_per_rank["walkers_sent_per_event"] = float(allr[:, 0].sum()) / ev
        if stats0 is not None:
            # ... and of this pass alone (what the per-event times above moved)
            d_w = float(comm_stats['walkers_sent'] - stats0['walkers_sent'])
            d_b = float(comm_stats['bytes_sent'] - stats0['bytes_sent'])
            tot2 = numpy.zeros(2 * world)
            comm.Allgather(numpy.array([d_w, d_b]), tot2)
            exchange_timing["walkers_sent_per_event_in_this_pass"] = float(tot2[0::2].sum()) / n_events
            exchange_timing["bytes_sent_per_event_in_this_pass"] = float(tot2[1::2].sum()) / n_events
    return exchange_timing, comm_stats, comm_per_rank


def run_bench(args, scaling, comm, world, rank, backend, system, trial, state, with_cpu_baseline):
    import torch
    from pauxy_amd.qmc.afqmc import AFQMC
    from pauxy_amd.context import release_context
    if args.walkers_per_gpu is not None:
        nw = args.walkers_per_gpu
    elif scaling == "strong":
        if STRONG_TOTAL % world:
            raise SystemExit("strong scaling: %d walkers do not divide over %d GPUs" % (STRONG_TOTAL, world))
        nw = STRONG_TOTAL // world
    else:
        nw = NW_PER_GPU
    repeats = args.repeats if args.repeats else (5 if args.steps < 100 else 3 if args.steps < 500 else 1)
    options = {
        'qmc': {'timestep': DT, 'num_steps': NSTEPS_BLOCK, 'blocks': 10 ** 6, 'stabilise_freq': NSTBLZ,
                'pop_control_freq': NPOP, 'num_walkers': nw * world, 'rng_seed': 7},
        'propagator': {'device_rng': not args.host_rng, 'rng_seed': 7, 'rng_stream': rank},
        'estimators': {'mixed': {'verbose': False}, 'write_file': False},      # in-memory block rows only
    }
    if os.environ.get("AFQ_BENCH_DEVICE_COMM"):
        # 'rccl' | 'sendrecv' | 'ipc' | '0': pin one communicator ('ipc' also works with every rank on one GPU over gloo)
        # ('auto': the whole candidate chain rccl -> sendrecv -> ipc -> host, also when the ranks' own process group is gloo)
        v = os.environ["AFQ_BENCH_DEVICE_COMM"]
        options['walkers'] = {'device_comm': False if v == '0' else True if v == 'auto' else v}
    state["phase"] = "set-up"
    afqmc = AFQMC(comm=comm, options=options, system=system, trial=trial)
    dev = afqmc.psi.dev
    state["dev"] = dev
    if getattr(args, "open_shell", False):
        from pauxy_amd import _lib as L_
        rng_o = numpy.random.RandomState(1000 + rank)
        phi_o = dev.get(L_.F_PHI)
        phi_o[:, :, N:] += 1e-3 * (rng_o.rand(*phi_o[:, :, N:].shape) + 1j * rng_o.rand(*phi_o[:, :, N:].shape))
        dev.set(L_.F_PHI, phi_o)
        afqmc.psi.phi_version += 1
        dev.set(L_.F_OT, dev.calc_overlap())
        afqmc.psi._invalidate()
    device_comm = bool(getattr(afqmc.psi, 'device_comm', False))
    comm_kind = getattr(afqmc.psi, 'device_comm_kind', '')
    comm_note = getattr(afqmc.psi, 'device_comm_error', '') or ''
    if world > 1 and backend == "nccl" and comm_kind != 'rccl':
        # never silently: the N > 1 number is meant to be the device comb over the library's RCCL communicator with the
        # walkers moving through mapped peer windows; anything else is said on stderr and in the result line
        sys.stderr.write("bench.py: rank %d: population control runs as %r, not the RCCL + peer-window path (%s)\n"
                         % (rank, comm_kind or 'host-mediated', comm_note or 'no reason recorded'))
        if os.environ.get("AFQ_BENCH_REQUIRE_DEVICE_COMM", "0") == "1":
            raise SystemExit(5)
    pc_label = population_control_label(afqmc, comm_kind, comm_note, world, backend)

    def barrier():
        dev.sync()
        torch.cuda.synchronize()
        if comm is not None:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    # Clock pre-heat (not AFQMC steps, walker state untouched): ~0.3 s of Green's function + local energy
    # evaluations so that the W warm-up steps and the timed region run at the sustained GPU clock even
    # when the process starts on an idle device.
    t_heat = time.perf_counter()
    while time.perf_counter() - t_heat < 0.3:
        dev.greens(want_G=False, fetch=False)
        for _ in range(8):
            dev.local_energy(fetch=False)
        dev.sync()
    state["phase"] = "warm-up"
    eshift = afqmc.run_batched(args.warmup, first_step=1, eshift=0.0)
    barrier()
    state["phase"] = "timed region"
    # Event pairs around a sample of the launches of the dominant kernel (fused propagator), read after the timed
    # region.  An event pair costs a few microseconds of pipeline bubble on either side of the launch: with every hot
    # kernel traced the step is 4 % slower, so the other kernels are timed in a separate, untimed pass below.
    from pauxy_amd import _lib as L
    in_region = [L.K_PROPAGATOR]
    # only a sample of the propagator's launches: the event pair around a launch opens a ~5.6 us bubble on either side of
    # it (rocprofv3 kernel trace), 3 % of the step if taken every step
    trace_stride = max(2, args.steps // 4)
    dev.kernel_trace_stride(L.K_PROPAGATOR, trace_stride)
    dev.kernel_trace(True, in_region)
    # `repeats` timed regions of EXACTLY --steps steps each, every one bracketed by barrier + synchronize on both sides
    # and reduced with MAX over the ranks; the median region is the one reported (a 20-step region is ~6 ms: one
    # sample of that is fragile evidence), all of them are listed.
    regions, own = [], []
    first = args.warmup + 1
    for _ in range(repeats):
        t0 = time.perf_counter()
        eshift = afqmc.run_batched(args.steps, first_step=first, eshift=eshift)
        barrier()
        elapsed = time.perf_counter() - t0
        own.append(elapsed)
        first += args.steps
        if comm is not None:
            import torch.distributed as dist
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        regions.append(elapsed)
    dev.kernel_trace(False)
    dev.kernel_trace_stride(L.K_PROPAGATOR, 1)
    state["phase"] = "after the timed region"
    order = sorted(range(repeats), key=lambda i: regions[i])
    imed = order[(repeats - 1) // 2]                     # lower median: an actually measured region
    elapsed = regions[imed]
    rank_ms = [1e3 * own[imed] / args.steps]
    if comm is not None:
        gathered = numpy.zeros(world)
        comm.Allgather(numpy.array(rank_ms), gathered)
        rank_ms = gathered.tolist()

    # Rooflines of the hot kernels from HIP events recorded on the library's stream around every launch
    # (afq_kernel_trace): the dominant kernel INSIDE the timed region, the GEMMs and the exchange energy in an extra pass
    # of 4 blocks right after it; algorithmic flops per launch as in DESIGN.md.
    traced = {kind: dev.kernel_trace_get(kind) for kind in in_region}
    extra_steps = 4 * NSTEPS_BLOCK
    dev.counters(reset=True)
    dev.kernel_trace(True)
    eshift = afqmc.run_batched(extra_steps, first_step=first, eshift=eshift)
    dev.sync()
    dev.kernel_trace(False)
    # walker steps the fused propagator took through its closed-shell deal (spin blocks bitwise equal, checked per walker in
    # the kernel: column slot 1 = [beta 0..15], the twin of slot 0, left out of the Taylor products)
    counted = dev.counters(n=8)
    closed_prop = float(counted[3]) / (extra_steps * nw)
    nt = 2 * N
    rc = numpy.asarray(trial._rchol)
    fb_same_spin_block = bool(numpy.array_equal(rc[:N * M], rc[N * M:2 * N * M]))
    b_real = bool(numpy.abs(numpy.imag(afqmc.propagators.propagator.BH1)).max() == 0.0)
    psi_real = bool(numpy.abs(numpy.imag(numpy.asarray(trial.psi))).max() == 0.0)
    # the same work model as the --config lines (work_table): SURVEY 8d count, and the count with real operands at 4 flops
    wt = work_table(dict(M=M, na=N, nb=N, K=K, nw=nw), b_real=b_real, psi_real=psi_real, rchol_same=fb_same_spin_block)
    quad = dev.exchange_algorithm() == 2
    # closed-shell population (every walker's spin blocks bitwise equal, RHF trial): the library evaluates the exchange energy
    # of one spin and counts it twice (verified on the device per evaluation) -- the executed work is one spin's.  The device
    # says so itself: afq_counters_ext [4] counts the walkers of every evaluation that took one spin
    one_spin_evals = int(counted[4]) // nw
    closed_pop = bool(quad and one_spin_evals > 0)
    if closed_pop and one_spin_evals != extra_steps // NSTEPS_BLOCK:
        raise RuntimeError("%d of the pass's %d exchange-energy evaluations took one spin: a population that changes kind inside "
                           "the traced pass has no one work model" % (one_spin_evals, extra_steps // NSTEPS_BLOCK))
    if closed_pop:
        b_, w_, wr_, n_ = wt["launch_exx_quadratic"]
        wt["launch_exx_quadratic"] = (b_, 0.5 * w_, wr_, n_ + "; closed-shell population: one spin evaluated, counted twice "
                                        "(the beta slices ride in the same launch as work-groups that return at once)")
    prop_note = ""
    prop_issued_scale = 0.0
    if closed_prop > 0.0:
        # executed work: the Taylor products of a closed-shell walker run on its alpha half (na of the na + nb columns; the
        # one-body passes multiply all of them); SURVEY's count (frac_survey) stays what the reference does for the same walkers
        b_, w_, wr_, n_ = wt["prop_fused_kernel"]
        ob_ = 4.0 if b_real else 8.0
        cols = nt - N * closed_prop
        wt["prop_fused_kernel"] = (b_, w_, M * M * (8.0 * 6 * cols + ob_ * 2 * nt) * nw, n_)
        prop_note = ("; closed-shell walkers (%.0f %% of the walker steps): Taylor products on the alpha half, %d of %d columns; "
                     "`frac` prices the executed columns" % (100.0 * closed_prop, N, nt))
        # matrix-pipe flops: the library's own count per walker for either deal (afq_propagator_issued_flops), mixed by the
        # device's count of the walkers that took the closed-shell deal
        open_w, closed_w = dev.propagator_issued_flops()
        prop_issued_scale = closed_prop * (closed_w - open_w) * nw
    kernels = [
        ("prop_fused_kernel (B exp(V) B, 2 one-body + 6 Taylor products per walker%s)" % prop_note, L.K_PROPAGATOR, wt["prop_fused_kernel"]),
        # Cholesky exchange energy.  Algorithm 2 (quadratic form g^T Atil g, one [nw x NM] x [NM x NM] real-by-complex
        # GEMM per spin) executes 4 (N M)^2 flops per spin and walker -- K / M = 5 times fewer than the
        # T-intermediate formulation of the reference (4 K M N^2); both counts are reported
        # (Atil is symmetric: the library stores its upper triangle and contracts only that -- NM (NM + 1) / 2 pairs)
        (("mfma_gemm_wg_kernel<ExxQProb> (Cholesky exchange energy as the quadratic form g^T Atil g, upper triangle%s)"
          % ("; closed-shell population: one spin evaluated, counted twice" if closed_pop else ""),
          L.K_EXCHANGE, wt["launch_exx_quadratic"]) if quad else
         ("exx_kernel (Cholesky exchange energy, T intermediate)", L.K_EXCHANGE, wt["exx_kernel"])),
        # symmetric Cholesky matrices: only the M(M+1)/2 columns p <= q are contracted
        ("mfma_gemm_wg_kernel<VhsProb> (HS potential, packed symmetric columns)", L.K_VHS, wt["k_vhs_generic"]),
        # force bias: both spins of an RHF-type trial contract with the same half-rotated Cholesky block, so the library
        # contracts Ghalf_a + Ghalf_b once (distributivity: half the flops of the two-spin contraction the reference
        # does, SURVEY 8d: 4 K (Na + Nb) M); the executed count is reported
        ("mfma_gemm_wg_kernel<ForceBiasProb> (force bias)", L.K_FORCE_BIAS, wt["force_bias_generic_impl"]),
    ]
    rows = []
    for name, kind, (_bound, flops, flops_real, _note) in kernels:
        live = kind in traced
        ms = traced[kind] if live else dev.kernel_trace_get(kind)
        if len(ms) == 0:
            continue
        avg = float(numpy.mean(ms))
        extra = {}
        if kind == L.K_EXCHANGE and dev.exchange_algorithm() == 2:
            ref_flops = exchange_flops_per_walker(M, K, N, N) * nw
            extra = {"reference_formulation_flops_per_launch": ref_flops,
                     "effective_vs_reference_formulation": ref_flops / (avg * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS}
        if kind == L.K_FORCE_BIAS and fb_same_spin_block:
            ref_flops = 4.0 * K * nt * M * nw
            extra = {"reference_formulation_flops_per_launch": ref_flops,
                     "effective_vs_reference_formulation": ref_flops / (avg * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS}
        # flops the matrix pipe actually executes per launch (MFMA instructions x their flop count, padding
        # included, 3-multiplication complex products counted as 3): what `frac_issued` = pipe utilisation is priced on
        issued = dev.kernel_issued_flops(kind)
        if kind == L.K_PROPAGATOR and closed_prop > 0.0:
            issued += prop_issued_scale                   # (the library reports the full deal's count; see above)
        if issued > 0:
            extra["issued_flops_per_launch"] = issued
            extra["frac_issued"] = issued / (avg * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS
        if flops_real is not None and flops_real != flops:
            # SURVEY 8d prices the two B phi products as complex by complex; BH1 is real in this run (real trial, real L_n) and
            # the kernel multiplies it as such: `frac` counts 4 flops per MAC there, frac_survey keeps SURVEY's count
            extra["flops_per_launch_survey"] = flops
            extra["frac_survey"] = flops / (avg * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS
            flops = flops_real
        rows.append({"kernel": name, "launches": int(len(ms)), "avg_ms": avg, **extra,
                     "measured": (("timed regions, every %d-th launch" % trace_stride if kind == L.K_PROPAGATOR
                                   else "timed regions") if live
                                  else "extra pass of %d steps after the timed regions" % extra_steps),
                     # (the propagator runs once per step; only a sample of its launches is timed in the region)
                     "ms_per_step": (avg if live and kind == L.K_PROPAGATOR else
                                     float(numpy.sum(ms)) / (args.steps * repeats if live else extra_steps)),
                     "flops_per_launch": flops,
                     "achieved": flops / (avg * 1e-3) / 1e12, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": flops / (avg * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS})
    if not rows:
        raise RuntimeError("no traced kernel launches in the timed region")
    dom = max(rows, key=lambda r: r["ms_per_step"])
    dominant_note = None
    if not dom["measured"].startswith("timed region"):
        if world > torch.cuda.device_count():
            # several ranks share a GPU (AFQ_BENCH_BACKEND=gloo: the functional check of the multi-rank path): a rank's
            # kernels run between those of the other ranks and their event-timed durations include the waiting -- which
            # kernel looks longest is noise.  The roofline stays on the kernel traced inside the timed region, and says so.
            dominant_note = ("ranks share a GPU: %s took longer per step in the extra pass (%.4f ms against %.4f), "
                             "durations include other ranks' kernels" % (dom["kernel"].split(" ")[0], dom["ms_per_step"],
                                                                        max(r["ms_per_step"] for r in rows if r["measured"].startswith("timed region"))))
            dom = max((r for r in rows if r["measured"].startswith("timed region")), key=lambda r: r["ms_per_step"])
        else:
            raise RuntimeError("dominant kernel %s was not traced inside the timed region" % dom["kernel"])
    for r in rows:
        if not (r["frac"] <= 1.0 and r.get("frac_issued", 0.0) <= 1.0):
            raise RuntimeError("kernel %r priced above the peak (frac %.3f, issued %.3f): the timed launch cannot be doing the "
                               "counted work" % (r["kernel"], r["frac"], r.get("frac_issued", 0.0)))
    # HBM-side bytes per launch from the committed rocprofv3 --pmc passes of this same command
    traffic_table, traffic_source = committed_traffic("C3")
    traffic = traffic_table.get(dom["kernel"].split(" ")[0])
    if traffic is None:
        traffic_source = None

    exchange_timing, comm_stats, comm_per_rank = measure_exchange(state, afqmc, dev, comm, world, device_comm,
                                                                  first + extra_steps, eshift,
                                                                  stress=getattr(args, "exchange_stress", 0.0), rank=rank)
    exq = next((r for r in rows if "exchange energy" in r["kernel"]), None)
    out = None
    if rank == 0:
        total_walkers = nw * world
        out = {
            "metric": "walker_steps_per_sec",
            "value": total_walkers * args.steps / elapsed,
            "unit": "walker-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_min": 1e3 * min(regions) / args.steps, "ms_per_step_max": 1e3 * max(regions) / args.steps,
            "repeats": repeats, "timed_regions_ms": [1e3 * r for r in regions],
            "statistic": "median of %d timed regions of %d steps each (max over ranks per region)" % (repeats, args.steps),
            "higher_is_better": True, "scaling": scaling if args.walkers_per_gpu is None else "weak",
            "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "generic Cholesky AFQMC: Nbasis=100, Nchol=500, RHF trial 25+25 electrons, "
                                   "%d walkers/GPU, dt=0.005, reortho/10, comb/5, energy/10 "
                                   "(BASELINE configs[2])%s" % (nw, "; OPEN-SHELL walkers (--open-shell: beta blocks perturbed by 1e-3)"
                                                                if getattr(args, "open_shell", False) else ""),
                       "walkers_total": total_walkers, "rng": "host-numpy" if args.host_rng else "device-philox",
                       "backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
                       "population_control": pc_label},
            "population_control": pc_label,
            "comm_stats": comm_stats,
            "comm_stats_per_rank": comm_per_rank,
            # Walkers._init_device_comm returns a communicator only after afq_comm_probe (known-answer all-gather, one
            # full slot to and from every peer on the chosen transport, all-reduce) passed on EVERY rank
            "comm_probe": ("passed on every rank (%s)" % comm_kind) if device_comm else
                          (None if world == 1 else "no device communicator: " + (comm_note or "not requested")),
            "exchange_timing": exchange_timing,
            "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "per_rank": rank_ms},
            # (device Philox streams: the same library gives the same number every run; kernel variants that claim
            #  bit-equal results can be held against it)
            "last_block_ETotal": last_block_energy(afqmc),
            # what the device counted in the traced pass (afq_counters_ext [3], [4]): tools/roofline_configs.py prices the
            # profiler's launches of the same run with it
            "closed_shell": {"propagator_walker_step_share": closed_prop, "exchange_one_spin": closed_pop,
                             "propagator_issued_flops_per_walker": dict(zip(("open", "closed"), dev.propagator_issued_flops()))},
            "roofline": {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved"],
                         "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac"],
                         "frac_survey": dom.get("frac_survey"),
                         "frac_issued": dom.get("frac_issued"),
                         "frac_is": "frac: the algorithmic work this run needs (8 flops per complex MAC; the real one-body "
                                    "propagator at 4 flops per MAC) over the time; frac_survey: SURVEY 8d's count (BH1 priced as "
                                    "complex: 8 x 16 M^2 N per walker); frac_issued: MFMA instructions x their flops, padding "
                                    "included, 3-multiplication products counted as 3 -- the share of the matrix pipe that is busy",
                         "traffic": traffic,
                         "traffic_source": traffic_source,
                         "kernel_ms": dom["avg_ms"], "launches": dom["launches"], "measured": dom["measured"],
                         "dominant_note": dominant_note,
                         "flops_per_launch": dom["flops_per_launch"],
                         "issued_flops_per_launch": dom.get("issued_flops_per_launch"),
                         # BASELINE's second figure: fp64 MFMA fraction of the Cholesky energy contraction
                         "cholesky_energy": None if exq is None else {
                             "kernel": exq["kernel"], "kernel_ms": exq["avg_ms"], "launches": exq["launches"],
                             "flops_per_launch": exq["flops_per_launch"], "achieved": exq["achieved"],
                             "frac": exq["frac"], "frac_issued": exq.get("frac_issued"),
                             "frac_vs_reference_formulation": exq.get("effective_vs_reference_formulation"),
                             "measured": exq["measured"]}},
            "roofline_all": rows,
        }
        if with_cpu_baseline:                                   # reported at N = 1 only
            state["phase"] = "cpu baseline"
            out["cpu_baseline"] = cpu_baseline(system, trial)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    state["dev"] = None
    release_context(system, trial)
    return out


if __name__ == "__main__":
    main()
