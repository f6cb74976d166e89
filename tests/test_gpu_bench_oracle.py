"""bench.py's own number against the oracle.  The benchmark's population (BASELINE configs[2]: M = 100, K = 500, 25 + 25, 256
walkers, device Philox fields, `run_batched` with overlapped block boundaries, riding estimator terms, device comb) and the
ORACLE fed with the same stream restated on the host (tests/philox_ref.py) and the same comb uniforms: every block row must
agree -- the total energy of the last block is the `last_block_ETotal` of the bench line.  The oracle runs as a pool of
processes (a walker-step costs it ~45 ms): by default 60 steps; AFQ_BENCH_ORACLE_STEPS=580 is the schedule of the default
`python bench.py` (20 warm-up + 5 x 100 timed + 40 + 20 traced steps), whose result is kept under profiles/."""
import multiprocessing
import os

import numpy
import pytest

from oracle import afqmc_ref as ref

pytestmark = pytest.mark.gpu
MODEL = None          # the oracle's model: built by every worker for itself (_init_worker), and by the parent


def _build_model():
    import bench
    from pauxy_amd import systems, trial as trial_mod
    from pauxy_amd.propagation.setup import generic_propagator_arrays
    system = systems.synthetic_generic(bench.M, bench.K, (bench.N, bench.N), seed=7)
    trial = trial_mod.rhf_trial_generic(system)
    BH1, mf_shift = generic_propagator_arrays(system, trial, bench.DT)
    # (the HS potential as complex numbers: the oracle's V = chol . x then needs no conversion per call)
    return ref.RefModel('generic', bench.M, bench.N, bench.N, trial.psi, BH1, mf_shift, bench.DT,
                        hs_pot=system.hs_pot.astype(numpy.complex128), rchol=trial._rchol,
                        H1=system.H1.astype(complex), ecore=system.ecore), system, trial


def _init_worker():
    """Spawned (not forked: the parent holds a GPU context) single-threaded numpy process with its own copy of the model."""
    global MODEL
    MODEL = _build_model()[0]


def _propagate_segment(args):
    """steps [s0, s0 + len(xi)) for a chunk of walkers: re-orthogonalisation when due, propagation, weight cap, and the
    estimator terms of every step but the last (whose terms follow the comb).  Returns (walkers, partial sums per step)."""
    walkers, xi, s0, eshift_of_step, nstblz, eef, last_has_comb = args
    model = MODEL
    parts = []
    for j in range(xi.shape[0]):
        step = s0 + j
        if step % nstblz == 0:
            for w in walkers:
                detR = ref.reortho(w['phi'], model.na, model.nb, 0.0)
                w['detR'] = detR
                w['log_detR'] += numpy.log(detR)
                w['ot'] = w['ot'] / detR
                w['ovlp'] = w['ot']
        for iw, w in enumerate(walkers):
            if abs(w['weight']) > 1e-8:
                ref.propagate_walker_phaseless(model, w, xi[j, iw], eshift_of_step[j], True)
            if (abs(w['weight']) > w['total_weight'] * 0.10) and step > 1:
                w['weight'] = w['total_weight'] * 0.10
        if not (last_has_comb and j == xi.shape[0] - 1):
            est = numpy.zeros(10, dtype=numpy.complex128)
            ref.mixed_update(model, est, walkers, step, eef)
            parts.append(est)
    return walkers, parts


def _estimate_chunk(args):
    walkers, step, eef = args
    est = numpy.zeros(10, dtype=numpy.complex128)
    ref.mixed_update(MODEL, est, walkers, step, eef)
    return est


@pytest.mark.timeout(1500)
def test_bench_population_against_the_oracle_block_by_block(monkeypatch):
    global MODEL
    import bench
    from pauxy_amd.context import release_context
    from pauxy_amd.qmc.afqmc import AFQMC
    from tests.philox_ref import device_normals_fast
    nsteps_total = int(os.environ.get("AFQ_BENCH_ORACLE_STEPS", "60"))
    M, K, N, nw = bench.M, bench.K, bench.N, bench.NW_PER_GPU
    nsteps, nstblz, npop = bench.NSTEPS_BLOCK, bench.NSTBLZ, bench.NPOP
    assert nsteps_total % nsteps == 0
    MODEL, system, trial = _build_model()
    options = {'qmc': {'timestep': bench.DT, 'num_steps': nsteps, 'blocks': 10 ** 6, 'stabilise_freq': nstblz,
                       'pop_control_freq': npop, 'num_walkers': nw, 'rng_seed': 7},
               'propagator': {'device_rng': True, 'rng_seed': 7, 'rng_stream': 0},
               'estimators': {'mixed': {'verbose': False}, 'write_file': False}}
    afqmc = AFQMC(options=options, system=system, trial=trial)
    drawn = []
    host_random = numpy.random.random

    def random():
        x = host_random()
        drawn.append(x)
        return x

    monkeypatch.setattr(numpy.random, 'random', random)
    afqmc.run_batched(nsteps_total, first_step=1, eshift=0.0)
    got = numpy.array(afqmc.estimators.estimators['mixed'].blocks)
    afqmc.finalise()
    release_context(system, trial)
    # ---- the oracle
    walkers = [ref.new_walker(MODEL, numpy.asarray(trial.psi, dtype=complex).copy()) for _ in range(nw)]
    for w in walkers:
        w['total_weight'] = nw
    nproc = max(1, min(64, (os.cpu_count() or 2) - 1, nw))
    ctx = multiprocessing.get_context("spawn")
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):      # inherited by the spawned workers
        monkeypatch.setenv(var, "1")
    neqlb = int(2.0 / bench.DT)
    est = numpy.zeros(10, dtype=numpy.complex128)
    eshift_pair = numpy.array([0, 0], dtype=numpy.complex128)
    eshift = 0.0
    blocks = []
    rs = iter(drawn)
    chunks = numpy.array_split(numpy.arange(nw), nproc)
    with ctx.Pool(nproc, initializer=_init_worker) as pool:
        # step-0 pass (qmc/afqmc.py:214-221; folded into the first block: the driver is not verbose)
        for e_ in pool.map(_estimate_chunk, [([walkers[i] for i in c], 0, nsteps) for c in chunks]):
            est += e_
        step = 1
        while step <= nsteps_total:
            seg = npop - (step - 1) % npop                           # up to and including the next comb step
            xi = numpy.array([device_normals_fast(nw * K, 7, 0, s - 1).reshape(nw, K) for s in range(step, step + seg)])
            # the shift changes only behind a block: constant over a segment (a comb step never straddles a block end)
            shifts = [eshift] * seg
            res = pool.map(_propagate_segment, [([walkers[i] for i in c], xi[:, c], step, shifts, nstblz, nsteps, True)
                                                for c in chunks])
            walkers = [w for ws, _ in res for w in ws]
            for j in range(seg - 1):
                for _, parts in res:
                    est += parts[j]
            last = step + seg - 1
            ref.pop_control(MODEL, walkers, nw, next(rs))
            for e_ in pool.map(_estimate_chunk, [([walkers[i] for i in c], last, nsteps) for c in chunks]):
                est += e_
            for s in range(step, last + 1):                          # qmc/afqmc.py:246-250, step by step
                if s % nsteps == 0:
                    gs, eshift_pair = ref.block_reduce(est, nsteps)
                    blocks.append(gs)
                    est[:] = 0
                if s < neqlb:
                    eshift = eshift_pair[0].real
                else:
                    eshift += (eshift_pair[0].real - eshift)
            step = last + 1
    want = numpy.array(blocks)
    assert got.shape[0] == want.shape[0] == nsteps_total // nsteps
    scale = numpy.maximum(1.0, numpy.abs(want[:, :9]))
    err = numpy.abs(got[:, 1:10] - want[:, :9]) / scale
    print("bench population vs oracle: %d steps, %d blocks; last block ETotal device %.17g oracle %.17g; max rel. error of a "
          "block row entry %.2e" % (nsteps_total, want.shape[0], got[-1, 5].real, want[-1, 4].real, err.max()))
    # (chaotic amplification of the last-bit differences between the device's and the host's log / cos / sin: 1e-8 over
    #  tens of steps; the long schedule is held to 1e-6)
    assert err.max() < (1e-8 if nsteps_total <= 100 else 1e-6)
