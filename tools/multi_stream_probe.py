#!/usr/bin/env python3
"""Does splitting ONE GPU's population over several handles (= streams) pay?  The step is a chain of dependent kernels,
most of them one latency-bound work-group per walker; independent sub-populations on separate streams let the hardware
co-schedule work-groups of different kernels on one CU.  R handles x (nw / R) walkers on one GPU, in-process communicator
for the comb and the block reduction (afq_comm_init_local), device RNG, bench cadence.  Prints walker-steps/s per R."""
import json
import os
import sys
import time

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import afqmc_ref as ref                          # noqa: E402  (model container only)
from pauxy_amd import _lib as L                              # noqa: E402
from pauxy_amd import device as devmod                       # noqa: E402
from pauxy_amd import systems, trial as trial_mod            # noqa: E402
from pauxy_amd.propagation.setup import generic_propagator_arrays   # noqa: E402
from tests.helpers import make_device                        # noqa: E402

M, K, N, NW, DT = 100, 500, 25, 256, 0.005


def run(R, steps, warmup):
    s = systems.synthetic_generic(M, K, (N, N), seed=7)
    t = trial_mod.rhf_trial_generic(s)
    BH1, mf = generic_propagator_arrays(s, t, DT)
    model = ref.RefModel('generic', M, N, N, t.psi, BH1, mf, DT, hs_pot=s.hs_pot, rchol=t._rchol,
                         H1=s.H1.astype(complex), ecore=s.ecore)
    nw = NW // R
    devs = [make_device(model, nw) for _ in range(R)]
    for i, d in enumerate(devs):
        d.rng_seed(7, i)
        phi = numpy.broadcast_to(t.psi, (nw,) + t.psi.shape).copy()
        d.set(L.F_PHI, phi)
        d.set(L.F_OT, d.calc_overlap())
    if R > 1:
        devmod.comm_init_local(devs)
    eshift = 0.0

    def step(n):
        nonlocal eshift
        for d in devs:
            if n % 10 == 0:
                d.reortho(fetch=False)
            d.set_weight_cap(0.10 if n > 1 else 0.0, -1.0)
            d.propagate_begin(None)
        for d in devs:
            d.propagate_finish(eshift)
        if n % 5 == 0:
            if R > 1:
                devmod.popcontrol_comb_local(devs, 0.5 + 0.001 * (n % 400), NW, fetch=False)
            else:
                devs[0].popcontrol_comb(0.5 + 0.001 * (n % 400), NW, fetch=False)
        for d in devs:
            d.estimates_update(n % 10 == 0)
        if n % 10 == 0:
            if R > 1:
                devmod.estimates_allreduce_local(devs)
            est = [d.estimates_get(zero=True) for d in devs][0]
            eshift = float((est[7] / est[1]).real)
    for d in devs:
        d.estimates_update(True)
    for n in range(1, warmup + 1):
        step(n)
    for d in devs:
        d.sync()
    t0 = time.perf_counter()
    for n in range(warmup + 1, warmup + steps + 1):
        step(n)
    for d in devs:
        d.sync()
    el = time.perf_counter() - t0
    for d in devs:
        d.close()
    return {"handles": R, "walkers_per_handle": nw, "ms_per_step": 1e3 * el / steps, "walker_steps_per_sec": NW * steps / el,
            "eshift": eshift}


def run_threads(R, steps, warmup):
    """The same without anything across the handles (no comb, no reduction): one HOST THREAD per handle, so that the
    launch overhead of the handles overlaps (ctypes releases the GIL inside the library).  Upper bound of what independent
    sub-populations can gain from kernel concurrency."""
    import threading
    s = systems.synthetic_generic(M, K, (N, N), seed=7)
    t = trial_mod.rhf_trial_generic(s)
    BH1, mf = generic_propagator_arrays(s, t, DT)
    model = ref.RefModel('generic', M, N, N, t.psi, BH1, mf, DT, hs_pot=s.hs_pot, rchol=t._rchol,
                         H1=s.H1.astype(complex), ecore=s.ecore)
    nw = NW // R
    devs = [make_device(model, nw) for _ in range(R)]
    for i, d in enumerate(devs):
        d.rng_seed(7, i)
        d.set(L.F_PHI, numpy.broadcast_to(t.psi, (nw,) + t.psi.shape).copy())
        d.set(L.F_OT, d.calc_overlap())

    def loop(d, n0, n1):
        for n in range(n0, n1):
            if n % 10 == 0:
                d.reortho(fetch=False)
            d.set_weight_cap(0.10 if n > 1 else 0.0, float(NW))
            d.propagate(None, -133.7)
            if n % 5 == 0:
                d.popcontrol_comb(0.5 + 0.001 * (n % 400), nw, fetch=False)
            d.estimates_update(n % 10 == 0)
            if n % 10 == 0:
                d.estimates_get(zero=True)
        d.sync()

    def phase(n0, n1):
        th = [threading.Thread(target=loop, args=(d, n0, n1)) for d in devs]
        for x in th:
            x.start()
        for x in th:
            x.join()
    phase(1, warmup + 1)
    t0 = time.perf_counter()
    phase(warmup + 1, warmup + steps + 1)
    el = time.perf_counter() - t0
    for d in devs:
        d.close()
    return {"handles": R, "mode": "one host thread per handle, independent populations", "walkers_per_handle": nw,
            "ms_per_step": 1e3 * el / steps, "walker_steps_per_sec": NW * steps / el}


if __name__ == "__main__":
    if os.environ.get("AFQ_PROBE_THREADS"):
        for R in [int(x) for x in (sys.argv[1:] or ["1", "2", "4"])]:
            print(json.dumps(run_threads(R, 200, 40)), flush=True)
        sys.exit(0)
    for R in [int(x) for x in (sys.argv[1:] or ["1", "2", "4"])]:
        print(json.dumps(run(R, 200, 40)), flush=True)
