#!/usr/bin/env python3
"""Throughput of the other BASELINE configurations (C1 Hubbard 4x4, C2 UEG, C4 Hubbard 16x16 shard)
through the batched step loop, device RNG.  Prints one JSON line per config."""
import json
import os
import sys
import time

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pauxy_amd import systems, trial as trial_mod          # noqa: E402
from pauxy_amd.context import release_context              # noqa: E402
from pauxy_amd.qmc.afqmc import AFQMC                      # noqa: E402


def run(name, system, trial, nw, dt, steps, warmup, prop=None, npop=5):
    prop = dict(prop or {})
    prop.update({'device_rng': True, 'rng_seed': 7})
    options = {'qmc': {'timestep': dt, 'num_steps': 10, 'blocks': 10 ** 6, 'stabilise_freq': 10,
                       'pop_control_freq': npop, 'num_walkers': nw, 'rng_seed': 7},
               'propagator': prop, 'estimators': {'mixed': {'verbose': False}, 'write_file': False}}
    afqmc = AFQMC(options=options, system=system, trial=trial)
    dev = afqmc.psi.dev
    es = afqmc.run_batched(warmup, first_step=1, eshift=0.0)
    dev.sync()
    dev.enable_timers(False)
    t0 = time.perf_counter()
    afqmc.run_batched(steps, first_step=warmup + 1, eshift=es)
    dev.sync()
    el = time.perf_counter() - t0
    mixed = afqmc.estimators.estimators['mixed']
    last = mixed.blocks[-1]
    print(json.dumps({"config": name, "walkers": nw, "steps": steps, "ms_per_step": 1e3 * el / steps,
                      "walker_steps_per_sec": nw * steps / el, "last_block_ETotal": float(last[6].real),
                      "last_block_weight": float(last[2].real)}))
    release_context(system, trial)


if __name__ == "__main__":
    which = sys.argv[1:] or ["C1", "C2", "C4"]
    if "C5sd" in which:
        # BASELINE configs[4] sizes with a single-determinant RHF trial (the NOMSD trial is a "next" row)
        s = systems.synthetic_generic(400, 2000, (50, 50), seed=7)
        run("C5 sizes, single determinant: generic M=400 K=2000 50+50, 256 walkers", s,
            trial_mod.rhf_trial_generic(s), 256, 0.005, 10, 5)
    if "C5" in which:
        # BASELINE configs[4]: generic M=400, K=2000, 50+50 electrons, 4-determinant NOMSD trial (mixed estimator;
        # the reference cannot combine a multi-determinant trial with back-propagation)
        s = systems.synthetic_generic(400, 2000, (50, 50), seed=7)
        t0 = trial_mod.rhf_trial_generic(s)
        rng = numpy.random.RandomState(3)
        dets = numpy.array([t0.psi + (0.0 if d == 0 else 0.05) * (rng.rand(400, 100) + 1j * rng.rand(400, 100))
                            for d in range(4)])
        t = trial_mod.MultiDetTrial(s, (numpy.array([0.8, 0.3, 0.2, 0.1], dtype=complex), dets), init=t0.psi)
        run("C5 generic M=400 K=2000 50+50, NOMSD ndet=4, 256 walkers (one GPU's shard)", s, t, 256, 0.005, 10, 5)
    if "C1h" in which:
        s = systems.Hubbard(4, 4, 8, 8, 4.0)
        run("C1 Hubbard 4x4 U=4 8+8, 10 walkers, discrete Hirsch HS", s, trial_mod.uhf_trial_hubbard(s), 10, 0.01,
            400, 50, prop={'hubbard_stratonovich': 'discrete'})
    if "C1" in which:
        s = systems.Hubbard(4, 4, 8, 8, 4.0)
        run("C1 Hubbard 4x4 U=4 8+8, 10 walkers", s, trial_mod.uhf_trial_hubbard(s), 10, 0.01, 400, 50)
    if "C2" in which:
        s = systems.UEG(2.0, 7, 7, 4.0)
        run("C2 UEG rs=2 14e M=93 nq=750, 256 walkers", s, trial_mod.hartree_fock_ueg(s), 256, 0.005, 100, 20)
    if "C4" in which:
        s = systems.Hubbard(16, 16, 128, 128, 8.0)
        run("C4 Hubbard 16x16 U=8 128+128, 256 walkers (one GPU's shard)", s, trial_mod.uhf_trial_hubbard(s), 256,
            0.005, 20, 10)
