#!/usr/bin/env python3
"""Duration of the re-orthogonalisation at C3 sizes (256 walkers, M=100, 25+25): back-to-back launches, wall clock.
Usage: [AFQ_RF_TS=1 in a tuning build: s_memtime phases of work-group 0] python tools/reortho_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy                                              # noqa: E402
from pauxy_amd import systems, trial as tm, _lib as L      # noqa: E402
from pauxy_amd.context import get_context, release_context  # noqa: E402
from pauxy_amd.propagation import setup                    # noqa: E402

s = systems.synthetic_generic(100, 500, (25, 25), seed=7)
t = tm.rhf_trial_generic(s)
dev = get_context(s, t).dev
BH1, mf = setup.generic_propagator_arrays(s, t, 0.005)
nw = 256
dev.walkers_alloc(nw)
dev.set_propagator(BH1, mf, 0.005)
rng = numpy.random.RandomState(1)
phi0 = t.psi[None] + 0.05 * (rng.rand(nw, 100, 50) + 1j * rng.rand(nw, 100, 50))
dev.set(L.F_PHI, phi0)
for _ in range(30):
    dev.reortho(fetch=False)
dev.sync()
t0 = time.perf_counter()
for _ in range(200):
    dev.reortho(fetch=False)
dev.sync()
print("re-orthogonalisation: %.1f us per call (all launches of it)" % ((time.perf_counter() - t0) / 200 * 1e6))
release_context(s, t)
