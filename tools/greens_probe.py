#!/usr/bin/env python3
"""Duration of the Green's function kernel at C3 sizes (256 walkers, M=100, 25+25): back-to-back launches, wall clock.
Usage: [AFQ_GREENS_DBG=<bits>] python tools/greens_probe.py   (the ablation bits exist in tuning builds only)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy                                              # noqa: E402
from pauxy_amd import systems, trial as tm, _lib as L      # noqa: E402
from pauxy_amd.context import get_context, release_context  # noqa: E402
from pauxy_amd.propagation import setup                    # noqa: E402

if os.environ.get("PROBE_CONFIG", "C3") == "C2":       # UEG rs = 2, 7 + 7 electrons, M = 93 (BASELINE configs[1])
    s = systems.UEG(2.0, 7, 7, 4.0)
    t = tm.hartree_fock_ueg(s)
    BH1, mf = setup.ueg_propagator_arrays(s, t, 0.005)
    M_, nt_ = s.nbasis, 14
else:
    s = systems.synthetic_generic(100, 500, (25, 25), seed=7)
    t = tm.rhf_trial_generic(s)
    BH1, mf = setup.generic_propagator_arrays(s, t, 0.005)
    M_, nt_ = 100, 50
dev = get_context(s, t).dev
nw = 256
dev.walkers_alloc(nw)
dev.set_propagator(BH1, mf, 0.005)
rng = numpy.random.RandomState(1)
phi0 = t.psi[None] + 0.05 * (rng.rand(nw, M_, nt_) + 1j * rng.rand(nw, M_, nt_))
if os.environ.get("PROBE_CLOSED"):                       # closed-shell walkers: beta block = alpha block, bit for bit
    phi0[:, :, nt_ // 2:] = phi0[:, :, :nt_ // 2]
dev.set(L.F_PHI, phi0)
for rep in range(3):
    for _ in range(50):
        dev.greens(want_G=False, fetch=False)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(400):
        dev.greens(want_G=False, fetch=False)
    dev.sync()
    dt = (time.perf_counter() - t0) / 400
print("%s AFQ_GREENS_DBG=%s: %.1f us per launch" % (os.environ.get("PROBE_CONFIG", "C3"), os.environ.get("AFQ_GREENS_DBG", "0"), dt * 1e6))
release_context(s, t)
