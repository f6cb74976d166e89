#!/usr/bin/env python3
"""Duration of greens_small_kernel<true> at C3 sizes (256 walkers, M=100, 25+25) with phases switched off
(tuning build: AFQ_GREENS_DBG bit 1 = no Gauss-Jordan, 2 = no O^-1 phi^T phase, 4 = no overlap phase, 8 = the LDS
Gauss-Jordan instead of the register one).  Usage: AFQ_GREENS_DBG=<bits> python tools/greens_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy                                              # noqa: E402
from pauxy_amd import systems, trial as tm, _lib as L      # noqa: E402
from pauxy_amd.context import get_context, release_context  # noqa: E402

s = systems.synthetic_generic(100, 500, (25, 25), seed=7)
t = tm.rhf_trial_generic(s)
dev = get_context(s, t).dev
nw = 256
dev.walkers_alloc(nw)
rng = numpy.random.RandomState(1)
dev.set(L.F_PHI, t.psi[None] + 0.05 * (rng.rand(nw, 100, 50) + 1j * rng.rand(nw, 100, 50)))
for _ in range(20):
    dev.greens(fetch=False)
dev.sync()
dev.kernel_trace(True, [L.K_GREENS])
for _ in range(100):
    dev.greens(fetch=False)
dev.sync()
ms = dev.kernel_trace_get(L.K_GREENS)
print("AFQ_GREENS_DBG=%s: %d launches, mean %.1f us, min %.1f us" % (os.environ.get("AFQ_GREENS_DBG", "0"), len(ms),
                                                                    float(numpy.mean(ms)) * 1e3, float(numpy.min(ms)) * 1e3))
release_context(s, t)
