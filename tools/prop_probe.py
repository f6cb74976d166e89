#!/usr/bin/env python3
"""Duration of prop_fused_kernel at C3 sizes (256 walkers, M=100, 25+25, K=500) on freshly reset walkers, so that
tuning-build experiments that break the numbers (AFQ_PF_DBG) still time a full population.
Usage: [AFQ_PF_DBG=<bits>] python tools/prop_probe.py"""
import os
import sys

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
xi = rng.normal(size=(nw, 500))
dev.kernel_trace(True, [L.K_PROPAGATOR, L.K_VHS, L.K_FORCE_BIAS])
for it in range(40):
    dev.set(L.F_PHI, phi0)
    dev.set(L.F_WEIGHT, numpy.ones(nw))
    dev.propagate(xi, 0.0)
dev.sync()
ms = numpy.array(dev.kernel_trace_get(L.K_PROPAGATOR))[10:]
print("AFQ_PF_DBG=%s: %d launches, mean %.1f us, min %.1f us" % (os.environ.get("AFQ_PF_DBG", "0"), len(ms),
                                                                float(numpy.mean(ms)) * 1e3, float(numpy.min(ms)) * 1e3))
for name, kind in (('VHS GEMM', L.K_VHS), ('force-bias GEMM', L.K_FORCE_BIAS)):
    x = numpy.array(dev.kernel_trace_get(kind))[10:]
    if len(x):
        print("  %s (AFQ_GEMM_DBG=%s): mean %.1f us, min %.1f us" % (name, os.environ.get("AFQ_GEMM_DBG", "0"),
                                                                    float(numpy.mean(x)) * 1e3, float(numpy.min(x)) * 1e3))
release_context(s, t)
