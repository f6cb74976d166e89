#!/usr/bin/env python3
"""Large-system GEMM chain (M > 128): launch trace of the propagation for a closed-shell and for an open-shell population.
Usage: python tools/closed_large_probe.py [M N nw]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy                                                # noqa: E402
from pauxy_amd import _lib as L                             # noqa: E402
from pauxy_amd import systems, trial as tm                  # noqa: E402
from pauxy_amd.qmc.afqmc import AFQMC                       # noqa: E402

M, N, nw = (int(x) for x in (sys.argv[1:4] + ['200', '50', '128'][len(sys.argv) - 1:]))
s = systems.synthetic_generic(M, 2 * M, (N, N), seed=7)
t = tm.rhf_trial_generic(s)
options = {'qmc': {'timestep': 0.005, 'num_steps': 10, 'blocks': 10 ** 6, 'stabilise_freq': 10, 'pop_control_freq': 5,
                   'num_walkers': nw, 'rng_seed': 7},
           'propagator': {'device_rng': True, 'rng_seed': 7}, 'estimators': {'mixed': {'verbose': False}, 'write_file': False}}
afqmc = AFQMC(options=options, system=s, trial=t)
dev = afqmc.psi.dev
phi0 = dev.get(L.F_PHI).copy()
for kind in ("closed", "open"):
    phi = phi0.copy()
    if kind == "open":
        rng = numpy.random.RandomState(3)
        phi[:, :, N:] += 1e-3 * rng.rand(*phi[:, :, N:].shape)
    dev.set(L.F_PHI, phi)
    dev.set(L.F_OT, dev.calc_overlap())
    dev.set(L.F_WEIGHT, numpy.ones(nw))
    for i in range(6):
        if i == 2:
            dev.sync()
            dev.launch_trace(True)
        dev.propagate(None, 0.0)
    dev.sync()
    dev.launch_trace(False)
    tr = dev.launch_trace_get()
    out = dev.get(L.F_PHI)
    nclosed = sum(numpy.array_equal(p[:, :N], p[:, N:]) for p in out)
    print(kind, "walkers closed after 6 steps: %d of %d;" % (nclosed, nw),
          "  ".join("%s %.1f us" % (k[:28], 1e3 * ms / c) for k, (c, ms) in sorted(tr.items(), key=lambda kv: -kv[1][1])[:6]))
