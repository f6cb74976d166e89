#!/usr/bin/env python3
"""The fused propagator alone at the benchmark's shape (M = 100, K = 500, 25 + 25, 256 closed-shell walkers): 40 launches on
the same walkers (re-uploaded every time, so that timing ablations of a tuning build that give WRONG results -- AFQ_PF_DBG --
cannot kill the population), launch trace of the last 20.  With AFQ_PF_TS=1 the tuning build prints its stamps at launch 30.
Usage: [AFQ_LIBRARY=.../libafqmc_hip_tuning.so AFQ_PF_TS=1 AFQ_PF_DBG=n] python tools/prop_probe.py [nw]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy                                                # noqa: E402
from pauxy_amd import _lib as L                             # noqa: E402
from pauxy_amd import systems, trial as tm                  # noqa: E402
from pauxy_amd.qmc.afqmc import AFQMC                       # noqa: E402

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = systems.synthetic_generic(100, 500, (25, 25), seed=7)
t = tm.rhf_trial_generic(s)
options = {'qmc': {'timestep': 0.005, 'num_steps': 10, 'blocks': 10 ** 6, 'stabilise_freq': 10, 'pop_control_freq': 5,
                   'num_walkers': nw, 'rng_seed': 7},
           'propagator': {'device_rng': True, 'rng_seed': 7}, 'estimators': {'mixed': {'verbose': False}, 'write_file': False}}
afqmc = AFQMC(options=options, system=s, trial=t)
dev = afqmc.psi.dev
phi0 = dev.get(L.F_PHI).copy()
ot0 = dev.get(L.F_OT).copy()
for i in range(40):
    dev.set(L.F_PHI, phi0)
    dev.set(L.F_OT, ot0)
    dev.set(L.F_WEIGHT, numpy.ones(nw))
    if i == 20:
        dev.sync()
        dev.launch_trace(True)
    dev.propagate(None, 0.0)
dev.sync()
dev.launch_trace(False)
tr = dev.launch_trace_get()
print("  ".join("%s %.1f us" % (k[:40], 1e3 * ms / c) for k, (c, ms) in sorted(tr.items(), key=lambda kv: -kv[1][1])[:6]))
