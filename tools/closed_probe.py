#!/usr/bin/env python3
"""What the step kernels cost for ONE spin of the C3 walkers (M=100, K=500, 25 + 0 electrons, 256 walkers): the ceiling of a
closed-shell (alpha == beta) specialisation of BASELINE configs[2].  Usage: python tools/closed_probe.py [na nb]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy                                              # noqa: E402
from pauxy_amd import systems, trial as tm, _lib as L      # noqa: E402
from pauxy_amd.qmc.afqmc import AFQMC                      # noqa: E402

na, nb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (25, 0)
s = systems.synthetic_generic(100, 500, (na, nb), seed=7)
t = tm.rhf_trial_generic(s)
options = {'qmc': {'timestep': 0.005, 'num_steps': 10, 'blocks': 10 ** 6, 'stabilise_freq': 10, 'pop_control_freq': 5,
                   'num_walkers': 256, 'rng_seed': 7},
           'propagator': {'device_rng': True, 'rng_seed': 7}, 'estimators': {'mixed': {'verbose': False}, 'write_file': False}}
afqmc = AFQMC(options=options, system=s, trial=t)
dev = afqmc.psi.dev
e = afqmc.run_batched(40, first_step=1, eshift=0.0)
dev.sync()
dev.launch_trace(True)
afqmc.run_batched(40, first_step=41, eshift=e)
dev.sync()
dev.launch_trace(False)
tr = dev.launch_trace_get()
tot = 0.0
for k, (c, ms) in sorted(tr.items(), key=lambda kv: -kv[1][1])[:10]:
    print("%-40s launches %4d  avg %8.1f us  per step %8.1f us" % (k[:40], c, 1e3 * ms / c, 1e3 * ms / 40))
print("(%d, %d) electrons: sum of traced launches per step %.1f us" % (na, nb, 1e3 * sum(ms for c, ms in tr.values()) / 40))
