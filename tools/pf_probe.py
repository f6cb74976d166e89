#!/usr/bin/env python3
"""Duration of prop_fused_kernel at C3 sizes for several Taylor orders (events inside the step loop).

order 0 isolates the two one-body stages plus prologue / epilogue; the slope in the order is the cost of one
Taylor product.  Usage: python tools/pf_probe.py [order ...]   (default: 6 0)
Environment knobs of the library apply (AFQ_NO_REAL_B, AFQ_NO_SAME_B, AFQ_NO_FUSED)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy                                              # noqa: E402
from pauxy_amd import systems, trial as tm, _lib as L      # noqa: E402
from pauxy_amd.qmc.afqmc import AFQMC                      # noqa: E402
from pauxy_amd.context import release_context              # noqa: E402

s = systems.synthetic_generic(100, 500, (25, 25), seed=7)
t = tm.rhf_trial_generic(s)
for order in [int(x) for x in (sys.argv[1:] or ['6', '0'])]:
    opt = {'qmc': {'timestep': 0.005, 'num_steps': 10, 'blocks': 10 ** 6, 'stabilise_freq': 10, 'pop_control_freq': 5,
                   'num_walkers': 256, 'rng_seed': 7},
           'propagator': {'device_rng': True, 'rng_seed': 7, 'expansion_order': order},
           'estimators': {'mixed': {'verbose': False}}}
    a = AFQMC(options=opt, system=s, trial=t)
    dev = a.psi.dev
    es = a.run_batched(10)
    dev.kernel_trace(True)
    a.run_batched(30, first_step=11, eshift=es)
    dev.sync()
    ms = dev.kernel_trace_get(L.K_PROPAGATOR)
    print("order %d: %d launches, %.1f us" % (order, len(ms), float(numpy.mean(ms)) * 1e3))
    dev.kernel_trace(False)
    release_context(s, t)
