#!/usr/bin/env python3
"""The step kernels of an RHF run (closed-shell walkers) at M basis functions, N + N electrons, 256 walkers, K = 5 M:
launch trace of 40 steps.  With the tuning build, AFQ_PF_NOCLOSED=1 / AFQ_NO_CLOSED_EXX=1 switch the closed-shell paths off.
Usage: python tools/closed_deal_probe.py M N"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pauxy_amd import systems, trial as tm                 # noqa: E402
from pauxy_amd.qmc.afqmc import AFQMC                      # noqa: E402

M, N = int(sys.argv[1]), int(sys.argv[2])
s = systems.synthetic_generic(M, 5 * M, (N, N), seed=7)
t = tm.rhf_trial_generic(s)
options = {'qmc': {'timestep': 0.005, 'num_steps': 10, 'blocks': 10 ** 6, 'stabilise_freq': 10, 'pop_control_freq': 5,
                   'num_walkers': 256, 'rng_seed': 7},
           'propagator': {'device_rng': True, 'rng_seed': 7}, 'estimators': {'mixed': {'verbose': False}, 'write_file': False}}
afqmc = AFQMC(options=options, system=s, trial=t)
dev = afqmc.psi.dev
e = afqmc.run_batched(40, first_step=1, eshift=0.0)
dev.sync()
dev.counters(reset=True)
dev.launch_trace(True)
afqmc.run_batched(40, first_step=41, eshift=e)
dev.sync()
dev.launch_trace(False)
tr = dev.launch_trace_get()
took = int(dev.counters()[3])
line = "  ".join("%s %.1f" % (k[:22], 1e3 * ms / c) for k, (c, ms) in sorted(tr.items(), key=lambda kv: -kv[1][1])[:5])
blocks = afqmc.estimators.estimators['mixed'].blocks
print("M=%d N=%d+%d [%s]: %.1f us per step in traced launches; closed-shell deal on %d of %d walker steps; E %.17g | %s"
      % (M, N, N, " ".join(k + "=" + v for k, v in os.environ.items() if k.startswith("AFQ_") and k != "AFQ_LIBRARY") or "default",
         1e3 * sum(ms for c, ms in tr.values()) / 40, took, 40 * 256, float(blocks[-1][6].real) if blocks else float("nan"), line))
