#!/usr/bin/env python3
"""Do the walkers of an RHF run stay closed-shell BIT FOR BIT (alpha block == beta block) on other paths of the library than the
C3 one (tools/closedness_probe.py)?  Usage: python tools/closedness_probe_sizes.py M N [K] [nw]  (e.g. 400 50 2000: the large-system
path -- ring GEMMs for the one-body / Taylor products, blocked Green's function, blocked re-orthogonalisation)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy                                               # noqa: E402
from pauxy_amd import _lib as L, systems, trial as tm      # noqa: E402
from pauxy_amd.qmc.afqmc import AFQMC                      # noqa: E402

M, N = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5 * M
nw = int(sys.argv[4]) if len(sys.argv) > 4 else 64
s = systems.synthetic_generic(M, K, (N, N), seed=7)
t = tm.rhf_trial_generic(s)
options = {'qmc': {'timestep': 0.005, 'num_steps': 10, 'blocks': 10 ** 6, 'stabilise_freq': 10, 'pop_control_freq': 5,
                   'num_walkers': nw, 'rng_seed': 7},
           'propagator': {'device_rng': True, 'rng_seed': 7}, 'estimators': {'mixed': {'verbose': False}, 'write_file': False}}
afqmc = AFQMC(options=options, system=s, trial=t)
dev = afqmc.psi.dev
e, first = 0.0, 1
for n in (1, 1, 3, 5, 10, 10):
    e = afqmc.run_batched(n, first_step=first, eshift=e)
    first += n
    phi = dev.get(L.F_PHI)
    same = sum(bool(numpy.array_equal(p[:, :N], p[:, N:])) for p in phi)
    print("M=%d N=%d+%d K=%d: after step %2d: %d of %d walkers closed, max |alpha - beta| = %.3e"
          % (M, N, N, K, first - 1, same, nw, numpy.abs(phi[:, :, :N] - phi[:, :, N:]).max()))
