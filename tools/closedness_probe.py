#!/usr/bin/env python3
"""How long the C3 walkers of the benchmark stay closed-shell BIT FOR BIT (alpha block == beta block): the condition of the
one-spin path of greens_small_kernel.  They start closed (RHF trial) and every operator of the step acts on both spins
alike -- provided the fused propagator keeps a column and its twin of the other spin in column slots that are multiplied by
the same code (PropFusedArgs::symcols; with the plain contiguous order 23 of 256 walkers lost the property in the first
step: NEGATIVES.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pauxy_amd import _lib as L
import bench
from pauxy_amd.qmc.afqmc import AFQMC
system, trial = bench.build_inputs()
options = {'qmc': {'timestep': 0.005, 'num_steps': 10, 'blocks': 10 ** 6, 'stabilise_freq': 10, 'pop_control_freq': 5, 'num_walkers': 256, 'rng_seed': 7},
           'propagator': {'device_rng': True, 'rng_seed': 7, 'rng_stream': 0}, 'estimators': {'mixed': {'verbose': False}, 'write_file': False}}
afqmc = AFQMC(options=options, system=system, trial=trial)
dev = afqmc.psi.dev
e = 0.0
first = 1
for n in (1, 1, 3, 5, 10, 30):
    e = afqmc.run_batched(n, first_step=first, eshift=e)
    first += n
    phi = dev.get(L.F_PHI)
    same = [numpy.array_equal(p[:, :25], p[:, 25:]) for p in phi]
    d = numpy.abs(phi[:, :, :25] - phi[:, :, 25:]).max()
    print("after step %d: %d of 256 walkers closed, max |alpha - beta| = %.3e" % (first - 1, sum(same), d))
    if first - 1 <= 2:
        diff = phi[:, :, :25] != phi[:, :, 25:]
        print("   differing entries per column:", diff.sum(axis=(0, 1)).tolist())
        print("   differing entries per row (first 12 / last 8):", diff.sum(axis=(0, 2))[:12].tolist(), diff.sum(axis=(0, 2))[-8:].tolist())
