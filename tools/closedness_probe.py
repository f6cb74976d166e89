#!/usr/bin/env python3
"""How long the C3 walkers of the benchmark stay closed-shell BIT FOR BIT (alpha block == beta block): the condition of the
one-spin path of greens_small_kernel.  They start closed (RHF trial) and every operator of the step acts on both spins
alike, but the fused propagator's contiguous-column deal multiplies columns 48, 49 (beta columns 23, 24) as 4x4x4 MFMA units
and their alpha twins inside a 16x16x4 tile: last-bit differences appear in a few walkers per step (NEGATIVES.md)."""
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
