#!/usr/bin/env python3
"""Writes the inputs of the BASELINE configs[2] workload (the ones bench.py builds: synthetic generic Hamiltonian
Nbasis=100, Nchol=500, RHF trial 25+25, dt=0.005) as one raw binary file for tools/stress (no GPU is touched)."""
import os
import struct
import sys

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(path, M=100, K=500, N=25, dt=0.005):
    from pauxy_amd.systems import synthetic_generic
    from pauxy_amd.trial import rhf_trial_generic
    from pauxy_amd.propagation.setup import generic_propagator_arrays
    system = synthetic_generic(M, K, (N, N), seed=7)
    trial = rhf_trial_generic(system)
    BH1, mf = generic_propagator_arrays(system, trial, dt)
    with open(path, 'wb') as f:
        f.write(struct.pack('<4id', M, K, N, N, dt))
        for a, dtp in ((system.hs_pot, numpy.float64), (trial._rchol[:2 * N * M], numpy.complex128),
                       (system.H1.astype(complex), numpy.complex128), (trial.psi, numpy.complex128),
                       (BH1, numpy.complex128), (mf, numpy.complex128)):
            f.write(numpy.ascontiguousarray(a, dtype=dtp).tobytes())
    print("wrote %s (%.1f MB)" % (path, os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/afq_stress_c3.bin")
