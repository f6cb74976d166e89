"""One-time host-side construction of the propagator constants.

These run once per calculation with numpy/scipy, exactly as the reference builds
them in its propagator constructors; their outputs (``BH1[2,M,M]``,
``mf_shift[K]``) are hot-path *inputs* uploaded with ``afq_set_propagator``.

Reference (paths under /root/reference/pauxy):
  propagation/generic.py:66-107, propagation/hubbard.py:392-402,456-467,
  propagation/planewave.py:21-25,39-55.
"""
import numpy
import scipy.linalg


def generic_propagator_arrays(system, trial, dt):
    """mf_shift = i hs_pot^T vec(G_up + G_dn) (generic.py:78-79);
    BH1[s] = expm(-dt/2 (h1e_mod[s] - i reshape(hs_pot mf_shift))) (generic.py:103-107)."""
    nb = system.nbasis
    if getattr(trial, 'ndets', 1) > 1:
        # generic.py:82-86: one contract_one_body per field; it is linear in the integrals, so one
        # contraction of hs_pot with the trial's (unnormalised) transition density does all fields
        gamma, denom = trial.one_body_density()
        mf_shift = 1j * numpy.dot(system.hs_pot.T, gamma.ravel()) / denom
    else:
        mf_shift = 1j * numpy.dot(system.hs_pot.T, (trial.G[0] + trial.G[1]).ravel())
    shift = 1j * system.hs_pot.dot(mf_shift).reshape(nb, nb)
    H1 = system.h1e_mod - numpy.array([shift, shift])
    BH1 = numpy.array([scipy.linalg.expm(-0.5 * dt * H1[0]), scipy.linalg.expm(-0.5 * dt * H1[1])])
    return BH1, mf_shift


def hubbard_propagator_arrays(system, trial, dt, charge_decomposition=True):
    nb = system.nbasis
    if charge_decomposition:
        iu_fac = 1j * system.U ** 0.5                                   # hubbard.py:376
        mf_shift = iu_fac * (numpy.diag(trial.G[0]) + numpy.diag(trial.G[1]))   # :402
        vi1b = iu_fac * numpy.diag(mf_shift)
        H1 = system.h1e_mod - numpy.array([vi1b, vi1b])                 # :394-395
    else:
        mf_shift = system.U ** 0.5 * numpy.diag(trial.G[0] - trial.G[1])        # :467
        I = numpy.eye(nb)
        vi1b = system.U ** 0.5 * numpy.diag(mf_shift)
        H1 = system.H1 + 0.5 * system.U * numpy.array([I, I]) - numpy.array([vi1b, vi1b])   # :458-460
    BH1 = numpy.array([scipy.linalg.expm(-0.5 * dt * H1[0]), scipy.linalg.expm(-0.5 * dt * H1[1])])
    return BH1, numpy.asarray(mf_shift, dtype=numpy.complex128)


def ueg_propagator_arrays(system, trial, dt):
    H1 = system.h1e_mod                                                 # planewave.py:52-55
    BH1 = numpy.array([scipy.linalg.expm(-0.5 * dt * H1[0]), scipy.linalg.expm(-0.5 * dt * H1[1])])
    return BH1.astype(numpy.complex128), numpy.zeros(system.nfields, dtype=numpy.complex128)
