"""Green's-function free functions of pauxy/estimators/greens_function.py, evaluated by the device kernels.

``gab``, ``gab_mod``, ``gab_spin``, ``gab_mod_ovlp`` and ``gab_multi_det`` take bare Slater matrices
(no system object), so each call runs the batched Green's-function kernel of ``libafqmc_hip.so`` on a
scratch handle sized for the matrices at hand: the bra ``A`` plays the trial, the ket ``B`` the walker
(:5-38, :41-73, :75-80, :82-115, :117-160).  They are set-up / analysis utilities in the reference too
(trial construction, back-propagation); the per-step path uses the walker-batched entry points.
"""
import numpy

from pauxy_amd import _lib as L
from pauxy_amd.device import AfqDevice

_scratch = {}


def _device(M, na, nb, device_id=0):
    key = (M, na, nb, device_id)
    dev = _scratch.get(key)
    if dev is None:
        dev = AfqDevice(device_id)
        # any system fixes the dimensions; the Green's-function kernels only read psi and phi
        dev.set_system_hubbard(numpy.zeros((2, M, M), dtype=numpy.complex128), 0.0, na, nb)
        dev.walkers_alloc(1)
        _scratch[key] = dev
    return dev


def _greens(Aa, Ab, Ba, Bb, want_inverse=False):
    """One device Green's function for the spin blocks (Aa|Ab) <- trial, (Ba|Bb) <- walker."""
    M, na, nb = Aa.shape[0], Aa.shape[1], Ab.shape[1]
    dev = _device(M, na, nb)
    dev.set_trial(numpy.concatenate([Aa, Ab], axis=1))
    dev.set(L.F_PHI, numpy.concatenate([Ba, Bb], axis=1)[None])
    det = dev.greens(want_G=True)[0]
    G = dev.get(L.F_G)[0]
    gh = dev.get(L.F_GHALF)[0]
    inv = dev.inverse_overlap()[0][0] if want_inverse else None
    return G, gh, det, inv


def gab_mod(A, B):
    """(G, Ghalf) with G = conj(A) (B^T conj(A))^-1 B^T (greens_function.py:41-73)."""
    A = numpy.asarray(A, dtype=numpy.complex128)
    B = numpy.asarray(B, dtype=numpy.complex128)
    n = A.shape[1]
    G, gh, _, _ = _greens(A, A, B, B)
    return G[0], gh[:n]


def gab(A, B):
    """B (A^H B)^-1 A^H (greens_function.py:5-38) = transpose of gab_mod's G."""
    return gab_mod(A, B)[0].T.copy()


def gab_spin(A, B, na, nb):
    """greens_function.py:75-80: both spin blocks at once."""
    A = numpy.asarray(A, dtype=numpy.complex128)
    B = numpy.asarray(B, dtype=numpy.complex128)
    if nb == 0:
        G, gh = gab_mod(A[:, :na], B[:, :na])
        return numpy.array([G, numpy.zeros_like(G)]), [gh, numpy.zeros((0, A.shape[0]), dtype=numpy.complex128)]
    G, gh, _, _ = _greens(A[:, :na], A[:, na:], B[:, :na], B[:, na:])
    return G, [gh[:na], gh[na:]]


def gab_mod_ovlp(A, B):
    """(G, Ghalf, inv_O) with inv_O = (B^T conj(A))^-1 (greens_function.py:82-115)."""
    A = numpy.asarray(A, dtype=numpy.complex128)
    B = numpy.asarray(B, dtype=numpy.complex128)
    n = A.shape[1]
    G, gh, _, inv = _greens(A, A, B, B, want_inverse=True)
    return G[0], gh[:n], inv[0, :n, :n].copy()


def gab_multi_det(A, B, coeffs):
    """greens_function.py:117-160: sum_i c_i ovlp_i G_i^T-convention / sum_i c_i ovlp_i for a list of bras A[i]
    (one spin block; ``coeffs`` are assumed complex conjugated already)."""
    A = numpy.asarray(A, dtype=numpy.complex128)
    B = numpy.asarray(B, dtype=numpy.complex128)
    M, n = B.shape
    E = numpy.eye(M, dtype=numpy.complex128)[:, :n]       # second spin block with unit overlap: det = det O_a
    num, den = 0.0, 0.0
    for c, Ai in zip(coeffs, A):
        G, gh, det, _ = _greens(Ai, E, B, E)
        num = num + c * det * G[0]
        den = den + c * det
    return num / den


def release():
    for dev in _scratch.values():
        dev.close()
    _scratch.clear()
