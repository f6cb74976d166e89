"""Single-determinant trial wavefunction container (set-up only).

Exposes the attributes the hot path reads from PAUXY trial objects
(``psi[M, na+nb]``, ``G[2,M,M]``, ``GH``, ``init``, ``ndets``, ``coeffs``,
``_rchol``, ``rot_chol``/``rot_hs_pot``) so that either this class or a genuine
``pauxy.trial_wavefunction.MultiSlater`` (ndets == 1) / ``UHF`` /
``HartreeFock`` object can be passed to the propagator / walkers.

Reference (paths under /root/reference/pauxy):
  trial_wavefunction/multi_slater.py:17-97 (attributes), :370-409 (half_rotate),
  :422-448 (rot_chol / rot_hs_pot); trial_wavefunction/hartree_fock.py:46-94;
  trial_wavefunction/uhf.py:49-103,236-250 (mean-field iteration).
"""
import numpy
import scipy.linalg


def _gab_mod(A, B):
    # estimators/greens_function.py:41-73
    O = numpy.dot(B.T, A.conj())
    GHalf = numpy.dot(scipy.linalg.inv(O), B.T)
    G = numpy.dot(A.conj(), GHalf)
    return G, GHalf


class SingleDetTrial(object):
    def __init__(self, system, psi, init=None, name="MultiSlater"):
        self.name = name
        self.type = name
        na, nb = system.nup, system.ndown
        self.psi = numpy.array(psi, dtype=numpy.complex128)
        assert self.psi.shape == (system.nbasis, na + nb)
        self.coeffs = numpy.array([1.0 + 0j])
        self.ndets = 1
        Ga, Gha = _gab_mod(self.psi[:, :na], self.psi[:, :na])
        if nb > 0:
            Gb, Ghb = _gab_mod(self.psi[:, na:], self.psi[:, na:])
        else:
            Gb = numpy.zeros_like(Ga)
            Ghb = numpy.zeros((0, system.nbasis), dtype=numpy.complex128)
        self.G = numpy.array([Ga, Gb])
        self.GH = [Gha, Ghb]
        self.init = self.psi.copy() if init is None else numpy.array(init, dtype=numpy.complex128)
        self._nalpha, self._nbeta, self._nbasis = na, nb, system.nbasis
        self._rchol = None
        self._rot_hs_pot = None
        self._eri = None
        self._UVT = None
        self._mem_required = 0.0
        self.le_oratio = 1.0
        self.error = False
        if system.name == "Generic":
            self.half_rotate(system)

    def half_rotate(self, system, comm=None):
        """rchol[i*M+p, n] = sum_m conj(psi[m,i]) L[m,p,n], alpha block then beta
        (trial_wavefunction/multi_slater.py:402-409); stored complex128."""
        M, na, nb = self._nbasis, self._nalpha, self._nbeta
        chol = system.chol_vecs.reshape((M, M, -1))
        nchol = chol.shape[-1]
        rchol = numpy.zeros((M * (na + nb), nchol), dtype=numpy.complex128)
        rchol[:M * na] = numpy.tensordot(self.psi[:, :na].conj(), chol,
                                         axes=((0), (0))).reshape((na * M, nchol))
        rchol[M * na:] = numpy.tensordot(self.psi[:, na:].conj(), chol,
                                         axes=((0), (0))).reshape((nb * M, nchol))
        self._rchol = rchol
        self._rot_hs_pot = rchol
        self._mem_required = rchol.nbytes / (1024.0 ** 3)

    def rot_chol(self, idet=0, spin=None):
        stride = self._nbasis * (self._nalpha + self._nbeta)
        alpha = self._nbasis * self._nalpha
        if spin is None:
            return self._rchol[idet * stride:(idet + 1) * stride]
        if spin == 0:
            return self._rchol[idet * stride:idet * stride + alpha]
        return self._rchol[idet * stride + alpha:(idet + 1) * stride]

    rot_hs_pot = rot_chol


def rhf_trial_generic(system):
    """RHF-like trial for the synthetic generic Hamiltonian: lowest eigenvectors
    of h1e."""
    e, v = numpy.linalg.eigh(system.H1[0])
    na, nb = system.nup, system.ndown
    psi = numpy.zeros((system.nbasis, na + nb), dtype=numpy.complex128)
    psi[:, :na] = v[:, :na]
    psi[:, na:] = v[:, :nb]
    return SingleDetTrial(system, psi)


def hartree_fock_ueg(system):
    """Closed-shell Fermi sphere: identity columns
    (trial_wavefunction/hartree_fock.py:46,55-56)."""
    na, nb = system.nup, system.ndown
    I = numpy.eye(system.nbasis)
    psi = numpy.zeros((system.nbasis, na + nb), dtype=numpy.complex128)
    psi[:, :na] = I[:, :na]
    psi[:, na:] = I[:, :nb]
    return SingleDetTrial(system, psi, name="hartree_fock")


def uhf_trial_hubbard(system, ueff=0.4, nit_max=5000, alpha=0.5, deps=1e-8):
    """Unrestricted mean-field trial for the Hubbard model.

    Same fixed-point iteration as trial_wavefunction/uhf.py:236-250
    (H_up = T + ueff*diag(n_dn), H_dn = T + ueff*diag(n_up), linear density
    mixing), but started from a deterministic staggered density instead of the
    reference's ten random starts (uhf.py:188-196), so that bench inputs do not
    depend on an RNG stream.
    """
    M, na, nb = system.nbasis, system.nup, system.ndown
    nx = getattr(system, 'nx', M)
    stag = numpy.array([((i % nx) + (i // nx)) % 2 for i in range(M)], dtype=float)
    niup = (na / M) * (1.0 + 0.5 * (2 * stag - 1))
    nidn = (nb / M) * (1.0 - 0.5 * (2 * stag - 1))
    psi = numpy.zeros((M, na + nb), dtype=numpy.complex128)
    for it in range(nit_max):
        eu, vu = numpy.linalg.eigh(system.T[0] + numpy.diag(ueff * nidn))
        ed, vd = numpy.linalg.eigh(system.T[1] + numpy.diag(ueff * niup))
        psi[:, :na] = vu[:, :na]
        psi[:, na:] = vd[:, :nb]
        nu = numpy.sum(numpy.abs(psi[:, :na]) ** 2, axis=1)
        nd = numpy.sum(numpy.abs(psi[:, na:]) ** 2, axis=1)
        if (numpy.sum(numpy.abs(nu - niup)) / M < deps ** 0.5 and
                numpy.sum(numpy.abs(nd - nidn)) / M < deps ** 0.5 and it > 0):
            break
        niup = (1 - alpha) * nu + alpha * niup
        nidn = (1 - alpha) * nd + alpha * nidn
    return SingleDetTrial(system, psi, name="UHF")
