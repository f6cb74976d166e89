"""Hamiltonian containers: the read-only arrays the hot path consumes.

These are *set-up* objects (built once on the host with numpy/scipy); the
device library receives their arrays through ``afq_set_system_*``.  They expose
the attribute names PAUXY's system classes expose, so either these objects or
genuine ``pauxy.systems.*`` objects can be handed to the propagator / walkers /
estimators in this package.

Reference (paths under /root/reference/pauxy):
  Generic  systems/generic.py:74-166, construct_h1e_mod :202-210
  Hubbard  systems/hubbard.py:46-104,148-155; hopping matrix systems/hubbard_holstein.py:214-268
  UEG      systems/ueg.py:43-191 (basis, q-vectors, index lists), :336-428 (iA/iB)
"""
import math

import numpy
import scipy.sparse


class Generic(object):
    """Generic ab-initio Hamiltonian in Cholesky form.

    h1e : [2, M, M]; chol : [M*M, K] (real or complex); ecore : float.
    """

    def __init__(self, nelec, h1e, chol, ecore=0.0, h1e_mod=None):
        self.name = "Generic"
        self.nup, self.ndown = nelec
        self.nelec = tuple(nelec)
        self.ne = self.nup + self.ndown
        self.H1 = numpy.asarray(h1e)
        self.nbasis = self.H1.shape[-1]
        self.chol_vecs = numpy.ascontiguousarray(chol)
        self.hs_pot = self.chol_vecs
        self.nchol = self.chol_vecs.shape[-1]
        self.nfields = self.nchol
        self.ecore = ecore
        self.sparse = False
        self.control_variate = False
        self.stochastic_ri = False
        self.exact_eri = False
        self.pno = False
        self.mu = None
        self.vol = 1.0
        self.ktwist = numpy.array([None])
        if h1e_mod is None:
            # systems/generic.py:202-210: v0_ij = 1/2 sum_{k,n} L[ik,n] L[jk,n]
            M = self.nbasis
            c3 = self.chol_vecs.reshape((M, M, -1))
            v0 = 0.5 * numpy.einsum('ikn,jkn->ij', c3, c3, optimize=True)
            h1e_mod = numpy.array([self.H1[0] - v0, self.H1[1] - v0])
        self.h1e_mod = h1e_mod


def _decode_basis(nx, ny, i):
    # systems/hubbard.py:278-301: i = i_x + n_x * i_y
    if ny == 1:
        return numpy.array([i % nx])
    return numpy.array([i % nx, i // nx])


def hubbard_kinetic(t, nx, ny, xpbc=True, ypbc=True):
    """Nearest-neighbour hopping matrix with periodic boundaries
    (systems/hubbard_holstein.py:214-268, zero twist).  Returns [2, M, M] float."""
    M = nx * ny
    T = numpy.zeros((M, M), dtype=float)
    for i in range(M):
        xy1 = _decode_basis(nx, ny, i)
        for j in range(i + 1, M):
            xy2 = _decode_basis(nx, ny, j)
            dij = abs(xy1 - xy2)
            if sum(dij) == 1:
                T[i, j] = -t
            if ny == 1 and dij[0] == nx - 1 and xpbc:
                T[i, j] += -t
            elif ny > 1 and dij[0] == nx - 1 and dij[1] == 0 and xpbc:
                T[i, j] += -t
            elif ny > 1 and dij[0] == 0 and dij[1] == ny - 1 and ypbc:
                T[i, j] += -t
    T = T + T.T
    return numpy.array([T, T])


class Hubbard(object):
    """2-D (or 1-D) Hubbard model, systems/hubbard.py:46-104."""

    def __init__(self, nx, ny, nup, ndown, U, t=1.0):
        self.name = "Hubbard"
        self.nx, self.ny = nx, ny
        self.nup, self.ndown = nup, ndown
        self.nelec = (nup, ndown)
        self.ne = nup + ndown
        self.t = t
        self.U = U
        self.nbasis = nx * ny
        self.T = hubbard_kinetic(t, nx, ny)
        self.H1 = self.T
        self.ecore = 0.0
        self.nfields = self.nbasis
        self.symmetric = False
        self.control_variate = False
        self.ktwist = numpy.array(None)
        self.vol = nx * ny
        self.mu = None
        # systems/hubbard.py:148-155
        v0 = 0.5 * U * numpy.eye(self.nbasis)
        self.h1e_mod = numpy.array([self.H1[0] - v0, self.H1[1] - v0])


class UEG(object):
    """Uniform electron gas in a plane-wave basis, systems/ueg.py:43-191,336-428."""

    def __init__(self, rs, nup, ndown, ecut):
        self.name = "UEG"
        self.nup, self.ndown = nup, ndown
        self.nelec = (nup, ndown)
        self.rs = rs
        self.ecut = ecut
        self.ktwist = numpy.zeros(3)
        self.control_variate = False
        self.mu = None
        self.thermal = False
        self.sparse = True
        self.diagH1 = True
        self.ne = nup + ndown
        self.ecore = 0.5 * self.ne * self._madelung()
        self.L = rs * (4.0 * self.ne * math.pi / 3.) ** (1 / 3.)
        self.vol = self.L ** 3.0
        self.kfac = 2 * math.pi / self.L
        (self.sp_eigv, self.basis, self.nmax) = self._sp_energies(self.kfac, ecut)
        self.shifted_nmax = 2 * self.nmax
        self.imax_sq = numpy.dot(self.basis[-1], self.basis[-1])
        ix = [self._map(k) for k in self.basis]
        self.lookup = numpy.zeros(max(ix) + 1, dtype=int)
        for i, b in enumerate(ix):
            self.lookup[b] = i
        self.nbasis = len(self.sp_eigv)
        (_, qvecs, self.qnmax) = self._sp_energies(self.kfac, 4 * ecut)
        self.qvecs = numpy.copy(qvecs[1:])          # omit q = 0
        self.vqvec = numpy.array([4 * math.pi / numpy.dot(self.kfac * q, self.kfac * q)
                                  for q in self.qvecs])
        self.nchol = len(self.qvecs)
        self.nfields = 2 * self.nchol
        T = numpy.diag(self.sp_eigv)
        self.H1 = numpy.array([T, T])
        h1e_mod = self._mod_one_body(T)
        self.h1e_mod = numpy.array([h1e_mod, h1e_mod])
        nlimit = self.nup
        (self.ikpq_i, self.ikpq_kpq) = self._index_lists(+1, nlimit)
        (self.ipmq_i, self.ipmq_pmq) = self._index_lists(-1, nlimit)
        (self.chol_vecs, self.iA, self.iB) = self._two_body_potentials()

    # -- helpers ----------------------------------------------------------
    def _madelung(self):
        c1 = -2.837297
        c2 = (3.0 / (4.0 * math.pi)) ** (1.0 / 3.0)
        return c1 * c2 / (self.ne ** (1.0 / 3.0) * self.rs)

    def _sp_energies(self, kfac, ecut):
        nmax = int(math.ceil(numpy.sqrt((2 * ecut))))
        spval, kval = [], []
        for ni in range(-nmax, nmax + 1):
            for nj in range(-nmax, nmax + 1):
                for nk in range(-nmax, nmax + 1):
                    spe = 0.5 * (ni ** 2 + nj ** 2 + nk ** 2)
                    if spe <= ecut:
                        kval.append([ni, nj, nk])
                        spval.append(kfac ** 2 * spe)
        spval = numpy.array(spval)
        ix = numpy.argsort(spval, kind='mergesort')
        return (spval[ix], numpy.array(kval)[ix], nmax)

    def _map(self, k):
        s = self.shifted_nmax
        return (k[0] + self.nmax) + s * (k[1] + self.nmax) + s * s * (k[2] + self.nmax)

    def lookup_basis(self, vec):
        if numpy.dot(vec, vec) <= self.imax_sq:
            ix = self._map(vec)
            if ix >= len(self.lookup):
                return None
            return self.lookup[ix]
        return None

    def _mod_one_body(self, T):
        h1e_mod = numpy.copy(T)
        fac = 1.0 / (2.0 * self.vol)
        for i, ki in enumerate(self.basis):
            for j, kj in enumerate(self.basis):
                if i != j:
                    q = self.kfac * (ki - kj)
                    h1e_mod[i, i] = h1e_mod[i, i] - fac * 4 * math.pi / numpy.dot(q, q)
        return h1e_mod

    def _index_lists(self, sign, nlimit):
        li, lk = [], []
        for q in self.qvecs:
            a, b = [], []
            for i, k in enumerate(self.basis[0:nlimit]):
                idx = self.lookup_basis(k + sign * q)
                if idx is not None:
                    a.append(i)
                    b.append(idx)
            li.append(numpy.array(a, dtype=numpy.int64))
            lk.append(numpy.array(b, dtype=numpy.int64))
        return li, lk

    def _scaled_density_operator(self, transpose):
        nq = len(self.qvecs)
        M = self.nbasis
        rows, cols, vals = [], [], []
        for iq, q in enumerate(self.qvecs):
            qs = self.kfac * q
            factor = ((math.pi / self.vol) / numpy.dot(qs, qs)) ** 0.5
            for i, k in enumerate(self.basis):
                kpq = self.lookup_basis(k + q)
                if kpq is not None:
                    if transpose:
                        rows.append(kpq + i * M)
                    else:
                        rows.append(kpq * M + i)
                    cols.append(iq)
                    vals.append(factor)
        return scipy.sparse.csc_matrix((vals, (rows, cols)), shape=(M * M, nq),
                                       dtype=numpy.complex128)

    def _two_body_potentials(self):
        rho_q = self._scaled_density_operator(False)
        rho_qH = self._scaled_density_operator(True)
        return (rho_q, 1j * (rho_q + rho_qH), -(rho_q - rho_qH))


def synthetic_generic(M, K, nelec, seed=7):
    """Synthetic generic Hamiltonian of SURVEY section 8(d):
    h1 = (R+R^T)/2, R~U(0,1); L_n = (A_n+A_n^T)/2, A_n~N(0,(0.1/M)^2);
    chol = L.reshape(K, M*M).T (real [M*M, K])."""
    rng = numpy.random.RandomState(seed)
    R = rng.random_sample((M, M))
    h1 = 0.5 * (R + R.T)
    A = rng.normal(scale=0.1 / M, size=(K, M, M))
    L = 0.5 * (A + A.transpose(0, 2, 1))
    chol = numpy.ascontiguousarray(L.reshape(K, M * M).T)
    return Generic(nelec, numpy.array([h1, h1]), chol, ecore=0.0)


def get_generic_integrals(filename):
    """-> (h1e [2,M,M], chol [M*M,K], h1e_mod [2,M,M], ecore) from a QMCPACK-style HDF5 Hamiltonian
    (pauxy/systems/utils.py:62-124 without the MPI shared-memory window: every rank = one GPU keeps
    its own copy on the device anyway)."""
    from pauxy_amd.utils.io import read_integrals
    hcore, chol, ecore = read_integrals(filename)
    h1e = numpy.array([hcore, hcore])
    M = hcore.shape[-1]
    c3 = numpy.asarray(chol).reshape((M, M, -1))
    v0 = 0.5 * numpy.einsum('ikn,jkn->ij', c3, c3, optimize=True)       # systems/generic.py:202-210
    return h1e, chol, numpy.array([h1e[0] - v0, h1e[1] - v0]), ecore


def get_system(sys_opts):
    """Option-driven constructor (pauxy/systems/utils.py:9-60) for the three systems on the device path."""
    name = sys_opts['name']
    if name == 'Generic':
        filename = sys_opts.get('integrals')
        if filename is None:
            raise ValueError("Generic system: 'integrals' file not specified")
        nup, ndown = sys_opts.get('nup'), sys_opts.get('ndown')
        if nup is None or ndown is None:
            raise ValueError("Generic system: number of electrons not specified")
        h1e, chol, h1e_mod, ecore = get_generic_integrals(filename)
        if numpy.iscomplexobj(chol) and numpy.abs(numpy.imag(chol)).max() == 0.0:
            chol = numpy.ascontiguousarray(numpy.real(chol))
        return Generic((nup, ndown), h1e, chol, ecore, h1e_mod=h1e_mod)
    if name == 'Hubbard':
        return Hubbard(sys_opts['nx'], sys_opts['ny'], sys_opts['nup'], sys_opts['ndown'], sys_opts['U'],
                       t=sys_opts.get('t', 1.0))
    if name == 'UEG':
        return UEG(sys_opts['rs'], sys_opts['nup'], sys_opts['ndown'], sys_opts['ecut'])
    raise ValueError("unrecognized system name {}".format(name))
