"""CPU oracle for the phaseless-AFQMC walker-propagation hot path.

TEST INFRASTRUCTURE ONLY.  This module is a from-scratch numpy/scipy
restatement of the reference algorithm (pauxy-qmc/pauxy); it is imported only
by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.  Nothing under ``pauxy_amd/`` imports it: the product path runs on
the HIP library or fails.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the genuine
reference (scratch copy of /root/reference/pauxy) in the build container and
stores its inputs/outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against those vectors and against the known-answer
constants lifted from the reference's own tests.

Each function cites the reference file:line it follows (paths relative to
/root/reference/pauxy).  All arrays are numpy, complex128 unless noted, C order.
A walker's Slater matrix is ``phi[M, na+nb]`` (alpha columns first).

The per-walker functions deliberately keep the reference's per-walker loop
structure and numpy/scipy call choices (scipy.linalg.inv / numpy.linalg.slogdet /
scipy.linalg.qr, one dense dot per operator) so that, timed on host cores, they
are a fair stand-in ("port") for the reference CPU path.
"""
import cmath
import math

import numpy
import scipy.linalg


# --------------------------------------------------------------------------
# Green's functions / overlaps
# --------------------------------------------------------------------------
def greens_function(phi, psi, na, nb, log_shift=0.0):
    """walkers/single_det.py:295-321 (same maths as estimators/greens_function.py:41-73).

    Returns (det, Ghalf[2], G[2,M,M]).  G is the "1-G" convention:
    G_s = conj(psi_s) . inv(phi_s^T conj(psi_s)) . phi_s^T
    """
    M = phi.shape[0]
    G = numpy.zeros((2, M, M), dtype=numpy.complex128)
    Ghalf = [None, None]
    O = numpy.dot(phi[:, :na].T, psi[:, :na].conj())
    Ghalf[0] = numpy.dot(scipy.linalg.inv(O), phi[:, :na].T)
    G[0] = numpy.dot(psi[:, :na].conj(), Ghalf[0])
    sign_a, logdet_a = numpy.linalg.slogdet(O)
    sign_b, logdet_b = 1.0, 0.0
    if nb > 0:
        O = numpy.dot(phi[:, na:].T, psi[:, na:].conj())
        sign_b, logdet_b = numpy.linalg.slogdet(O)
        Ghalf[1] = numpy.dot(scipy.linalg.inv(O), phi[:, na:].T)
        G[1] = numpy.dot(psi[:, na:].conj(), Ghalf[1])
    else:
        Ghalf[1] = numpy.zeros((0, M), dtype=numpy.complex128)
    det = sign_a * sign_b * numpy.exp(logdet_a + logdet_b - log_shift)
    return det, Ghalf, G


def calc_overlap(phi, psi, na, nb, log_shift=0.0):
    """walkers/single_det.py:170-199.

    The reference partitions with ``na = self.ndown`` (single_det.py:183), which
    is only right for na == nb; this restatement uses the correct partition and
    parity is claimed for na == nb (every BASELINE configuration).
    """
    Oa = numpy.dot(psi[:, :na].conj().T, phi[:, :na])
    sign_a, logdet_a = numpy.linalg.slogdet(Oa)
    sign_b, logdet_b = 1.0, 0.0
    if nb > 0:
        Ob = numpy.dot(psi[:, na:].conj().T, phi[:, na:])
        sign_b, logdet_b = numpy.linalg.slogdet(Ob)
    return sign_a * sign_b * numpy.exp(logdet_a + logdet_b - log_shift)


def gab(A, B):
    """estimators/greens_function.py:5-38."""
    inv_O = scipy.linalg.inv((A.conj().T).dot(B))
    return B.dot(inv_O.dot(A.conj().T))


def gab_mod(A, B):
    """estimators/greens_function.py:41-73."""
    O = numpy.dot(B.T, A.conj())
    GHalf = numpy.dot(scipy.linalg.inv(O), B.T)
    G = numpy.dot(A.conj(), GHalf)
    return (G, GHalf)


# --------------------------------------------------------------------------
# One-body propagation
# --------------------------------------------------------------------------
def kinetic_real(phi, BH1, na):
    """propagation/operations.py:29-52 (in place)."""
    phi[:, :na] = BH1[0].dot(phi[:, :na])
    phi[:, na:] = BH1[1].dot(phi[:, na:])


# --------------------------------------------------------------------------
# Force bias / HS potential, per system
# --------------------------------------------------------------------------
def force_bias_generic(Ghalf, rchol, na, nb, M, sqrt_dt, mf_shift):
    """propagation/generic.py:130-152 (dense branch)."""
    vbias = numpy.dot(rchol[:na * M].T, Ghalf[0].ravel())
    vbias = vbias + numpy.dot(rchol[na * M:(na + nb) * M].T, Ghalf[1].ravel())
    return -sqrt_dt * (1j * vbias - mf_shift)


def force_bias_generic_full(G, hs_pot, sqrt_dt, mf_shift):
    """propagation/generic.py:109-128 (construct_force_bias_slow)."""
    vbias = numpy.dot(hs_pot.T, G[0].ravel())
    vbias = vbias + numpy.dot(hs_pot.T, G[1].ravel())
    return -sqrt_dt * (1j * vbias - mf_shift)


def vhs_generic(hs_pot, xshifted, M, sqrt_dt):
    """propagation/generic.py:164-179."""
    VHS = hs_pot.dot(xshifted)
    return 1j * sqrt_dt * VHS.reshape(M, M)


def force_bias_hubbard(G, U, sqrt_dt, mf_shift):
    """propagation/hubbard.py:404-407 (charge decomposition)."""
    vbias = 1j * U ** 0.5 * (numpy.diag(G[0]) + numpy.diag(G[1]))
    return -sqrt_dt * (vbias - mf_shift)


def vhs_hubbard(xshifted, U, sqrt_dt):
    """propagation/hubbard.py:409-413: dense diagonal matrix."""
    return numpy.diag(sqrt_dt * 1j * U ** 0.5 * xshifted)


def force_bias_hubbard_spin(G, U, sqrt_dt, mf_shift):
    """propagation/hubbard.py:469-473 (spin decomposition)."""
    vbias = U ** 0.5 * numpy.diag(G[0] - G[1])
    return -sqrt_dt * (vbias - mf_shift)


def vhs_hubbard_spin(xshifted, U, dt):
    """propagation/hubbard.py:475-480: one diagonal per spin."""
    ut_fac = (dt * U) ** 0.5
    return numpy.array([numpy.diag(-ut_fac * xshifted),
                        numpy.diag(ut_fac * xshifted)])


def force_bias_ueg(G, iA, iB, sqrt_dt):
    """propagation/planewave.py:57-76.  iA, iB: scipy.sparse csc [M*M, nq]."""
    M = G.shape[-1]
    nq = iA.shape[1]
    Gvec = G.reshape(2, M * M)
    vbias = numpy.zeros(2 * nq, dtype=numpy.complex128)
    vbias[:nq] = Gvec[0].T * iA + Gvec[1].T * iA
    vbias[nq:] = Gvec[0].T * iB + Gvec[1].T * iB
    return -sqrt_dt * vbias


def vhs_ueg(iA, iB, xshifted, M, sqrt_dt):
    """propagation/planewave.py:94-112."""
    nq = iA.shape[1]
    VHS = iA * xshifted[:nq] + iB * xshifted[nq:]
    return sqrt_dt * VHS.reshape(M, M)


# --------------------------------------------------------------------------
# Two-body propagator
# --------------------------------------------------------------------------
def apply_exponential(phi, VHS, order=6):
    """propagation/continuous.py:82-111: truncated Taylor series, in place."""
    Temp = numpy.array(phi, copy=True)
    for n in range(1, order + 1):
        Temp = VHS.dot(Temp) / n
        phi += Temp
    return phi


def shift_fields(xi, xbar, mf_shift, sqrt_dt):
    """propagation/continuous.py:140-158: clip |xbar_i|>1 to unit modulus, then
    shifted field and the two constant factors.  Returns (xshifted, cmf, cfb, ntrig)."""
    xbar = numpy.array(xbar, dtype=numpy.complex128, copy=True)
    ntrig = 0
    for i in range(xbar.shape[0]):
        a = numpy.absolute(xbar[i])
        if a > 1.0:
            ntrig += 1
            xbar[i] /= a
    xshifted = xi - xbar
    cmf = -sqrt_dt * xshifted.dot(mf_shift)
    cfb = xi.dot(xbar) - 0.5 * xbar.dot(xbar)
    return xshifted, cmf, cfb, ntrig


def apply_bound_hybrid(ehyb, eshift, ebound):
    """propagation/continuous.py:202-214.  Returns (ehyb, triggered)."""
    if abs(eshift) < 1e-10:
        return ehyb, 0
    es = complex(eshift).real
    if ehyb.real > es + ebound:
        return es + ebound + 1j * ehyb.imag, 1
    if ehyb.real < es - ebound:
        return es - ebound + 1j * ehyb.imag, 1
    return ehyb, 0


def update_weight_hybrid(w, ovlp, ovlp_new, cfb, cmf, eshift, dt):
    """propagation/continuous.py:264-292.  ``w`` is a dict with keys weight, ot,
    ovlp, hybrid_energy.  On an infinite importance function the reference
    raises NameError (undefined ``ot_new``, :291); here the weight is set to 0."""
    ebound = (2.0 / dt) ** 0.5
    ovlp_ratio = ovlp_new / ovlp
    ehyb = -(cmath.log(ovlp_ratio) + cfb + cmf) / dt
    ehyb, trig = apply_bound_hybrid(ehyb, eshift, ebound)
    imp = cmath.exp(-dt * (0.5 * (ehyb + w['hybrid_energy']) - eshift))
    (magn, phase) = cmath.polar(imp)
    w['hybrid_energy'] = ehyb
    if not math.isinf(magn):
        dtheta = (-dt * ehyb - cfb).imag
        cosine_fac = max(0, math.cos(dtheta))
        w['weight'] *= magn * cosine_fac
        w['ot'] = ovlp_new
        w['ovlp'] = ovlp_new
        # back-propagation bookkeeping (continuous.py:284-289, walkers/stack.py:51-76)
        w['_wfac'] = (imp / magn, cosine_fac) if magn > 1e-16 else (0.0, 0.0)
    else:
        w['ot'] = ovlp_new
        w['weight'] = 0.0
        w['_wfac'] = None
    return trig


def apply_bound_local_energy(eloc, eshift, ebound):
    """propagation/continuous.py:216-230 (returns the complex eloc when inside the bounds,
    a real number when clamped -- as the reference does)."""
    if abs(eshift) < 1e-10:
        return eloc, 0
    es = complex(eshift).real
    if eloc.real > es + ebound:
        return es + ebound, 1
    if eloc.real < es - ebound:
        return es - ebound, 1
    return eloc, 0


def update_weight_local_energy(w, eloc, ovlp, ovlp_new, eshift, dt):
    """propagation/continuous.py:294-318.  ``eloc`` is the local energy of the walker
    BEFORE the step (walker.local_energy(system) uses the G of the Green's function
    evaluated at the start of propagate_walker_phaseless)."""
    ebound = (2.0 / dt) ** 0.5
    ovlp_ratio = ovlp_new / ovlp
    re_eloc, trig = apply_bound_local_energy(eloc, eshift, ebound)
    magn = numpy.exp(-0.5 * dt * complex(re_eloc + w['eloc'] - eshift).real)
    w_eloc_old = w['eloc']
    w['eloc'] = eloc
    if not math.isinf(magn):
        dtheta = cmath.phase(ovlp_ratio)
        cosine_fac = max(0, math.cos(dtheta))
        w['weight'] *= magn * cosine_fac
        w['ot'] = ovlp_new
        wfac_imag = numpy.exp(-0.5 * dt * complex(eloc + w_eloc_old - eshift).imag)   # continuous.py:299
        w['_wfac'] = (wfac_imag, cosine_fac) if magn > 1e-16 else (0.0, 0.0)
    else:
        w['ot'] = ovlp_new
        w['weight'] = 0.0
        w['_wfac'] = None
    return trig


# --------------------------------------------------------------------------
# Local energies
# --------------------------------------------------------------------------
def local_energy_generic_cholesky_opt(H1, ecore, G, Ghalf, rchol, na, nb):
    """estimators/generic.py:156-221 (half-rotated Cholesky energy)."""
    M = H1.shape[-1]
    e1b = numpy.sum(H1[0] * G[0]) + numpy.sum(H1[1] * G[1])
    naux = rchol.shape[1]
    Ga, Gb = Ghalf[0], Ghalf[1]
    Xa = rchol[:na * M].T.dot(Ga.ravel())
    Xb = rchol[na * M:(na + nb) * M].T.dot(Gb.ravel())
    ecoul = numpy.dot(Xa, Xa)
    ecoul += numpy.dot(Xb, Xb)
    ecoul += 2 * numpy.dot(Xa, Xb)
    rchol_a = rchol[:na * M].T
    rchol_b = rchol[na * M:(na + nb) * M].T
    Ta = numpy.zeros((naux, na, na), dtype=rchol.dtype)
    Tb = numpy.zeros((naux, nb, nb), dtype=rchol.dtype)
    GaT = Ga.T
    GbT = Gb.T
    for x in range(naux):
        Ta[x] = rchol_a[x].reshape((na, M)).dot(GaT)
        Tb[x] = rchol_b[x].reshape((nb, M)).dot(GbT)
    exxa = numpy.tensordot(Ta, Ta, axes=((0, 1, 2), (0, 2, 1)))
    exxb = numpy.tensordot(Tb, Tb, axes=((0, 1, 2), (0, 2, 1)))
    e2b = 0.5 * (ecoul - (exxa + exxb))
    return (e1b + e2b + ecore, e1b + ecore, e2b)


def local_energy_generic_cholesky(H1, ecore, G, chol):
    """estimators/generic.py:398-434 (full-G Cholesky energy; chol is [M*M, K])."""
    M = H1.shape[-1]
    e1b = numpy.sum(H1[0] * G[0]) + numpy.sum(H1[1] * G[1])
    nchol = chol.shape[-1]
    Ga, Gb = G[0], G[1]
    Xa = numpy.dot(chol.T, Ga.ravel())
    Xb = numpy.dot(chol.T, Gb.ravel())
    ecoul = numpy.dot(Xa, Xa)
    ecoul += numpy.dot(Xb, Xb)
    ecoul += 2 * numpy.dot(Xa, Xb)
    cv = chol.reshape((M, M, nchol))
    # T[l,k,n] = sum_i L[i,k,n] G[i,l];  exx = sum_{nlk} T[l,k,n] T[k,l,n]
    Ta = numpy.tensordot(Ga, cv, axes=((0), (0)))
    exxa = numpy.tensordot(Ta, Ta, axes=((0, 1, 2), (1, 0, 2)))
    Tb = numpy.tensordot(Gb, cv, axes=((0), (0)))
    exxb = numpy.tensordot(Tb, Tb, axes=((0, 1, 2), (1, 0, 2)))
    e2b = 0.5 * (ecoul - (exxa + exxb))
    return (e1b + e2b + ecore, e1b + ecore, e2b)


def local_energy_hubbard(T, U, G):
    """estimators/hubbard.py:93-114 (the 'symmetric' branch is overwritten there)."""
    ke = numpy.sum(T[0] * G[0] + T[1] * G[1])
    pe = U * numpy.dot(G[0].diagonal(), G[1].diagonal())
    return (ke + pe, ke, pe)


def local_energy_ueg(H1diag, vqvec, vol, ikpq_i, ikpq_kpq, ipmq_i, ipmq_pmq, G):
    """estimators/ueg.py:27-88 with the Cython gathers of
    estimators/ueg_kernels.pyx:42-75 restated as fancy-index sums.
    H1diag: [2, M] diagonal of H1; index lists: ragged, one int array per q."""
    nq = len(vqvec)
    ke = 0.0
    for s in (0, 1):
        ke = ke + numpy.dot(H1diag[s], numpy.diag(G[s]))
    Gkpq = numpy.zeros((2, nq), dtype=numpy.complex128)
    Gpmq = numpy.zeros((2, nq), dtype=numpy.complex128)
    Gprod = numpy.zeros((2, nq), dtype=numpy.complex128)
    for s in (0, 1):
        Gs = G[s]
        for iq in range(nq):
            ki, kk = ikpq_i[iq], ikpq_kpq[iq]
            pi, pp = ipmq_i[iq], ipmq_pmq[iq]
            Gkpq[s, iq] = Gs[ki, kk].sum()
            Gpmq[s, iq] = Gs[pi, pp].sum()
            if len(ki) and len(pi):
                # sum_{a,b} G[pi[b], kk[a]] * G[ki[a], pp[b]]
                A = Gs[numpy.ix_(pi, kk)]      # [b, a]
                B = Gs[numpy.ix_(ki, pp)]      # [a, b]
                Gprod[s, iq] = numpy.sum(A * B.T)
    fac = 1.0 / (2.0 * vol)
    essa = fac * vqvec.dot(Gkpq[0] * Gpmq[0] - Gprod[0])
    essb = fac * vqvec.dot(Gkpq[1] * Gpmq[1] - Gprod[1])
    eos = fac * vqvec.dot(Gkpq[0] * Gpmq[1]) + fac * vqvec.dot(Gkpq[1] * Gpmq[0])
    pe = essa + essb + eos
    return (ke + pe, ke, pe)


# --------------------------------------------------------------------------
# Re-orthogonalisation
# --------------------------------------------------------------------------
def reortho(phi, na, nb, detR_shift=0.0):
    """walkers/single_det.py:215-255.  In place; returns detR (product of |R_ii|
    over both spins).  Caller applies ``ot /= detR``."""
    (phi[:, :na], Rup) = scipy.linalg.qr(phi[:, :na], mode='economic')
    Rup_diag = numpy.diag(Rup)
    signs_up = numpy.sign(Rup_diag)
    phi[:, :na] = numpy.dot(phi[:, :na], numpy.diag(signs_up))
    log_det = numpy.sum(numpy.log(numpy.abs(Rup_diag)))
    if nb > 0:
        (phi[:, na:], Rdn) = scipy.linalg.qr(phi[:, na:], mode='economic')
        Rdn_diag = numpy.diag(Rdn)
        signs_dn = numpy.sign(Rdn_diag)
        phi[:, na:] = numpy.dot(phi[:, na:], numpy.diag(signs_dn))
        log_det += numpy.sum(numpy.log(numpy.abs(Rdn_diag)))
    return numpy.exp(log_det - detR_shift)


# --------------------------------------------------------------------------
# Population control (comb)
# --------------------------------------------------------------------------
def comb_parent_ix(weights, target, r):
    """walkers/handler.py:269-286: comb teeth against cumulative weights."""
    n = len(weights)
    parent_ix = numpy.zeros(n, dtype='i')
    total_weight = sum(weights)
    cprobs = numpy.cumsum(weights)
    comb = [(i + r) * (total_weight / target) for i in range(target)]
    iw = 0
    ic = 0
    while ic < len(comb):
        if comb[ic] < cprobs[iw]:
            parent_ix[iw] += 1
            ic += 1
        else:
            iw += 1
    return parent_ix


def comb_pairs(parent_ix):
    """walkers/handler.py:295-301: (clone, kill) pairs; ``zip`` truncates, so a
    parent with multiplicity >= 3 is copied once only."""
    kill = numpy.where(parent_ix == 0)[0]
    clone = numpy.where(parent_ix > 1)[0]
    return list(zip(clone.tolist(), kill.tolist()))


# --------------------------------------------------------------------------
# The step loop (single rank), restating qmc/afqmc.py:200-255 +
# estimators/mixed.py:133-289 + walkers/handler.py:225-338
# --------------------------------------------------------------------------
EST = dict(uweight=0, weight=1, enumer=2, edenom=3, eproj=4, e1b=5, e2b=6,
           ehyb=7, ovlp=8, time=9)   # estimators/mixed.py:460-469





# --------------------------------------------------------------------------
# Discrete Hirsch HS transformation for the Hubbard model
# --------------------------------------------------------------------------
class HirschModel(object):
    """Constants of propagation/hubbard.py:28-85 (Hirsch.__init__) for RHF/UHF-type trials."""

    def __init__(self, T, U, psi, na, nb, dt, charge_decomposition=False):
        self.kind = 'hubbard_hirsch'
        self.M, self.na, self.nb = T.shape[-1], na, nb
        self.psi = psi
        self.H1 = T
        self.U = U
        self.dt = dt
        self.bt2 = numpy.array([scipy.linalg.expm(-0.5 * dt * T[0]), scipy.linalg.expm(-0.5 * dt * T[1])])
        self.BH1 = self.bt2
        self.charge = charge_decomposition
        if charge_decomposition:
            self.gamma = numpy.arccosh(numpy.exp(-0.5 * dt * U + 0j))
            auxf = numpy.array([[numpy.exp(self.gamma), numpy.exp(self.gamma)],
                                [numpy.exp(-self.gamma), numpy.exp(-self.gamma)]])
            self.aux_wfac = numpy.exp(0.5 * dt * U) * numpy.array([numpy.exp(-self.gamma), numpy.exp(self.gamma)])
        else:
            self.gamma = numpy.arccosh(numpy.exp(0.5 * dt * U))
            auxf = numpy.array([[numpy.exp(self.gamma), numpy.exp(-self.gamma)],
                                [numpy.exp(-self.gamma), numpy.exp(self.gamma)]])
            self.aux_wfac = numpy.array([1.0, 1.0])
        self.auxf = auxf * numpy.exp(-0.5 * dt * U)
        self.delta = self.auxf - 1
        self.nfields = self.M

    def greens(self, phi):
        return greens_function(phi, self.psi, self.na, self.nb, getattr(self, 'log_shift', 0.0))

    def overlap(self, phi):
        return calc_overlap(phi, self.psi, self.na, self.nb, getattr(self, 'log_shift', 0.0))

    def local_energy(self, G, Ghalf):
        return local_energy_hubbard(self.H1, self.U, G)


def hirsch_inverse_overlap(model, w):
    """walkers/single_det.py:95-115."""
    na = model.na
    w['inv_ovlp'] = [scipy.linalg.inv((model.psi[:, :na].conj()).T.dot(w['phi'][:, :na])),
                     scipy.linalg.inv((model.psi[:, na:].conj()).T.dot(w['phi'][:, na:]))]


def hirsch_calc_otrial(w, log_shift=0.0):
    """walkers/single_det.py:141-168: 1 / (det inv_ovlp_a det inv_ovlp_b exp(-log_shift)) -- the shift enters the
    determinant of the INVERSE here (:159), i.e. with the opposite sign of calc_overlap (:192)."""
    sa, la = numpy.linalg.slogdet(w['inv_ovlp'][0])
    sb, lb = numpy.linalg.slogdet(w['inv_ovlp'][1])
    return 1.0 / (sa * sb * numpy.exp(la + lb - log_shift))


def hirsch_kinetic_importance_sampling(model, w):
    """propagation/hubbard.py:148-172."""
    kinetic_real(w['phi'], model.bt2, model.na)
    hirsch_inverse_overlap(model, w)
    ot_new = hirsch_calc_otrial(w, getattr(model, 'log_shift', 0.0))
    ratio = ot_new / w['ot']
    if abs(cmath.phase(ratio)) < 0.5 * math.pi:
        w['weight'] = w['weight'] * ratio.real
        w['ot'] = ot_new
    else:
        w['weight'] = 0.0


def sherman_morrison(Ainv, u, vt):
    """utils/linalg.py:6-30."""
    return Ainv - (Ainv.dot(numpy.outer(u, vt)).dot(Ainv)) / (1.0 + vt.dot(Ainv).dot(u))


def hirsch_two_body_single_site(model, w, uniform):
    """propagation/hubbard.py:174-225.  ``uniform()`` is numpy.random.random; returns the chosen fields."""
    na, M = model.na, model.M
    delta = model.delta
    fields = []
    for i in range(M):
        Gii = []
        for s, sl in ((0, slice(0, na)), (1, slice(na, None))):                   # :110-122
            q = numpy.dot(w['inv_ovlp'][s].T, w['phi'][i, sl])
            Gii.append(numpy.dot(model.psi.conj()[i, sl], q))
        probs = 0.5 * numpy.array([(1 + delta[0][0] * Gii[0]) * (1 + delta[0][1] * Gii[1]),
                                   (1 + delta[1][0] * Gii[0]) * (1 + delta[1][1] * Gii[1])])   # :549-551
        probs = probs * model.aux_wfac
        phaseless_ratio = numpy.maximum(probs.real, [0, 0])
        norm = sum(phaseless_ratio)
        r = uniform()
        if norm > 0:
            w['weight'] = w['weight'] * norm
            xi = 0 if r < phaseless_ratio[0] / norm else 1
            vtup = w['phi'][i, :na] * delta[xi, 0]
            vtdown = w['phi'][i, na:] * delta[xi, 1]
            w['phi'][i, :na] = w['phi'][i, :na] + vtup
            w['phi'][i, na:] = w['phi'][i, na:] + vtdown
            w['ot'] = 2 * w['ot'] * probs[xi]                                      # single_det.py:213
            fields.append(xi)
            if 'bp' in w:
                bp_push_field(w['bp'], xi)                                        # hubbard.py:215-216
            w['inv_ovlp'][0] = sherman_morrison(w['inv_ovlp'][0], model.psi[i, :na].conj(), vtup)
            w['inv_ovlp'][1] = sherman_morrison(w['inv_ovlp'][1], model.psi[i, na:].conj(), vtdown)
        else:
            w['weight'] = 0
            return fields
    return fields


def hirsch_two_body_direct(model, w, uniform):
    """propagation/hubbard.py:222-275 (two_body_direct, ``single_site_update: False``): every site's field drawn from the
    dynamic force bias of the CURRENT Green's function (Phys. Rev. A 92, 033603), all sites applied at once, one overlap."""
    na, M = model.na, model.M
    det, Ghalf, G = greens_function(w['phi'], model.psi, na, model.nb)                 # :238 walker.greens_function(trial)
    nia, nib = G[0].diagonal(), G[1].diagonal()
    fb_term = nia + nib - 1 if model.charge else nia - nib                            # :242-245
    fields = []
    fb_fac = 1.0
    for i in range(M):
        pp = 0.5 * numpy.exp(model.gamma * fb_term[i]).real
        pm = 0.5 * numpy.exp(-model.gamma * fb_term[i]).real
        norm = pp + pm
        r = uniform()
        if r < pp / norm:
            fields.append(0)
            fb_fac *= 0.5 * norm * numpy.exp(-model.gamma * fb_term[i]).real
        else:
            fields.append(1)
            fb_fac *= 0.5 * norm * numpy.exp(model.gamma * fb_term[i]).real
    BVa = numpy.array([model.auxf[xi, 0] for xi in fields])
    BVb = numpy.array([model.auxf[xi, 1] for xi in fields])
    w['phi'][:, :na] = BVa[:, None] * w['phi'][:, :na]                               # :261-264
    w['phi'][:, na:] = BVb[:, None] * w['phi'][:, na:]
    ovlp = calc_overlap(w['phi'], model.psi, na, model.nb, getattr(model, 'log_shift', 0.0))
    wfac = 1.0 + 0j
    for xi in fields:
        wfac *= model.aux_wfac[xi]
    ratio = wfac * ovlp / w['ot']
    if abs(cmath.phase(ratio)) < 0.5 * math.pi:
        w['ot'] = ovlp
        w['weight'] *= (fb_fac * ratio).real
    else:
        w['weight'] = 0
    return fields


def propagate_walker_hirsch(model, w, uniform, eshift):
    """propagation/hubbard.py:285-312 (propagate_walker_constrained)."""
    fields = None
    if abs(w['weight']) > 0:
        hirsch_kinetic_importance_sampling(model, w)
    if abs(w['weight']) > 0 and not getattr(model, 'single_site', True):
        fields = hirsch_two_body_direct(model, w, uniform)
    elif abs(w['weight']) > 0:
        fields = hirsch_two_body_single_site(model, w, uniform)
    if abs(numpy.real(w['weight'])) > 0:
        hirsch_kinetic_importance_sampling(model, w)
    w['weight'] *= numpy.exp(model.dt * eshift)
    w['ovlp'] = w['ot']
    return fields


def propagate_walker_hirsch_free(model, w, uniform, eshift=0):
    """propagation/hubbard.py:303-343 (propagate_walker_free): no importance sampling; every site takes field 0 or 1
    with probability 1/2, the weight picks up |prod aux_wfac| and exp(dt eshift), the phase arg(prod aux_wfac)."""
    na, M = model.na, model.M
    kinetic_real(w['phi'], model.bt2, na)
    wfac = 1.0
    fields = []
    for i in range(M):
        if abs(w['weight']) > 0:
            r = uniform()
            xi = 0 if r < 0.5 else 1
            vtup = w['phi'][i, :na] * model.delta[xi, 0]
            vtdown = w['phi'][i, na:] * model.delta[xi, 1]
            w['phi'][i, :na] = w['phi'][i, :na] + vtup
            w['phi'][i, na:] = w['phi'][i, na:] + vtdown
            wfac *= model.aux_wfac[xi]
            fields.append(xi)
    kinetic_real(w['phi'], model.bt2, na)
    hirsch_inverse_overlap(model, w)
    ovlp = hirsch_calc_otrial(w, getattr(model, 'log_shift', 0.0))
    magn, dtheta = cmath.polar(wfac)
    w['weight'] *= numpy.exp(model.dt * eshift) * magn
    w['phase'] = w.get('phase', 1.0 + 0j) * numpy.exp(1j * dtheta)
    w['ot'] = ovlp
    w['ovlp'] = ovlp
    return fields


# --------------------------------------------------------------------------
# Back-propagation
# --------------------------------------------------------------------------
def bp_new(nfields, nbp):
    """walkers/stack.py:19-32 (FieldConfig with nprop_tot == nbp)."""
    return dict(configs=numpy.zeros((nbp, nfields), dtype=numpy.complex128),
                cos_fac=numpy.zeros(nbp), weight_fac=numpy.zeros(nbp, dtype=numpy.complex128),
                step=0, nbp=nbp, ib=0)


def bp_push_field(fc, xi):
    """walkers/stack.py:35-49 (FieldConfig.push, the discrete propagator's one-field-at-a-time variant): neither the
    weight nor the cosine factors are recorded on this path."""
    fc['configs'][fc['step'], fc['ib']] = xi
    fc['ib'] = (fc['ib'] + 1) % fc['configs'].shape[1]
    if fc['ib'] == 0:
        fc['step'] += 1


def bp_push(fc, config, wfac):
    """walkers/stack.py:51-76 (FieldConfig.update)."""
    fc['configs'][fc['step']] = config
    fc['weight_fac'][fc['step']] = wfac[0]
    fc['cos_fac'][fc['step']] = numpy.real(wfac[1])
    fc['step'] += 1


def bp_copy(fc):
    return dict(configs=fc['configs'].copy(), cos_fac=fc['cos_fac'].copy(), weight_fac=fc['weight_fac'].copy(),
                step=fc['step'], nbp=fc['nbp'], ib=fc.get('ib', 0))


def exponentiate_matrix(Mx, order=6):
    """utils/linalg.py:163-170."""
    T = numpy.copy(Mx)
    EXPM = numpy.identity(Mx.shape[0], dtype=Mx.dtype)
    for n in range(1, order + 1):
        EXPM += T
        T = Mx.dot(T) / (n + 1)
    return EXPM


def reortho_qr(A):
    """utils/linalg.py:82-105."""
    (Q, R) = scipy.linalg.qr(A, mode='economic')
    signs = numpy.diag(numpy.sign(numpy.diag(R)))
    return Q.dot(signs), scipy.linalg.det(signs.dot(R))


def back_propagate_generic(phi, configs, hs_pot, M, na, nstblz, BT2, dt, vhs=None):
    """propagation/generic.py:253-290 with :181-211 (and, with ``vhs`` = the plane-wave HS potential,
    propagation/planewave.py:114-178): apply B(x_n)^H ... B(x_1)^H to the trial, most recent field
    configuration first, re-orthogonalising every nstblz steps (not at i == 0)."""
    for (i, c) in enumerate(configs[::-1]):
        VHS = vhs(c) if vhs is not None else 1j * dt ** 0.5 * hs_pot.dot(c).reshape(M, M)
        EXP_VHS = exponentiate_matrix(VHS)
        Bup = BT2[0].dot(EXP_VHS).dot(BT2[0])
        Bdn = BT2[1].dot(EXP_VHS).dot(BT2[1])
        phi[:, :na] = numpy.dot(Bup.conj().T, phi[:, :na])
        phi[:, na:] = numpy.dot(Bdn.conj().T, phi[:, na:])
        if i != 0 and i % nstblz == 0:
            phi[:, :na], _ = reortho_qr(phi[:, :na])
            phi[:, na:], _ = reortho_qr(phi[:, na:])
    return phi


def back_propagate_hirsch(phi, configs, U, na, nstblz, BT2, dt):
    """propagation/hubbard.py:568-600,634-672: B(x)^H = (BT2 diag(auxf[x, spin]) BT2)^H per recorded configuration, most
    recent first, with the SPIN decomposition's auxf whatever the propagator used (:589-591)."""
    gamma = numpy.arccosh(numpy.exp(0.5 * dt * U))
    auxf = numpy.array([[numpy.exp(gamma), numpy.exp(-gamma)], [numpy.exp(-gamma), numpy.exp(gamma)]])
    for (i, c) in enumerate(configs[::-1]):
        bv_up = numpy.array([auxf[int(xi.real), 0] for xi in c])
        bv_down = numpy.array([auxf[int(xi.real), 1] for xi in c])
        Bup = BT2[0].dot(numpy.einsum('i,ij->ij', bv_up, BT2[0]))
        Bdown = BT2[1].dot(numpy.einsum('i,ij->ij', bv_down, BT2[1]))
        phi[:, :na] = Bup.conj().T.dot(phi[:, :na])
        phi[:, na:] = Bdown.conj().T.dot(phi[:, na:])
        if i != 0 and i % nstblz == 0:
            phi[:, :na], _ = reortho_qr(phi[:, :na])
            phi[:, na:], _ = reortho_qr(phi[:, na:])
    return phi


def bp_update(model, walkers, nstblz, est, restore_weights=None, init=None, eval_energy=False, reset=True):
    """estimators/back_propagation.py:127-226 (update_uhf, one_rdm only).  ``est`` is
    [3 energies, denominator, G.flatten()]; called when the field buffers hold one of the split lengths; at the last
    split (``reset``) it resets them and copies phi -> phi_old (walkers/handler.py:200-203)."""
    M, na = model.M, model.na
    vhs = None
    if model.kind == 'ueg':
        vhs = lambda c: vhs_ueg(model.iA, model.iB, c, M, model.sqrt_dt)       # noqa: E731  planewave.py:94-112
    for w in walkers:
        fc = w['bp']
        phi_bp = numpy.array(model.psi if init is None else init, dtype=numpy.complex128, copy=True)
        if model.kind == 'hubbard_hirsch':
            back_propagate_hirsch(phi_bp, fc['configs'][:fc['step']], model.U, na, nstblz, model.bt2, model.dt)
        else:
            back_propagate_generic(phi_bp, fc['configs'][:fc['step']], getattr(model, 'hs_pot', None), M, na, nstblz,
                                   model.BH1, model.dt, vhs=vhs)
        G = numpy.array([gab(phi_bp[:, :na], w['phi_old'][:, :na]).T,
                         gab(phi_bp[:, na:], w['phi_old'][:, na:]).T])
        if restore_weights is not None:
            cosine_fac = numpy.prod(fc['cos_fac'][:fc['step']])
            ph_fac = numpy.prod(fc['weight_fac'][:fc['step']])
            wfac = ph_fac / cosine_fac if restore_weights == "full" else ph_fac
            weight = w['weight'] * wfac
        else:
            weight = w['weight']
        if eval_energy:      # local_energy(system, G, opt=False) -> full-G Cholesky energy (:159-163)
            est[:3] += weight * numpy.array(local_energy_generic_cholesky(model.H1, model.ecore, G, model.hs_pot))
        est[3] += weight
        est[4:] += weight * G.flatten()
        if reset:
            fc['step'] = 0                                  # FieldConfig.reset (stack.py:124-127)
    if reset:
        for w in walkers:
            w['phi_old'] = w['phi'].copy()


# --------------------------------------------------------------------------
# Multi-determinant trial |psi_T> = sum_d c_d |D_d>  
# --------------------------------------------------------------------------
def msd_greens_function(phi, psi, coeffs, na, nb):
    """walkers/multi_det.py:194-229.  psi[ndet, M, na+nb].  Returns (tot_ovlp,
    weights[ndet] = conj(c_d) <D_d|phi>, Gi[ndet, 2, M, M]).  Determinants whose
    overlap is below 1e-16 are skipped by the reference (their Gi keeps the
    previous content); here they get zero weight and a zero Gi."""
    ndet, M = psi.shape[0], psi.shape[1]
    Gi = numpy.zeros((ndet, 2, M, M), dtype=numpy.complex128)
    weights = numpy.zeros(ndet, dtype=numpy.complex128)
    tot = 0.0
    for ix in range(ndet):
        det = psi[ix]
        Oup = numpy.dot(phi[:, :na].T, det[:, :na].conj())
        ovlp = scipy.linalg.det(Oup)
        if abs(ovlp) < 1e-16:
            continue
        Gi[ix, 0] = numpy.dot(det[:, :na].conj(), numpy.dot(scipy.linalg.inv(Oup), phi[:, :na].T))
        Odn = numpy.dot(phi[:, na:].T, det[:, na:].conj())
        ovlp *= scipy.linalg.det(Odn)
        if abs(ovlp) < 1e-16:
            continue
        Gi[ix, 1] = numpy.dot(det[:, na:].conj(), numpy.dot(scipy.linalg.inv(Odn), phi[:, na:].T))
        weights[ix] = coeffs[ix].conj() * ovlp
        tot += weights[ix]
    return tot, weights, Gi


def msd_calc_overlap(phi, psi, coeffs, na, nb, weights_out=None):
    """walkers/multi_det.py:135-162.  The reference also refreshes walker.weights here (:160);
    pass ``weights_out`` to receive them."""
    tot = 0.0
    for ix in range(psi.shape[0]):
        Oup = numpy.dot(psi[ix, :, :na].conj().T, phi[:, :na])
        Odn = numpy.dot(psi[ix, :, na:].conj().T, phi[:, na:])
        wd = coeffs[ix].conj() * scipy.linalg.det(Oup) * scipy.linalg.det(Odn)
        if weights_out is not None:
            weights_out[ix] = wd
        tot += wd
    return tot


def force_bias_msd(Gi, weights, hs_pot, sqrt_dt, mf_shift):
    """propagation/generic.py:154-157 with walkers/multi_det.py:283-290: for every field
    vbias_n = sum_d w_d (G_d^a + G_d^b) . V_n / sum_d w_d (one vectorised dot here)."""
    Gw = numpy.einsum('d,dpq->pq', weights, Gi[:, 0] + Gi[:, 1]) / numpy.sum(weights)
    vbias = numpy.dot(hs_pot.T, Gw.ravel())
    return -sqrt_dt * (1j * vbias - mf_shift)


def local_energy_msd(H1, ecore, Gi, weights, chol):
    """estimators/mixed.py:439-448 with the full-G Cholesky energy (estimators/generic.py:398-434)."""
    num = numpy.zeros(3, dtype=numpy.complex128)
    for w, G in zip(weights, Gi):
        num += w * numpy.array(local_energy_generic_cholesky(H1, ecore, G, chol))
    return tuple(num / numpy.sum(weights))


class RefModel(object):
    """Plain-array bundle of everything the hot path reads.

    kind: 'generic' | 'generic_msd' | 'hubbard' | 'hubbard_spin' | 'ueg'.  For
    'generic_msd' psi is [ndet, M, na+nb] and ``coeffs`` [ndet] is required; the
    second and third items the Green's function returns are then (weights, Gi).
    """

    def __init__(self, kind, M, na, nb, psi, BH1, mf_shift, dt, **kw):
        self.kind = kind
        self.M, self.na, self.nb = M, na, nb
        self.psi = psi
        self.BH1 = BH1
        self.mf_shift = numpy.asarray(mf_shift, dtype=numpy.complex128)
        self.dt = dt
        self.sqrt_dt = dt ** 0.5
        self.exp_order = kw.get('exp_order', 6)
        self.__dict__.update(kw)
        self.nfields = len(self.mf_shift)

    # -- trial dispatch -------------------------------------------------
    def greens(self, phi):
        if self.kind == 'generic_msd':
            return msd_greens_function(phi, self.psi, self.coeffs, self.na, self.nb)
        return greens_function(phi, self.psi, self.na, self.nb, getattr(self, 'log_shift', 0.0))

    def overlap(self, phi):
        if self.kind == 'generic_msd':
            return msd_calc_overlap(phi, self.psi, self.coeffs, self.na, self.nb)
        return calc_overlap(phi, self.psi, self.na, self.nb, getattr(self, 'log_shift', 0.0))

    # -- system dispatch ------------------------------------------------
    def force_bias(self, Ghalf, G):
        if self.kind == 'generic_msd':
            return force_bias_msd(G, Ghalf, self.hs_pot, self.sqrt_dt, self.mf_shift)
        if self.kind == 'generic':
            return force_bias_generic(Ghalf, self.rchol, self.na, self.nb, self.M,
                                      self.sqrt_dt, self.mf_shift)
        if self.kind == 'hubbard':
            return force_bias_hubbard(G, self.U, self.sqrt_dt, self.mf_shift)
        if self.kind == 'hubbard_spin':
            return force_bias_hubbard_spin(G, self.U, self.sqrt_dt, self.mf_shift)
        if self.kind == 'ueg':
            return force_bias_ueg(G, self.iA, self.iB, self.sqrt_dt)
        raise ValueError(self.kind)

    def vhs(self, xs):
        if self.kind in ('generic', 'generic_msd'):
            return vhs_generic(self.hs_pot, xs, self.M, self.sqrt_dt)
        if self.kind == 'hubbard':
            return vhs_hubbard(xs, self.U, self.sqrt_dt)
        if self.kind == 'hubbard_spin':
            return vhs_hubbard_spin(xs, self.U, self.dt)
        if self.kind == 'ueg':
            return vhs_ueg(self.iA, self.iB, xs, self.M, self.sqrt_dt)
        raise ValueError(self.kind)

    def local_energy(self, G, Ghalf):
        if self.kind == 'generic_msd':
            return local_energy_msd(self.H1, self.ecore, G, Ghalf, self.hs_pot)
        if self.kind == 'generic':
            return local_energy_generic_cholesky_opt(self.H1, self.ecore, G, Ghalf,
                                                     self.rchol, self.na, self.nb)
        if self.kind in ('hubbard', 'hubbard_spin'):
            return local_energy_hubbard(self.H1, self.U, G)
        if self.kind == 'ueg':
            e = local_energy_ueg(self.H1diag, self.vqvec, self.vol, self.ikpq_i,
                                 self.ikpq_kpq, self.ipmq_i, self.ipmq_pmq, G)
            return e
        raise ValueError(self.kind)


def new_walker(model, phi0, weight=1.0):
    """walkers/walker.py:24-61 + single_det.py:64-67 (the state the loop touches)."""
    phi = numpy.array(phi0, dtype=numpy.complex128, copy=True)
    ot = model.overlap(phi)
    w = dict(phi=phi, weight=weight, unscaled_weight=weight, ot=ot, ovlp=ot,
             hybrid_energy=0.0, total_weight=0.0, detR=1.0, phase=1.0 + 0j, eloc=0.0, log_detR=0.0)
    if getattr(model, 'track_G', False):
        w['G'] = model.greens(phi)[2]              # walker.G as left by the constructor (single_det.py:81)
    return w


def propagate_walker_free(model, w, xi, eshift):
    """propagation/continuous.py:175-200 (no force bias)."""
    na, nb = model.na, model.nb
    kinetic_real(w['phi'], model.BH1, na)
    xbar = numpy.zeros(model.nfields)
    xs, cmf, cfb, ntrig = shift_fields(xi, xbar, model.mf_shift, model.sqrt_dt)
    VHS = model.vhs(xs)
    if VHS.ndim == 3:
        apply_exponential(w['phi'][:, :na], VHS[0], model.exp_order)
        apply_exponential(w['phi'][:, na:], VHS[1], model.exp_order)
    else:
        apply_exponential(w['phi'][:, :na], VHS, model.exp_order)
        apply_exponential(w['phi'][:, na:], VHS, model.exp_order)
    kinetic_real(w['phi'], model.BH1, na)
    ovlp_new = model.overlap(w['phi'])
    (magn, dtheta) = cmath.polar(cmath.exp(cmf + model.dt * eshift))
    w['weight'] *= magn
    w['phase'] *= cmath.exp(1j * dtheta)
    w['ot'] = ovlp_new
    w['ovlp'] = ovlp_new
    return 0, 0


def propagate_walker_phaseless(model, w, xi, eshift, hybrid=True):
    """propagation/continuous.py:232-262 with the two-body part of :113-173.
    ``xi`` is the real normal field vector the reference draws at :133.
    Returns (nfb_trig, nhe_trig)."""
    na, nb = model.na, model.nb
    ovlp, Ghalf, G = model.greens(w['phi'])
    if 'G' in w:
        w['G'] = G                                 # walker.G: the Green's function BEFORE this step (continuous.py:245)
    kinetic_real(w['phi'], model.BH1, na)
    xbar = model.force_bias(Ghalf, G)
    xs, cmf, cfb, ntrig = shift_fields(xi, xbar, model.mf_shift, model.sqrt_dt)
    VHS = model.vhs(xs)
    if VHS.ndim == 3:
        apply_exponential(w['phi'][:, :na], VHS[0], model.exp_order)
        if nb > 0:
            apply_exponential(w['phi'][:, na:], VHS[1], model.exp_order)
    else:
        apply_exponential(w['phi'][:, :na], VHS, model.exp_order)
        if nb > 0:
            apply_exponential(w['phi'][:, na:], VHS, model.exp_order)
    kinetic_real(w['phi'], model.BH1, na)
    ovlp_new = model.overlap(w['phi'])
    if hybrid:
        htrig = update_weight_hybrid(w, ovlp, ovlp_new, cfb, cmf, eshift, model.dt)
        if 'bp' in w and w['_wfac'] is not None:
            bp_push(w['bp'], xs, w['_wfac'])
    else:
        if model.kind == 'generic_msd':
            # Reference behaviour (continuous.py:296 after :261): the energy is evaluated AFTER
            # calc_overlap refreshed walker.weights for the propagated walker (multi_det.py:160), but
            # with the Green's functions Gi of the un-propagated walker.
            Ghalf = numpy.zeros(len(model.coeffs), dtype=numpy.complex128)
            msd_calc_overlap(w['phi'], model.psi, model.coeffs, na, nb, weights_out=Ghalf)
        eloc = complex(model.local_energy(G, Ghalf)[0])
        htrig = update_weight_local_energy(w, eloc, ovlp, ovlp_new, eshift, model.dt)
        if 'bp' in w and w['_wfac'] is not None:
            bp_push(w['bp'], xs, w['_wfac'])
    return ntrig, htrig


def update_log_ovlp(model, walkers):
    """walkers/handler.py:456-475 (use_log_shift: True) for one rank: running averages of log <|ot|>, log <|detR|>
    and <|log_detR|>, kept on the model since every walker carries the same values (walker.py:49-52)."""
    n = getattr(model, 'shift_counter', 1)
    nw = len(walkers)
    log_shift = numpy.log(sum(abs(w['ot']) for w in walkers) / nw)
    detR_shift = numpy.log(sum(abs(w['detR']) for w in walkers) / nw)
    log_detR_shift = sum(abs(w['log_detR']) for w in walkers) / nw
    model.log_shift = (getattr(model, 'log_shift', 0.0) * (n - 1) + log_shift) / n
    model.log_detR_shift = (getattr(model, 'log_detR_shift', 0.0) * (n - 1) + log_detR_shift) / n
    model.detR_shift = (getattr(model, 'detR_shift', 0.0) * (n - 1) + detR_shift) / n
    model.shift_counter = n + 1


def pop_control(model, walkers, target, r, use_log_shift=False):
    """walkers/handler.py:225-338 for one rank.  ``r`` is the uniform the
    reference draws at :276.  Returns parent_ix."""
    if len(walkers) == 1:
        return None
    if use_log_shift:
        update_log_ovlp(model, walkers)
    weights = numpy.array([abs(w['weight']) for w in walkers])
    total_weight = sum(weights)
    scale = total_weight / target
    if total_weight < 1e-8:
        raise RuntimeError("total weight %g" % total_weight)
    for w in walkers:
        w['total_weight'] = total_weight
        w['unscaled_weight'] = w['weight']
        w['weight'] = w['weight'] / scale
    parent_ix = comb_parent_ix(weights / scale, target, r)
    for (c, k) in comb_pairs(parent_ix):
        src = walkers[c]
        dst = walkers[k]
        for key, val in src.items():
            if key == 'bp':
                dst[key] = bp_copy(val)
            else:
                dst[key] = numpy.array(val, copy=True) if isinstance(val, numpy.ndarray) else val
    for w in walkers:
        w['weight'] = 1.0
    return parent_ix


def mixed_update(model, est, walkers, step, energy_eval_freq, free_projection=False, rdm=None):
    """estimators/mixed.py:180-225 (importance-sampling branch, le_oratio == 1) and
    :151-175 (free projection: wfac = weight * ot * phase).  ``rdm`` [2, M, M] (one_rdm: True, :226-229):
    += weight * walker.G.real with whatever walker.G currently holds."""
    if free_projection:
        for w in walkers:
            wfac = w['weight'] * w['ot'] * w['phase']
            if step % energy_eval_freq == 0:
                _, Ghalf, G = model.greens(w['phi'])
                E, T, V = model.local_energy(G, Ghalf)
                est[EST['enumer']] += wfac * E
                est[EST['e1b']] += wfac * T
                est[EST['e2b']] += wfac * V
                est[EST['edenom']] += wfac
            est[EST['uweight']] += w['unscaled_weight']
            est[EST['weight']] += wfac
            est[EST['ehyb']] += wfac * w['hybrid_energy']
            est[EST['ovlp']] += w['weight'] * abs(w['ot'])
        return
    for w in walkers:
        if step % energy_eval_freq == 0:
            _, Ghalf, G = model.greens(w['phi'])
            if 'G' in w:
                w['G'] = G
            E, T, V = model.local_energy(G, Ghalf)
            est[EST['enumer']] += w['weight'] * complex(E).real
            est[EST['e1b']] += w['weight'] * complex(T).real
            est[EST['e2b']] += w['weight'] * complex(V).real
            est[EST['edenom']] += w['weight']
        est[EST['uweight']] += w['unscaled_weight']
        est[EST['weight']] += w['weight']
        est[EST['ovlp']] += w['weight'] * abs(w['ot'])
        est[EST['ehyb']] += w['weight'] * w['hybrid_energy']
        if rdm is not None:
            rdm += w['weight'] * w['G'].real


def block_reduce(est, nsteps):
    """estimators/mixed.py:256-274 for one rank.  Returns (global_estimates, eshift)."""
    es = est.copy()
    es[EST['uweight']:EST['weight'] + 1] /= nsteps
    es[EST['ehyb']:EST['time'] + 1] /= nsteps
    gs = es.copy()
    gs[EST['eproj']] = gs[EST['enumer']]
    gs[EST['eproj']:EST['e2b'] + 1] = gs[EST['eproj']:EST['e2b'] + 1] / gs[EST['edenom']]
    gs[EST['ehyb']] /= gs[EST['weight']]
    gs[EST['ovlp']] /= gs[EST['weight']]
    return gs, numpy.array([gs[EST['ehyb']], gs[EST['eproj']]])


def run_afqmc(model, walkers, xi_source, r_source, nsteps, nblocks, nstblz=10,
              npop_control=1, energy_eval_freq=None, eqlb_time=2.0, hybrid=True,
              record=None, verbose=False, free_projection=False, nbp=None, bp_out=None,
              restore_weights=None, bp_energy=False, uniform_source=None, bp_nsplit=1, rdm_out=None,
              use_log_shift=False):
    """qmc/afqmc.py:200-255 for one rank.

    xi_source(step, iw) -> real [nfields] normal field for walker iw (called
    only for live walkers, in walker order, exactly as numpy.random.normal at
    propagation/continuous.py:133); r_source(step) -> comb uniform.
    ``record`` (optional list) receives per-step dict(weight, ot, hybrid_energy,
    parent_ix).  Returns list of per-block global estimates.

    Discrete Hirsch propagation (model.kind == 'hubbard_hirsch'): ``uniform_source(step)`` returns a
    callable standing in for numpy.random.random during that step (site updates of every live walker
    in walker order, then the comb); xi_source / r_source are unused.

    ``verbose`` mirrors the driver flag: the step-0 estimates are reduced and
    zeroed only when it is set (qmc/afqmc.py:220-221); otherwise they stay in
    the accumulator and are folded into the first block, as in the reference.
    """
    if energy_eval_freq is None:
        energy_eval_freq = nsteps
    if nbp is not None:
        # estimators/handler.py:87-92, walkers/walker.py:43,55-58: field history + phi_old per walker
        for w in walkers:
            w['bp'] = bp_new(model.nfields, nbp)
            w['phi_old'] = w['phi'].copy()
    ntot = len(walkers)
    for w in walkers:
        w['total_weight'] = ntot       # walkers/handler.py:164
    neqlb = int(eqlb_time / model.dt)
    est = numpy.zeros(10, dtype=numpy.complex128)
    blocks = []
    eshift_pair = numpy.array([0, 0], dtype=numpy.complex128)
    eshift = 0
    rdm = None
    if rdm_out is not None:                       # mixed estimator with one_rdm: True (walkers carry w['G'])
        rdm = numpy.zeros((2, model.M, model.M))
    # step-0 estimator pass (qmc/afqmc.py:214-221)
    mixed_update(model, est, walkers, 0, energy_eval_freq, free_projection, rdm)
    if verbose:
        gs, eshift_pair = block_reduce(est, 1)
        blocks.append(gs)
        est[:] = 0
    for step in range(1, nsteps * nblocks + 1):
        if step % nstblz == 0:
            for w in walkers:
                detR = reortho(w['phi'], model.na, model.nb, getattr(model, 'detR_shift', 0.0))
                w['detR'] = detR
                w['log_detR'] += numpy.log(detR)         # single_det.py:251
                w['ot'] = w['ot'] / detR
                w['ovlp'] = w['ot']
                if free_projection:                    # walkers/handler.py:178-181
                    (magn, dtheta) = cmath.polar(detR)
                    w['weight'] *= magn
                    w['phase'] *= cmath.exp(1j * dtheta)
        uni = uniform_source(step) if uniform_source is not None else None
        for iw, w in enumerate(walkers):
            if abs(w['weight']) > 1e-8:
                if uni is not None and free_projection:
                    propagate_walker_hirsch_free(model, w, uni, eshift)
                elif uni is not None:
                    propagate_walker_hirsch(model, w, uni, eshift)
                elif free_projection:
                    propagate_walker_free(model, w, xi_source(step, iw), eshift)
                else:
                    propagate_walker_phaseless(model, w, xi_source(step, iw), eshift, hybrid)
            if (abs(w['weight']) > w['total_weight'] * 0.10) and step > 1:
                w['weight'] = w['total_weight'] * 0.10
        parent_ix = None
        if step % npop_control == 0:
            parent_ix = pop_control(model, walkers, ntot, uni() if uni is not None else r_source(step),
                                    use_log_shift)
        mixed_update(model, est, walkers, step, energy_eval_freq, free_projection, rdm)
        if nbp is not None:                                          # back_propagation.py:68-69,145-147
            splits = [(i + 1) * (nbp // bp_nsplit) for i in range(bp_nsplit)]
            cur = walkers[0]['bp']['step']
            if cur in splits:
                bpe = numpy.zeros(4 + 2 * model.M * model.M, dtype=numpy.complex128)
                bp_update(model, walkers, nstblz, bpe, restore_weights, eval_energy=bp_energy, reset=cur == splits[-1])
                bp_out.append(bpe if bp_nsplit == 1 else (cur, bpe))   # print_step: one Reduce per window
        if record is not None:
            record.append(dict(
                weight=numpy.array([w['weight'] for w in walkers]),
                unscaled_weight=numpy.array([w['unscaled_weight'] for w in walkers]),
                ot=numpy.array([w['ot'] for w in walkers]),
                hybrid_energy=numpy.array([w['hybrid_energy'] for w in walkers]),
                phase=numpy.array([w['phase'] for w in walkers]),
                eloc=numpy.array([w['eloc'] for w in walkers]),
                parent_ix=None if parent_ix is None else parent_ix.copy()))
        if step % nsteps == 0:
            gs, eshift_pair = block_reduce(est, nsteps)
            blocks.append(gs)
            est[:] = 0
            if rdm is not None:                    # mixed.py:279-283
                rdm_out.append(rdm / nsteps / gs[EST['weight']])
                rdm[:] = 0
        if step < neqlb:
            eshift = eshift_pair[0].real if hybrid else eshift_pair[1].real
        else:
            eshift += (eshift_pair[0].real - eshift)
    return blocks
