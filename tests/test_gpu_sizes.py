"""Mid-size parity (odd dimensions, nw not a multiple of 16, na != nb): exercises
the work-group-tiled LDS-ring GEMMs, tile-edge predication and the zero-padded
leading dimensions.  HIP library vs the CPU oracle on every walker."""
import numpy
import pytest

from oracle import afqmc_ref as ref
from pauxy_amd import _lib as L
from pauxy_amd import systems, trial as trial_mod
from pauxy_amd.propagation import setup
from tests.helpers import make_device

pytestmark = pytest.mark.gpu
TOL = 1e-10


def close(a, b, tol=TOL):
    a, b = numpy.asarray(a), numpy.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(numpy.max(numpy.abs(b))))
    err = float(numpy.max(numpy.abs(a - b))) / scale
    assert err <= tol, err


def build(M, K, na, nb, complex_trial, seed=3, dt=0.01):
    rng = numpy.random.RandomState(seed)
    s = systems.synthetic_generic(M, K, (na, nb), seed=seed)
    e, v = numpy.linalg.eigh(s.H1[0])
    psi = numpy.zeros((M, na + nb), dtype=complex)
    psi[:, :na] = v[:, :na]
    psi[:, na:] = v[:, :nb]
    if complex_trial:
        psi = psi + 0.05 * (rng.rand(M, na + nb) + 1j * rng.rand(M, na + nb))
    t = trial_mod.SingleDetTrial(s, psi)
    BH1, mf = setup.generic_propagator_arrays(s, t, dt)
    model = ref.RefModel('generic', M, na, nb, t.psi, BH1, mf, dt, hs_pot=s.hs_pot, rchol=t._rchol,
                         H1=s.H1.astype(complex), ecore=0.37)
    return model, rng


@pytest.mark.parametrize("M,K,na,nb,nw,cplx", [(37, 45, 7, 6, 70, False), (24, 50, 5, 5, 130, True)])
def test_midsize_generic(M, K, na, nb, nw, cplx):
    model, rng = build(M, K, na, nb, cplx)
    dev = make_device(model, nw)
    phis = numpy.array([model.psi + 0.1 * (rng.rand(M, na + nb) + 1j * rng.rand(M, na + nb)) for _ in range(nw)])
    dev.set(L.F_PHI, phis)
    det = dev.greens(want_G=True)
    refs = [ref.greens_function(p, model.psi, na, nb) for p in phis]
    close(det, numpy.array([r[0] for r in refs]))
    close(dev.get(L.F_GHALF), numpy.array([numpy.concatenate([r[1][0], r[1][1]]) for r in refs]))
    close(dev.get(L.F_G), numpy.array([r[2] for r in refs]))
    xbar = dev.force_bias()
    close(xbar, numpy.array([model.force_bias(r[1], r[2]) for r in refs]))
    xi = rng.normal(size=(nw, K))
    xs = numpy.array([ref.shift_fields(xi[i], xbar[i], model.mf_shift, model.sqrt_dt)[0] for i in range(nw)])
    vhs = dev.vhs(xs)
    close(vhs[:, 0], numpy.array([model.vhs(x) for x in xs]))
    E = dev.local_energy()
    close(E, numpy.array([model.local_energy(r[2], r[1]) for r in refs]))
    # full step: half the walkers dead
    w0 = numpy.ones(nw)
    w0[::3] = 0.0
    dev.set(L.F_WEIGHT, w0)
    dev.set(L.F_OT, numpy.array([ref.calc_overlap(p, model.psi, na, nb) for p in phis]))
    dev.propagate(xi, -1.5)
    out_phi, out_w, out_e = dev.get(L.F_PHI), dev.get(L.F_WEIGHT), dev.get(L.F_HYBRID_ENERGY)
    for i in range(nw):
        if w0[i] == 0.0:
            assert numpy.array_equal(out_phi[i], phis[i]) and out_w[i] == 0.0
            continue
        w = ref.new_walker(model, phis[i])
        ref.propagate_walker_phaseless(model, w, xi[i], -1.5)
        close(out_phi[i], w['phi'])
        close(out_w[i], w['weight'])
        close(out_e[i], w['hybrid_energy'])
    detR = dev.reortho()
    q = dev.get(L.F_PHI)
    for i in range(0, nw, 7):
        p = out_phi[i].copy()
        d = ref.reortho(p, na, nb)
        close(q[i], p)
        close(detR[i], d)
    dev.close()


def test_vhs_nonsymmetric_cholesky():
    """The HS-potential GEMM contracts only the columns p <= q when every L_n is symmetric (the
    usual case, all other generic tests); a non-symmetric hs_pot must take the full-column path."""
    model, rng = build(21, 30, 4, 3, False)
    hs = model.hs_pot.copy()
    hs[5 * 21 + 7, :] += 0.01 * rng.rand(30)            # L_n[5,7] != L_n[7,5]
    model.hs_pot = hs
    nw = 40
    dev = make_device(model, nw)
    xs = rng.normal(size=(nw, 30)) + 1j * rng.normal(size=(nw, 30))
    vhs = dev.vhs(xs)
    close(vhs[:, 0], numpy.array([model.vhs(x) for x in xs]))
    assert abs(vhs[0, 0, 5, 7] - vhs[0, 0, 7, 5]) > 1e-6
    dev.close()


@pytest.mark.parametrize("M,K,na,nb,restore", [(37, 45, 7, 6, None), (120, 40, 9, 9, 'full'), (64, 12, 40, 40, None), (100, 10, 45, 45, None)])
def test_back_propagation_with_reortho(M, K, na, nb, restore):
    """afq_bp_update against the oracle with a window longer than the stabilisation period (the
    re-orthogonalisation inside the back-propagation, propagation/generic.py:286-288, is not reached by
    the reference's own 5-step test), a walker copy in the middle of the window (history travels with
    the walker) and both propagator paths (fused kernel for M=37, separate GEMMs for M=120)."""
    model, rng = build(M, K, na, nb, True)
    nw, nbp, nstblz = 5, 7, 3
    dev = make_device(model, nw)
    phis = numpy.array([model.psi + 0.1 * (rng.rand(M, na + nb) + 1j * rng.rand(M, na + nb)) for _ in range(nw)])
    dev.set(L.F_PHI, phis)
    dev.set(L.F_OT, dev.calc_overlap())
    dev.bp_configure(nbp)
    walkers = [ref.new_walker(model, p) for p in phis]
    for w in walkers:
        w['bp'] = ref.bp_new(K, nbp)
        w['phi_old'] = w['phi'].copy()
    for step in range(nbp):
        xi = rng.normal(size=(nw, K))
        dev.propagate(xi, 0.2)
        for w, x in zip(walkers, xi):
            ref.propagate_walker_phaseless(model, w, x, 0.2)
        if step == 3:
            dev.copy_walker(0, 3)
            src = walkers[0]
            walkers[3] = {k: (ref.bp_copy(v) if k == 'bp' else (v.copy() if isinstance(v, numpy.ndarray) else v))
                          for k, v in src.items()}
    assert list(dev.bp_steps()) == [nbp] * nw
    close(dev.get(L.F_WEIGHT), numpy.array([w['weight'] for w in walkers]), 1e-9)
    est = numpy.zeros(4 + 2 * M * M, dtype=complex)
    ref.bp_update(model, walkers, nstblz, est, restore, eval_energy=True)
    energies, denom, G = dev.bp_update(model.psi, nstblz, restore, eval_energy=True)
    close(denom, est[3], 1e-9)
    close(G, est[4:].reshape(2, M, M), 1e-8)
    close(energies, est[:3], 1e-8)        # full-G Cholesky energy of every G_bp (estimators/generic.py:398-434)
    assert list(dev.bp_steps()) == [0] * nw
    # phi_old <- phi: a second window starts from the propagated walkers
    xi = rng.normal(size=(nw, K))
    dev.propagate(xi, 0.2)
    for w, x in zip(walkers, xi):
        ref.propagate_walker_phaseless(model, w, x, 0.2)
    est = numpy.zeros(4 + 2 * M * M, dtype=complex)
    ref.bp_update(model, walkers, nstblz, est, restore)
    energies, denom, G = dev.bp_update(model.psi, nstblz, restore)
    close(denom, est[3], 1e-9)
    close(G, est[4:].reshape(2, M, M), 1e-8)
    assert numpy.all(energies == 0)
    dev.close()


def test_weight_cap_inside_propagate_equals_separate_cap():
    """afq_set_weight_cap (cap riding on the weight-update kernel) against afq_propagate + afq_cap_weights:
    same weights bit for bit, dead walkers capped too, switched off again by frac = 0."""
    from pauxy_amd import systems, trial as trial_mod
    from pauxy_amd.context import get_context, release_context
    from pauxy_amd.propagation import setup
    s = systems.synthetic_generic(12, 20, (3, 3), seed=9)
    t = trial_mod.rhf_trial_generic(s)
    dev = get_context(s, t).dev
    BH1, mf = setup.generic_propagator_arrays(s, t, 0.01)
    dev.set_propagator(BH1, mf, 0.01)
    nw = 9
    dev.walkers_alloc(nw)
    rng = numpy.random.RandomState(2)
    phi = numpy.asarray(t.psi)[None] + 0.05 * (rng.rand(nw, 12, 6) + 1j * rng.rand(nw, 12, 6))
    w0 = numpy.array([1.0, 3.0, 0.0, 0.2, 5.0, 1.0, 0.0, 2.5, 0.7])      # two dead walkers, several above the cap
    xi = rng.normal(size=(nw, 20))

    def run(fused):
        dev.set(L.F_PHI, phi)
        dev.set(L.F_WEIGHT, w0)
        dev.set(L.F_OT, dev.calc_overlap())
        dev.set(L.F_HYBRID_ENERGY, numpy.zeros(nw, dtype=complex))
        dev.set_weight_cap(0.1 if fused else 0.0, 9.0)
        dev.propagate(xi, 0.0)
        if not fused:
            dev.cap_weights(0.1, 9.0)
        return dev.get(L.F_WEIGHT), dev.get(L.F_PHI)

    wa, pa = run(False)
    wb, pb = run(True)
    assert numpy.array_equal(wa, wb) and numpy.array_equal(pa, pb)
    assert wa.max() <= 0.9 + 1e-15 and (wa == 0.9).sum() >= 2 and (wa[[2, 6]] == 0.0).all()
    dev.set_weight_cap(0.0)
    wc, _ = run(False)
    assert numpy.array_equal(wc, wa)
    release_context(s, t)


@pytest.mark.parametrize("M,K,na,nb,nw,cplx", [(40, 24, 6, 6, 20, False), (12, 30, 3, 0, 9, False), (19, 33, 4, 1, 33, True)])
def test_exchange_algorithms_odd_sizes(M, K, na, nb, nw, cplx):
    """Both exchange-energy algorithms (estimators/generic.py:198-216) at awkward sizes: K < M (the automatic choice is
    the T-intermediate kernel there), a fully polarised system (no beta electrons), one beta electron with a complex
    trial; each against the oracle."""
    model, rng = build(M, K, na, nb, cplx)
    phis = numpy.array([model.psi + 0.1 * (rng.rand(M, na + nb) + 1j * rng.rand(M, na + nb)) for _ in range(nw)])
    refs = [ref.greens_function(p, model.psi, na, nb) for p in phis]
    want = numpy.array([model.local_energy(r[2], r[1]) for r in refs])
    for mode in (0, 1, 2):
        dev = make_device(model, nw)
        dev.set_exchange_algorithm(mode)
        if mode == 0:
            assert dev.exchange_algorithm() == (2 if K >= M else 1)
        dev.set(L.F_PHI, phis)
        dev.greens()
        close(dev.local_energy(), want)
        dev.close()


@pytest.mark.parametrize("M,na,nb,onebody", [
    (100, 25, 25, 'same-real'),      # wide deal, every tile present (prop_fused_kernel<false, 7>); contiguous columns, 1 unit
    (100, 28, 27, 'same-real'),      # contiguous columns (one matrix for both spins): 55 columns = 3 slots + 2 units
    (97, 28, 28, 'same-real'),       # ... 56 columns, one remainder row
    (98, 30, 27, 'same-real'),       # 57 columns: the two-slots-per-spin layout
    (99, 17, 32, 'same-real'),       # ... 49 columns: one live column in the unit slot
    (100, 31, 30, 'same-real'),      # 61 columns: the two-slots-per-spin layout
    (100, 28, 25, 'spin-complex'),   # hybrid column tiling (second slot of a spin as 4-column units): 3 units, na != nb
    (99, 21, 24, 'same-real'),       # ... 2 units, three remainder rows
    (97, 17, 20, 'spin-complex'),    # ... 1 unit (one and four live columns), one remainder row
    (98, 24, 25, 'same-real'),       # unit counts differ between the spins (2 and 3): the plain deal
    (100, 29, 25, 'same-real'),      # 13 live columns in the second slot: the plain deal
    (97, 17, 32, 'spin-complex'),    # the same with spin-dependent complex one-body matrices
    (104, 30, 18, 'same-real'),
    (93, 7, 7, 'same-real'),         # one column tile per spin, six row tiles (prop_fused_kernel<true, 6>)
    (81, 16, 3, 'spin-complex'),
    (96, 9, 16, 'same-real'),
    (70, 7, 5, 'same-real'),         # narrow, generic tile tests (prop_fused_kernel<true, 0>)
    (90, 20, 18, 'spin-complex'),    # wide, six row tiles (prop_fused_kernel<false, 6>)
    (70, 30, 17, 'same-real'),       # wide, five row tiles (prop_fused_kernel<false, 5>)
    (60, 20, 18, 'spin-complex'),    # wide, generic tile tests (prop_fused_kernel<false, 0>)
    (104, 16, 17, 'same-real'),      # wide with a one-column second tile of one spin only: generic
])
def test_fused_propagator_shape_classes(M, na, nb, onebody):
    """One full step through every instantiation of the fused B exp(V) B kernel (tile-deal variants are chosen from
    M, na, nb on the host) against the oracle, with dead walkers in between."""
    K, nw = 24, 19
    model, rng = build(M, K, na, nb, False, seed=11)
    if onebody == 'spin-complex':
        # any pair of matrices is a valid one-body propagator for the kernel and for the oracle alike
        pert = 1e-3 * (rng.rand(2, M, M) + 1j * rng.rand(2, M, M))
        model.BH1 = model.BH1 + pert
    dev = make_device(model, nw)
    phis = numpy.array([model.psi + 0.1 * (rng.rand(M, na + nb) + 1j * rng.rand(M, na + nb)) for _ in range(nw)])
    dev.set(L.F_PHI, phis)
    w0 = numpy.ones(nw)
    w0[2::5] = 0.0
    dev.set(L.F_WEIGHT, w0)
    dev.set(L.F_OT, numpy.array([ref.calc_overlap(p, model.psi, na, nb) for p in phis]))
    xi = rng.normal(size=(nw, K))
    dev.propagate(xi, 0.3)
    out_phi, out_w = dev.get(L.F_PHI), dev.get(L.F_WEIGHT)
    for i in range(nw):
        if w0[i] == 0.0:
            assert numpy.array_equal(out_phi[i], phis[i]) and out_w[i] == 0.0
            continue
        w = ref.new_walker(model, phis[i])
        ref.propagate_walker_phaseless(model, w, xi[i], 0.3)
        close(out_phi[i], w['phi'])
        close(out_w[i], w['weight'])
    dev.close()


def test_spin_summed_force_bias_follows_clones_and_reorthos():
    """RHF-type trial, more than 32 walkers: the force bias contracts Ghalf_a + Ghalf_b once (both spins share the
    half-rotated Cholesky block).  The sum is written by the Green's-function kernel and must be redone after a comb
    cloned walkers (their cached Ghalf travels) -- four steps with a comb and a re-orthogonalisation in between,
    every walker against the oracle."""
    M, K, na, nb, nw = 20, 30, 4, 4, 40
    model, rng = build(M, K, na, nb, False, seed=21)
    assert numpy.array_equal(model.rchol[:na * M], model.rchol[na * M:])          # the spins do share the block
    dev = make_device(model, nw)
    phis = numpy.array([model.psi + 0.1 * (rng.rand(M, na + nb) + 1j * rng.rand(M, na + nb)) for _ in range(nw)])
    w0 = numpy.exp(0.8 * rng.normal(size=nw))
    dev.set(L.F_PHI, phis)
    dev.set(L.F_WEIGHT, w0)
    dev.set(L.F_OT, numpy.array([ref.calc_overlap(p, model.psi, na, nb) for p in phis]))
    walkers = [ref.new_walker(model, phis[i], weight=w0[i]) for i in range(nw)]
    for w in walkers:
        w['total_weight'] = nw
    cloned = False
    for step in range(1, 5):
        xi = rng.normal(size=(nw, K))
        if step == 3:
            dev.reortho(fetch=False)
            for w in walkers:
                d = ref.reortho(w['phi'], na, nb)
                w['ot'] = w['ot'] / d
        dev.propagate(xi, 0.1)
        for i, w in enumerate(walkers):
            ref.propagate_walker_phaseless(model, w, xi[i], 0.1)
        if step in (1, 2):
            r = rng.rand()
            pix, _ = dev.popcontrol_comb(r, nw)
            pix_ref = ref.pop_control(model, walkers, nw, r)
            assert numpy.array_equal(pix, pix_ref)
            cloned = cloned or (pix > 1).any()
        close(dev.get(L.F_PHI), numpy.array([w['phi'] for w in walkers]), 1e-9)
        close(dev.get(L.F_WEIGHT), numpy.array([w['weight'] for w in walkers]), 1e-9)
        close(dev.get(L.F_HYBRID_ENERGY), numpy.array([w['hybrid_energy'] for w in walkers]), 1e-9)
    assert cloned
    dev.close()


def test_fused_reortho_breakdown_fallback():
    """The one-work-group Cholesky-QR2 (N <= 32 per spin, populations of 64 walkers and more): healthy walkers against the
    oracle's QR, a numerically rank-deficient walker (Cholesky breakdown or not: either way the result must be an
    orthonormal basis of the same span) and na != nb with a spin of fewer than 16 electrons."""
    M, K, na, nb, nw = 40, 12, 19, 9, 66
    model, rng = build(M, K, na, nb, True, seed=5)
    dev = make_device(model, nw)
    phis = model.psi[None] + 0.05 * (rng.rand(nw, M, na + nb) + 1j * rng.rand(nw, M, na + nb))
    phis[2][:, 7] = phis[2][:, 3] * (1.0 + 1e-9) + 1e-10 * rng.rand(M)       # cond ~ 1e9
    phis[40][:, na + 4] = phis[40][:, na + 1] * (1.0 - 1e-9) + 1e-10 * rng.rand(M)
    dev.set(L.F_PHI, phis)
    dev.set(L.F_OT, numpy.ones(nw, dtype=complex))
    detR = dev.reortho()
    q = dev.get(L.F_PHI)
    ot = dev.get(L.F_OT)
    assert numpy.all(numpy.isfinite(detR)) and numpy.all(numpy.isfinite(q.view(float)))
    for w in range(nw):
        for sl in (slice(0, na), slice(na, na + nb)):
            close(q[w][:, sl].conj().T @ q[w][:, sl], numpy.eye(sl.stop - sl.start), 1e-10)
        if w not in (2, 40):
            p = phis[w].copy()
            d = ref.reortho(p, na, nb)
            close(q[w], p, 1e-9)
            close(detR[w], d, 1e-9)
            close(ot[w], 1.0 / d, 1e-9)
    for w, sl in ((2, slice(0, na)), (40, slice(na, na + nb))):                # span preserved for the repaired walkers
        qa = q[w][:, sl]
        close(qa @ (qa.conj().T @ phis[w][:, sl]), phis[w][:, sl], 1e-9)
    dev.close()


# ----------------------------------------------------------------------------- plane-wave step (k_ueg.hip)
def ueg_model_of(rs, nup, ndown, ecut, psi=None, dt=0.005):
    s = systems.UEG(rs, nup, ndown, ecut)
    t = trial_mod.hartree_fock_ueg(s) if psi is None else trial_mod.SingleDetTrial(s, psi(s))
    BH1, mf = setup.ueg_propagator_arrays(s, t, dt)
    H1diag = numpy.array([numpy.diag(s.H1[0]), numpy.diag(s.H1[1])])
    return ref.RefModel('ueg', s.nbasis, nup, ndown, t.psi, BH1, mf, dt, iA=s.iA, iB=s.iB, H1diag=H1diag,
                        vqvec=s.vqvec, vol=s.vol, ikpq_i=s.ikpq_i, ikpq_kpq=s.ikpq_kpq, ipmq_i=s.ipmq_i,
                        ipmq_pmq=s.ipmq_pmq, ecore=s.ecore)


def rotated_trial(s):
    """Occupied orbitals mixed among ALL plane waves: G has no zero rows, the step must take the general kernels."""
    rng = numpy.random.RandomState(17)
    M, ne = s.nbasis, s.nup + s.ndown
    a = numpy.eye(M, dtype=complex)[:, :max(s.nup, s.ndown)] + 0.05 * (rng.rand(M, max(s.nup, s.ndown)) + 1j * rng.rand(M, max(s.nup, s.ndown)))
    q, _ = numpy.linalg.qr(a)
    return numpy.concatenate([q[:, :s.nup], q[:, :s.ndown]], axis=1)


@pytest.mark.parametrize("rs,nup,ndown,ecut,psi", [(2.0, 7, 7, 4.0, None), (1.0, 7, 5, 2.5, None), (2.0, 2, 1, 1.0, None),
                                                    (2.0, 7, 7, 2.5, rotated_trial)],
                         ids=["C2-93pw", "7+5-unequal-spins", "2+1-tiny-basis", "rotated-trial-general-kernels"])
def test_planewave_step_shapes(rs, nup, ndown, ecut, psi):
    """One full step of the plane-wave path against the oracle for several shapes: the C2 system, unequal spins (the one
    16-column tile holds 7 + 5 columns), a basis of a handful of plane waves (one row tile, contraction of 2 k-steps per
    half) and a trial whose orbitals mix all plane waves (no zero rows of G: the checks of k_ueg.hip send the step to the
    general kernels).  Dead walkers in between; the force bias the step used (xbar) against the stand-alone
    afq_force_bias of the general gather kernel, its shifted fields against the oracle's."""
    model = ueg_model_of(rs, nup, ndown, ecut, psi)
    M, nt, K = model.M, nup + ndown, model.nfields
    nw = 21
    rng = numpy.random.RandomState(3)
    dev = make_device(model, nw)
    phis = numpy.array([model.psi + 0.05 * (rng.rand(M, nt) + 1j * rng.rand(M, nt)) for _ in range(nw)])
    dev.set(L.F_PHI, phis)
    w0 = numpy.ones(nw)
    w0[3::7] = 0.0
    dev.set(L.F_WEIGHT, w0)
    dev.set(L.F_OT, numpy.array([ref.calc_overlap(p, model.psi, nup, ndown) for p in phis]))
    dev.greens(want_G=True)
    xbar_general = dev.force_bias()                       # afq_force_bias: full G + sparse gather (k_models.hip)
    xi = rng.normal(size=(nw, K))
    dev.propagate(xi, 0.2)
    out_phi, out_w, out_e = dev.get(L.F_PHI), dev.get(L.F_WEIGHT), dev.get(L.F_HYBRID_ENERGY)
    xbar_step, xs_step = dev.get(L.F_XBAR), dev.get(L.F_XSHIFTED)
    for i in range(nw):
        if w0[i] == 0.0:
            assert numpy.array_equal(out_phi[i], phis[i]) and out_w[i] == 0.0
            continue
        close(xbar_step[i], xbar_general[i], 1e-12)
        w = ref.new_walker(model, phis[i])
        ref.propagate_walker_phaseless(model, w, xi[i], 0.2)
        close(out_phi[i], w['phi'])
        close(out_w[i], w['weight'])
        close(out_e[i], w['hybrid_energy'])
        close(xs_step[i], xi[i] - xbar_general[i], 1e-12)
    dev.close()


@pytest.mark.parametrize("na,nb", [(n, max(n - (n % 3), 0)) for n in range(1, 33)])
def test_greens_every_electron_count(na, nb):
    """The register-resident Gauss-Jordan keeps ceil(n / 2) columns per lane rounded up to 4 / 8 / 12 / 13 / 16 registers
    (gj_wave.h): overlap, Ghalf and G against the oracle for every n <= 32, unequal spins, walkers far from the trial
    (the pivot order is not the identity)."""
    M, nw = 40, 6
    model, rng = build(M, 12, na, nb, True, seed=na)
    dev = make_device(model, nw)
    phi = rng.rand(nw, M, na + nb) - 0.5 + 1j * (rng.rand(nw, M, na + nb) - 0.5)
    dev.set(L.F_PHI, phi)
    ot = dev.greens(want_G=True)
    G, gh = dev.get(L.F_G), dev.get(L.F_GHALF)
    for w in range(nw):
        det, ghalf_ref, G_ref = ref.greens_function(phi[w], model.psi, na, nb)
        assert abs(ot[w] - det) <= 1e-10 * abs(det)
        close(G[w], G_ref, 1e-9)
        close(gh[w].reshape(na + nb, M), numpy.concatenate([g for g in ghalf_ref if g is not None and len(g)]), 1e-9)
    dev.close()


@pytest.mark.parametrize("M", [17, 63, 64, 65, 97, 113, 128])
@pytest.mark.parametrize("cplx", [False, True])
def test_greens_basis_sizes_real_and_complex_trial(M, cplx):
    """Overlap phase of the small Green's function kernel: k-steps in groups of four with the trial fragments requested in
    two halves of sixteen (M <= 64: one half), ragged M (the last group runs past M), two products per k-step for a real
    trial and three for a complex one; 9 + 8 electrons (two row / column tiles would need n > 16: one tile per spin here)
    and 20 + 19 (four tiles per spin)."""
    for na, nb in ((9, 8), (20, 19)):
        if na >= M:
            continue
        model, rng = build(M, 6, na, nb, cplx, seed=M)
        nw = 5
        dev = make_device(model, nw)
        phi = rng.rand(nw, M, na + nb) - 0.5 + 1j * (rng.rand(nw, M, na + nb) - 0.5)
        dev.set(L.F_PHI, phi)
        ot = dev.greens(want_G=False)
        gh = dev.get(L.F_GHALF)
        for w in range(nw):
            det, ghalf_ref, _ = ref.greens_function(phi[w], model.psi, na, nb)
            assert abs(ot[w] - det) <= 1e-10 * abs(det)
            close(gh[w].reshape(na + nb, M), numpy.concatenate(ghalf_ref), 1e-9)
        dev.close()


@pytest.mark.parametrize("M,na,nb,cplx", [(64, 40, 37, False), (100, 45, 33, True), (90, 33, 33, False)])
def test_greens_33_to_45_electrons(M, na, nb, cplx):
    """33 .. 45 electrons per spin: nine overlap tiles per spin (up to three per wave: the trial fragments of the second and
    third are fetched inside the loop) and the LDS Gauss-Jordan of the small Green's function kernel."""
    model, rng = build(M, 6, na, nb, cplx, seed=M + na)
    nw = 4
    dev = make_device(model, nw)
    phi = rng.rand(nw, M, na + nb) - 0.5 + 1j * (rng.rand(nw, M, na + nb) - 0.5)
    dev.set(L.F_PHI, phi)
    ot = dev.greens(want_G=False)
    gh = dev.get(L.F_GHALF)
    for w in range(nw):
        det, ghalf_ref, _ = ref.greens_function(phi[w], model.psi, na, nb)
        assert abs(ot[w] - det) <= 1e-9 * abs(det)
        close(gh[w].reshape(na + nb, M), numpy.concatenate(ghalf_ref), 1e-8)
    dev.close()


@pytest.mark.parametrize("M,na,nb", [(140, 64, 40), (150, 128, 97), (136, 33, 48)])
def test_blocked_gauss_jordan_hands_badly_placed_pivots_to_the_step_by_step_kernel(M, na, nb):
    """The blocked Gauss-Jordan of k_bigdet.hip (more than 32 electrons per spin) pivots inside 16 x 16 tiles only.  Walkers
    whose overlap matrix is well conditioned as a whole but has (1) a vanishing leading tile (a permutation: the first
    orbitals of the walker are orthogonal to the first orbitals of the trial), (2) a leading tile 1e-10 times the rest
    (unbounded multipliers), (3) one vanishing diagonal tile further down, must come out as exact as the others: the kernel
    flags them (pivot spread of a tile, growth guard) and the step-by-step kernel with full partial pivoting redoes them.
    Reference: walkers/single_det.py:295-321 (scipy.linalg.inv: LAPACK partial pivoting)."""
    model, rng = build(M, 6, na, nb, True, seed=M + na)
    nw = 6
    nt = na + nb
    phi = rng.rand(nw, M, nt) - 0.5 + 1j * (rng.rand(nw, M, nt) - 0.5)

    def with_overlap(T, psi_s):
        # phi_s with psi_s^H phi_s = T
        return psi_s @ numpy.linalg.solve(psi_s.conj().T @ psi_s, T)

    for s, (o, n) in enumerate(((0, na), (na, nb))):
        psi_s = model.psi[:, o:o + n]
        noise = 1e-3 * (rng.rand(n, n) - 0.5 + 1j * (rng.rand(n, n) - 0.5))
        phi[1][:, o:o + n] = with_overlap(numpy.eye(n)[::-1] + 0j, psi_s)                 # (1) exact zeros in the tile
        T = rng.rand(n, n) - 0.5 + 1j * (rng.rand(n, n) - 0.5)
        T[:16, :16] *= 1e-10
        phi[2][:, o:o + n] = with_overlap(T, psi_s)                                        # (2)
        T = numpy.eye(n) + noise
        if n > 40:
            T[16:32, 16:32] = 0.0
            T[16:32, 32:48] += numpy.eye(16)
            T[32:48, 16:32] += numpy.eye(16)
            T[32:48, 32:48] -= numpy.eye(16)
        phi[3][:, o:o + n] = with_overlap(T, psi_s)                                        # (3)
        phi[4][:, o:o + n] = with_overlap(numpy.eye(n)[::-1] + noise, psi_s)               # (1) with noise in the tile
    dev = make_device(model, nw)
    dev.set(L.F_PHI, phi)
    dev.counters(reset=True)
    ot = dev.greens(want_G=False)
    gh = dev.get(L.F_GHALF)
    # afq_counters [2]: matrices the fallback pass actually processed (bench.py prices that launch by it): the
    # constructed walkers with more than 32 electrons in the spin, and none of the two ordinary ones
    flagged = int(dev.counters()[2])
    assert 1 <= flagged <= 8, flagged
    for w in range(nw):
        det, ghalf_ref, _ = ref.greens_function(phi[w], model.psi, na, nb)
        assert abs(ot[w] - det) <= 1e-9 * abs(det), (w, ot[w], det)
        close(gh[w].reshape(nt, M), numpy.concatenate(ghalf_ref), 1e-8)
    dev.set(L.F_PHI, numpy.array([phi[0]] * nw))
    dev.counters(reset=True)
    dev.greens(want_G=False)
    assert int(dev.counters()[2]) == 0
    dev.close()


@pytest.mark.parametrize("M,na,nb", [(140, 64, 40), (150, 128, 97), (136, 33, 48), (130, 113, 17)])
def test_blocked_cholesky_reortho_every_tile_count(M, na, nb):
    """Re-orthogonalisation above 32 electrons per spin: Gram GEMM + blocked inverse Cholesky factor (chol_mfma_kernel:
    16 x 16 tiles, 3 to 8 of them per spin, ragged last tile, unequal spins, one spin below 32) + Q GEMM, twice; every
    walker against the oracle's QR with the diagonal of R made positive (walkers/single_det.py:215-255), detR included."""
    model, rng = build(M, 6, na, nb, True, seed=M + nb)
    nw = 5
    nt = na + nb
    phi = model.psi[None] + 0.3 * (rng.rand(nw, M, nt) - 0.5 + 1j * (rng.rand(nw, M, nt) - 0.5))
    dev = make_device(model, nw)
    dev.set(L.F_PHI, phi)
    dev.set(L.F_OT, numpy.ones(nw, dtype=complex))
    detR = dev.reortho()
    q = dev.get(L.F_PHI)
    for w in range(nw):
        p = phi[w].copy()
        d = ref.reortho(p, na, nb)
        close(q[w], p, 1e-9)
        close(detR[w], d, 1e-9)
        for sl in (slice(0, na), slice(na, nt)):
            close(q[w][:, sl].conj().T @ q[w][:, sl], numpy.eye(sl.stop - sl.start), 1e-11)
    dev.close()


def test_dead_walkers_pass_through_the_large_one_body_products_untouched():
    """Dead walkers are not propagated (qmc/afqmc.py:232) but the one-body product writes into the other buffer of a
    ping-pong pair: the work-groups of the ring GEMM engine (M > 128, 64 walkers and more) copy a dead walker's columns
    through instead of multiplying them (INACTIVE_COPY), here on the un-fused generic step -- B, Taylor products, B -- and
    on the Hubbard step whose field kernel makes the diagonal potential's Taylor factors.  Dead walkers bit-equal, live ones
    against the oracle."""
    from pauxy_amd.propagation import setup as psetup
    nw = 64
    # generic, M = 140, 40 + 37 electrons
    model, rng = build(140, 12, 40, 37, True, seed=5)
    s = systems.Hubbard(12, 12, 60, 57, 4.0)
    th = trial_mod.uhf_trial_hubbard(s, ueff=0.4)
    BH1, mf = psetup.hubbard_propagator_arrays(s, th, 0.01, True)
    hub = ref.RefModel('hubbard', 144, 60, 57, th.psi, BH1, mf, 0.01, U=4.0, H1=s.T.astype(complex))
    for m in (model, hub):
        M, nt, K = m.M, m.na + m.nb, m.nfields
        r = numpy.random.RandomState(9)
        dev = make_device(m, nw)
        phis = m.psi[None] + 0.05 * (r.rand(nw, M, nt) + 1j * r.rand(nw, M, nt))
        dev.set(L.F_PHI, phis)
        w0 = numpy.ones(nw)
        w0[1::5] = 0.0
        dev.set(L.F_WEIGHT, w0)
        dev.set(L.F_OT, numpy.array([ref.calc_overlap(p, m.psi, m.na, m.nb) for p in phis]))
        xi = r.normal(size=(nw, K))
        dev.propagate(xi, 0.1)
        out_phi, out_w = dev.get(L.F_PHI), dev.get(L.F_WEIGHT)
        for i in range(nw):
            if w0[i] == 0.0:
                assert numpy.array_equal(out_phi[i], phis[i]) and out_w[i] == 0.0
        for i in (0, 2, 33, 63):
            w = ref.new_walker(m, phis[i])
            ref.propagate_walker_phaseless(m, w, xi[i], 0.1)
            close(out_phi[i], w['phi'], 1e-9)
            close(out_w[i], w['weight'], 1e-9)
        dev.close()


def _component_errors(a, b):
    """(normwise error, componentwise error over the significant components): real and imaginary parts are separate
    components; one is significant when it is at least 1e-4 of the largest magnitude of the array."""
    a, b = numpy.asarray(a, dtype=complex), numpy.asarray(b, dtype=complex)
    scale = float(numpy.max(numpy.abs(b)))
    norm = float(numpy.max(numpy.abs(a - b))) / scale
    comp = 0.0
    for part in (numpy.real, numpy.imag):
        x, y = part(a), part(b)
        sig = numpy.abs(y) >= 1e-4 * scale
        if numpy.any(sig):
            comp = max(comp, float(numpy.max(numpy.abs(x[sig] - y[sig]) / numpy.abs(y[sig]))))
    return norm, comp


@pytest.mark.parametrize("M,na,nb,cplx_trial", [(100, 25, 25, False), (100, 25, 25, True), (110, 50, 50, False), (110, 50, 50, True)])
@pytest.mark.parametrize("scaling", ["re>>im", "im>>re", "real", "imaginary"])
def test_badly_scaled_operands_through_the_three_multiplication_products(M, na, nb, cplx_trial, scaling):
    """The complex products of the overlap / Ghalf GEMMs, the Taylor chain and the blocked Gauss-Jordan are
    3-multiplication (Karatsuba) products: Im = (Ar + Ai)(Br + Bi) - Ar Br - Ai Bi.  Their error bound is NORMWISE
    (~eps |A| |B|), not componentwise: a component 1e-8 of the other one is computed to 1e-8 of itself, which is 1e-16 of
    the number.  All other tests draw Re ~ Im; here the walkers are badly scaled -- |Re| / |Im| = 1e8 and 1e-8, purely real,
    purely imaginary -- through the Green's function (walkers/single_det.py:295-321), one propagation step
    (continuous.py:232-262: B exp(V) B with the Taylor-6 chain) and the overlap of the propagated walker, for the one
    work-group kernels (25 + 25 electrons, fused propagator) and the GEMM / blocked Gauss-Jordan path (50 + 50 electrons,
    M = 110: un-fused Taylor GEMM chain), real and complex trial.  Pinned: 1e-13 NORMWISE on everything (measured on an
    MI355X: 4e-16 ... 1e-14), and 1e-10 COMPONENTWISE on every real or imaginary part within four decades of the largest
    number of its array (measured: up to 4e-12; a part of relative size s carries the absolute rounding error of the O(1)
    terms it is the sum of, eps / s of itself, in the reference's LAPACK / BLAS arithmetic as much as here -- with six decades
    the oracle-against-device difference is 1e-10 for that reason alone).  The small parts of the badly scaled numbers
    themselves (1e-8 of their number) are below that line: the 3-multiplication product returns them to ~1e-8 of
    themselves, i.e. 1e-16 of the number -- a phase error of 1e-16 rad in the importance function (continuous.py:264-292)."""
    nt = na + nb
    model, rng = build(M, 40, na, nb, cplx_trial, seed=M + na)
    nw = 6
    a, b = rng.rand(nw, M, nt) - 0.5, rng.rand(nw, M, nt) - 0.5
    base = numpy.real(model.psi)[None] + 0.1 * a
    other = numpy.real(model.psi)[None] + 0.1 * b
    phis = {"re>>im": base + 1e-8j * other, "im>>re": 1e-8 * other + 1j * base, "real": base + 0j, "imaginary": 1j * base}[scaling]
    dev = make_device(model, nw)
    dev.set(L.F_PHI, phis)
    det = dev.greens(want_G=False)
    gh = dev.get(L.F_GHALF)
    refs = [ref.greens_function(p, model.psi, na, nb) for p in phis]
    checks = {"det": _component_errors(det, numpy.array([r[0] for r in refs])),
              "ghalf": _component_errors(gh.reshape(nw, nt, M), numpy.array([numpy.concatenate(r[1]) for r in refs]))}
    dev.set(L.F_OT, det)
    walkers = [ref.new_walker(model, p) for p in phis]
    xi = rng.normal(size=(nw, 40))
    dev.propagate(xi, -0.2)
    for w, x in zip(walkers, xi):
        ref.propagate_walker_phaseless(model, w, x, -0.2)
    checks["phi"] = _component_errors(dev.get(L.F_PHI), numpy.array([w['phi'] for w in walkers]))
    checks["ot"] = _component_errors(dev.get(L.F_OT), numpy.array([w['ot'] for w in walkers]))
    checks["weight"] = _component_errors(dev.get(L.F_WEIGHT), numpy.array([w['weight'] for w in walkers]))
    dev.close()
    print("badly scaled %s M=%d N=%d cplx=%s: " % (scaling, M, na, cplx_trial) +
          "  ".join("%s %.1e/%.1e" % (k, v[0], v[1]) for k, v in checks.items()))
    for k, (norm, comp) in checks.items():
        assert norm <= 1e-13, (k, norm)
        assert comp <= 1e-10, (k, comp)


@pytest.mark.parametrize("M,N", [(100, 25), (100, 27), (98, 28), (40, 13), (64, 32)])
def test_closed_shell_walkers_take_the_one_spin_path_of_the_greens_kernel(M, N):
    """An RHF trial (alpha block == beta block) and walkers whose spin blocks are bitwise equal -- what an RHF run with a
    spin-free propagator produces -- let greens_small_kernel compute ONE spin (checked per walker, in LDS, every call):
    O_b = O_a, det = det_a^2, Ghalf_b = Ghalf_a.  A population that mixes such walkers with ordinary ones (beta block
    different) against the oracle for both kinds (walkers/single_det.py:295-321), the closed ones with bitwise equal spin
    blocks of Ghalf; then a step, re-orthogonalisation and a second step keep the closed walkers closed bit for bit
    (propagation/continuous.py:232-262 acts on both spins alike) -- the invariant the fast path rests on."""
    nt, K, nw = 2 * N, 30, 40
    s = systems.synthetic_generic(M, K, (N, N), seed=7)
    t = trial_mod.rhf_trial_generic(s)
    assert numpy.array_equal(t.psi[:, :N], t.psi[:, N:])
    dt = 0.005
    BH1, mf = setup.generic_propagator_arrays(s, t, dt)
    model = ref.RefModel('generic', M, N, N, t.psi, BH1, mf, dt, hs_pot=s.hs_pot, rchol=t._rchol,
                         H1=s.H1.astype(complex), ecore=s.ecore)
    rng = numpy.random.RandomState(M + N)
    half = t.psi[None, :, :N] + 0.05 * (rng.rand(nw, M, N) + 1j * rng.rand(nw, M, N))
    phis = numpy.concatenate([half, half], axis=2)
    mixed = numpy.arange(nw) % 3 == 2                                   # every third walker: an ordinary (open) one
    phis[mixed, :, N:] += 0.03 * (rng.rand(int(mixed.sum()), M, N) + 1j * rng.rand(int(mixed.sum()), M, N))
    dev = make_device(model, nw)
    dev.set(L.F_PHI, phis)
    det = dev.greens(want_G=False)
    gh = dev.get(L.F_GHALF).reshape(nw, nt, M)
    for w in range(nw):
        d_ref, gh_ref, _ = ref.greens_function(phis[w], model.psi, N, N)
        assert abs(det[w] - d_ref) <= 1e-10 * abs(d_ref)
        close(gh[w], numpy.concatenate(gh_ref), 1e-10)
        if not mixed[w]:
            assert numpy.array_equal(gh[w, :N], gh[w, N:])
        else:
            assert not numpy.array_equal(gh[w, :N], gh[w, N:])
    # force bias (the spin sum the kernel leaves behind) for both kinds
    xbar = dev.force_bias()
    for w in (0, 1, 2, 5, nw - 1):
        _, gh_ref, G_ref = ref.greens_function(phis[w], model.psi, N, N)
        close(xbar[w], model.force_bias(gh_ref, G_ref), 1e-10)
    # the exchange energy of a population that is NOT closed as a whole: the device flag sends the beta launch through
    E = dev.local_energy()
    for w in (0, 1, 2, 5, nw - 1):
        _, gh_ref, G_ref = ref.greens_function(phis[w], model.psi, N, N)
        close(E[w], numpy.array(model.local_energy(G_ref, gh_ref)), 1e-10)
    # ... and of a closed one (every walker): one spin evaluated, counted twice -- against the oracle, and against the two-spin
    # evaluation of the same walkers (the T-intermediate algorithm, which has no one-spin path)
    closed_phis = numpy.concatenate([half, half], axis=2)
    dev.set(L.F_PHI, closed_phis)
    dev.greens(want_G=False)
    E_closed = dev.local_energy()
    for w in (0, 2, 7, nw - 1):
        _, gh_ref, G_ref = ref.greens_function(closed_phis[w], model.psi, N, N)
        close(E_closed[w], numpy.array(model.local_energy(G_ref, gh_ref)), 1e-10)
    dev.set_exchange_algorithm(1)
    dev.greens(want_G=False)
    close(dev.local_energy(), E_closed, 1e-11)
    dev.set_exchange_algorithm(0)
    dev.set(L.F_PHI, phis)
    det = dev.greens(want_G=False)
    # the invariant: closed walkers stay closed through steps and the re-orthogonalisation, bit for bit
    dev.set(L.F_OT, det)
    dev.counters(reset=True)
    for step in range(2):
        xi = rng.normal(size=(nw, K))
        dev.propagate(xi, -0.1)
        if step == 0:
            # (afq_counters [3]: the closed-shell deal exists for 96 < M <= 100, 48 < 2 N <= 56 and ran for exactly the closed walkers)
            took = int(dev.counters()[3])
            assert took == (int((~mixed).sum()) if (96 < M <= 100 and 48 < 2 * N <= 56) else 0), took
            # the step itself against the oracle, closed walkers (the fused propagator's closed-shell deal at 48 < 2 N <= 56:
            # column slot 1 left out of the Taylor products and copied back) and open ones (the full deal) alike
            out = dev.get(L.F_PHI)
            for w in (0, 1, 2, 3, 5, nw - 1):
                wk = ref.new_walker(model, phis[w])
                ref.propagate_walker_phaseless(model, wk, xi[w], -0.1)
                close(out[w], wk['phi'], 1e-10)
            dev.reortho()
    out = dev.get(L.F_PHI)
    for w in range(nw):
        assert numpy.array_equal(out[w, :, :N], out[w, :, N:]) == (not mixed[w]), w
    dev.close()


@pytest.mark.parametrize("M,N,deal", [
    (100, 25, True),     # contiguous columns, twins in like slots: column slot 1 left out
    (100, 30, True),     # two slots per spin, seven row tiles with the 4-row remainder unit: the alpha half only
    (100, 20, True),     # ... 40 columns
    (104, 20, True),     # ... seven full row tiles
    (90, 20, True),      # six row tiles (prop_fused_kernel<false, 6>)
    (70, 18, True),      # five row tiles (prop_fused_kernel<false, 5>)
    (93, 7, True),       # one column tile per spin, six row tiles (prop_fused_kernel<true, 6>)
    (70, 7, False),      # narrow, generic tile tests: no closed-shell deal
    (60, 20, False),     # wide, generic tile tests: no closed-shell deal
])
def test_fused_propagator_closed_shell_deals(M, N, deal):
    """Walkers whose spin blocks are bitwise equal take the fused propagator's closed-shell deal (the kernel checks every walker
    on its LDS image): the Taylor products of the alpha half only -- of everything but the redundant slot in the
    contiguous-column layout -- copied into the beta half ahead of the closing one-body pass.  A population of closed, open and
    dead walkers, two steps against the oracle (propagation/continuous.py:232-262); afq_counters [3] counts exactly the live
    closed walkers; closed walkers stay closed bit for bit."""
    K, nw = 24, 21
    model, rng = build(M, K, N, N, False, seed=31)
    assert numpy.array_equal(model.BH1[0], model.BH1[1])
    half = model.psi[None, :, :N] + 0.1 * (rng.rand(nw, M, N) + 1j * rng.rand(nw, M, N))
    phis = numpy.concatenate([half, half], axis=2)
    is_open = numpy.arange(nw) % 3 == 1
    phis[is_open, :, N:] += 0.05 * (rng.rand(int(is_open.sum()), M, N) + 1j * rng.rand(int(is_open.sum()), M, N))
    w0 = numpy.ones(nw)
    w0[4::7] = 0.0
    dev = make_device(model, nw)
    dev.set(L.F_PHI, phis)
    dev.set(L.F_WEIGHT, w0)
    dev.set(L.F_OT, numpy.array([ref.calc_overlap(p, model.psi, N, N) for p in phis]))
    walkers = [ref.new_walker(model, phis[i]) for i in range(nw)]
    dev.counters(reset=True)
    live_closed = int(((w0 > 0) & ~is_open).sum())
    for step in range(2):
        xi = rng.normal(size=(nw, K))
        dev.propagate(xi, 0.2)
        out_phi, out_w = dev.get(L.F_PHI), dev.get(L.F_WEIGHT)
        for i in range(nw):
            if w0[i] == 0.0:
                assert numpy.array_equal(out_phi[i], phis[i]) and out_w[i] == 0.0
                continue
            ref.propagate_walker_phaseless(model, walkers[i], xi[i], 0.2)
            close(out_phi[i], walkers[i]['phi'], 1e-10 * (step + 1))
            close(out_w[i], walkers[i]['weight'], 1e-10 * (step + 1))
            assert numpy.array_equal(out_phi[i, :, :N], out_phi[i, :, N:]) == (not is_open[i]), (step, i)
        assert int(dev.counters()[3]) == (live_closed * (step + 1) if deal else 0)
    dev.close()


@pytest.mark.parametrize("M,N", [(136, 40), (200, 50), (150, 70)])
def test_large_system_chain_leaves_out_the_beta_tiles_of_closed_shell_walkers(M, N):
    """The GEMM chain of the large systems (M > 128: k_onebody / k_apply_exponential) checks at every step which walkers have
    bitwise equal spin blocks and does not compute the work-group tiles that lie wholly in their beta columns; the propagated
    alpha block is copied over the beta block behind the closing one-body product.  A population of closed, open and dead
    walkers (at least 64: the ring path), two steps against the oracle (propagation/continuous.py:232-262); closed walkers stay
    closed bit for bit, open ones open, dead ones untouched."""
    K, nw = 16, 66
    model, rng = build(M, K, N, N, False, seed=37)
    assert numpy.array_equal(model.BH1[0], model.BH1[1])
    half = model.psi[None, :, :N] + 0.1 * (rng.rand(nw, M, N) + 1j * rng.rand(nw, M, N))
    phis = numpy.concatenate([half, half], axis=2)
    is_open = numpy.arange(nw) % 3 == 1
    phis[is_open, :, N:] += 0.05 * (rng.rand(int(is_open.sum()), M, N) + 1j * rng.rand(int(is_open.sum()), M, N))
    w0 = numpy.ones(nw)
    w0[4::7] = 0.0
    dev = make_device(model, nw)
    dev.set(L.F_PHI, phis)
    dev.set(L.F_WEIGHT, w0)
    dev.set(L.F_OT, numpy.array([ref.calc_overlap(p, model.psi, N, N) for p in phis]))
    sample = [i for i in range(nw) if i % 5 in (0, 1)]              # oracle on a sample (closed, open and dead ones in it)
    walkers = {i: ref.new_walker(model, phis[i]) for i in sample}
    dev.counters(reset=True, n=8)
    for step in range(2):
        xi = rng.normal(size=(nw, K))
        dev.propagate(xi, 0.2)
        out_phi, out_w = dev.get(L.F_PHI), dev.get(L.F_WEIGHT)
        for i in range(nw):
            if w0[i] == 0.0:
                assert numpy.array_equal(out_phi[i], phis[i]) and out_w[i] == 0.0
                continue
            assert numpy.array_equal(out_phi[i, :, :N], out_phi[i, :, N:]) == (not is_open[i]), (step, i)
            if i in walkers:
                ref.propagate_walker_phaseless(model, walkers[i], xi[i], 0.2)
                close(out_phi[i], walkers[i]['phi'], 1e-10 * (step + 1))
                close(out_w[i], walkers[i]['weight'], 1e-10 * (step + 1))
    # afq_counters_ext [7]: every step of every live closed walker went through the chain as a closed-shell walker
    assert int(dev.counters(n=8)[7]) == 2 * int(((w0 > 0) & ~is_open).sum())
    dev.close()


@pytest.mark.parametrize("M,N", [(136, 40), (150, 70)])
def test_large_system_exchange_energy_evaluates_one_spin_of_a_closed_shell_population(M, N):
    """The Green's function of the large systems (k_greens_big: three GEMM-shaped launches) holds no whole walker anywhere, so
    the exchange energy compares the spin blocks of the Ghalf it is about to contract itself (ghalf_closed_check_kernel) and
    evaluates ONE spin when every walker's are bitwise equal (estimators/generic.py:84-119 contracts each spin alike).  A closed
    population, then one with an open walker in it, against the oracle; the two-spin evaluation of the same closed walkers (the
    T-intermediate algorithm) agrees; afq_counters_ext [4] counts the one-spin walker evaluations."""
    K, nw = 12, 40
    model, rng = build(M, K, N, N, False, seed=41)
    half = model.psi[None, :, :N] + 0.1 * (rng.rand(nw, M, N) + 1j * rng.rand(nw, M, N))
    closed_phis = numpy.concatenate([half, half], axis=2)
    dev = make_device(model, nw)
    dev.set_exchange_algorithm(2)                                   # the quadratic form (the automatic choice wants K >= M)
    dev.set(L.F_PHI, closed_phis)
    dev.greens(want_G=False)
    gh = dev.get(L.F_GHALF).reshape(nw, 2 * N, M)
    assert all(numpy.array_equal(g[:N], g[N:]) for g in gh)         # the launches of the large-system path treat both spins alike
    dev.counters(reset=True, n=8)
    E_closed = dev.local_energy()
    assert int(dev.counters(n=8)[4]) == nw
    for w in (0, 3, nw - 1):
        _, gh_ref, G_ref = ref.greens_function(closed_phis[w], model.psi, N, N)
        close(E_closed[w], numpy.array(model.local_energy(G_ref, gh_ref)), 1e-10)
    dev.set_exchange_algorithm(1)
    dev.greens(want_G=False)
    close(dev.local_energy(), E_closed, 1e-11)
    dev.set_exchange_algorithm(2)
    # one open walker: the verdict is the population's, both spins are evaluated for everybody
    phis = closed_phis.copy()
    phis[5, :, N:] += 0.05 * (rng.rand(M, N) + 1j * rng.rand(M, N))
    dev.set(L.F_PHI, phis)
    dev.greens(want_G=False)
    dev.counters(reset=True, n=8)
    E = dev.local_energy()
    assert int(dev.counters(n=8)[4]) == 0
    for w in (0, 5, nw - 1):
        _, gh_ref, G_ref = ref.greens_function(phis[w], model.psi, N, N)
        close(E[w], numpy.array(model.local_energy(G_ref, gh_ref)), 1e-10)
    mask = numpy.arange(nw) != 5
    close(E[mask], E_closed[mask], 1e-13)                          # one spin counted twice == both spins evaluated
    dev.close()


def test_exchange_energy_with_both_spins_slices_in_one_launch():
    """256 walkers at the benchmark's shape (M = 100, 25 + 25): each spin's contraction slices make one whole round over the eight
    XCDs, so the closed-shell try puts the beta slices behind the alpha slices of ONE launch (launch_exx_quadratic: `merged`) --
    work-groups that return at once for a closed population, that multiply for an open one.  Both populations against the oracle
    (estimators/generic.py:156-221) and against the T-intermediate algorithm; afq_counters_ext [4] says which form ran."""
    M, N, K, nw = 100, 25, 30, 256
    s = systems.synthetic_generic(M, K, (N, N), seed=19)
    t = trial_mod.rhf_trial_generic(s)
    BH1, mf = setup.generic_propagator_arrays(s, t, 0.005)
    model = ref.RefModel('generic', M, N, N, t.psi, BH1, mf, 0.005, hs_pot=s.hs_pot, rchol=t._rchol,
                         H1=s.H1.astype(complex), ecore=s.ecore)
    rng = numpy.random.RandomState(23)
    half = t.psi[None, :, :N] + 0.05 * (rng.rand(nw, M, N) + 1j * rng.rand(nw, M, N))
    closed_phis = numpy.concatenate([half, half], axis=2)
    dev = make_device(model, nw)
    dev.set_exchange_algorithm(2)
    sample = (0, 7, 100, nw - 1)

    def check(phis, E, which):
        for w in which:
            _, gh_ref, G_ref = ref.greens_function(phis[w], model.psi, N, N)
            close(E[w], numpy.array(model.local_energy(G_ref, gh_ref)), 1e-10)
    dev.set(L.F_PHI, closed_phis)
    dev.greens(want_G=False)
    dev.counters(reset=True, n=8)
    E_closed = dev.local_energy()
    assert int(dev.counters(n=8)[4]) == nw
    check(closed_phis, E_closed, sample)
    dev.set_exchange_algorithm(1)
    dev.greens(want_G=False)
    close(dev.local_energy(), E_closed, 1e-11)
    dev.set_exchange_algorithm(2)
    phis = closed_phis.copy()
    phis[100, :, N:] += 0.03 * (rng.rand(M, N) + 1j * rng.rand(M, N))
    dev.set(L.F_PHI, phis)
    dev.greens(want_G=False)
    dev.counters(reset=True, n=8)
    E = dev.local_energy()
    assert int(dev.counters(n=8)[4]) == 0
    check(phis, E, sample)
    mask = numpy.arange(nw) != 100
    close(E[mask], E_closed[mask], 1e-13)
    dev.close()
