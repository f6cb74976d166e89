"""Per-operation parity of the HIP library (through the C ABI) against the CPU
oracle and the golden vectors of the genuine reference.  fp64; tolerance 1e-10
relative to the largest element (LAPACK/MFMA summation-order differences)."""
import numpy
import pytest

from oracle import afqmc_ref as ref
from tests.helpers import generic_model, hubbard_model, ueg_model, make_device
from pauxy_amd import _lib as L

pytestmark = pytest.mark.gpu
TOL = 1e-10


def close(a, b, tol=TOL):
    a = numpy.asarray(a)
    b = numpy.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(numpy.max(numpy.abs(b)))) if b.size else 1.0
    err = float(numpy.max(numpy.abs(a - b))) / scale
    assert err <= tol, err


def perturbed(phi, nw, seed=3):
    rng = numpy.random.RandomState(seed)
    out = numpy.array([phi + 0.05 * (rng.rand(*phi.shape) + 1j * rng.rand(*phi.shape)) for _ in range(nw)])
    out[0] = phi
    return out


def check_ops(d, tag, model, nw=5):
    na, nb = model.na, model.nb
    dev = make_device(model, nw)
    phis = perturbed(d[tag + 'phi'], nw)
    dev.set(L.F_PHI, phis)
    # --- Green's function / overlaps: golden (walker 0) and oracle (all)
    det = dev.greens(want_G=True)
    gh = dev.get(L.F_GHALF)
    G = dev.get(L.F_G)
    close(det[0], d[tag + 'det'])
    close(gh[0, :na], d[tag + 'Ghalf_a'])
    close(gh[0, na:], d[tag + 'Ghalf_b'])
    close(G[0], d[tag + 'G'])
    refs = [ref.greens_function(phis[i], model.psi, na, nb) for i in range(nw)]
    close(det, numpy.array([r[0] for r in refs]))
    close(gh, numpy.array([numpy.concatenate([r[1][0], r[1][1]]) for r in refs]))
    close(G, numpy.array([r[2] for r in refs]))
    ov = dev.calc_overlap()
    close(ov, det)
    # --- force bias
    xbar = dev.force_bias()
    close(xbar[0], d[tag + 'xbar'])
    close(xbar, numpy.array([model.force_bias(r[1], r[2]) for r in refs]))
    # --- clip / shift
    xi = numpy.tile(d[tag + 'xi'], (nw, 1))
    big = numpy.tile(d[tag + 'xbar_big'], (nw, 1))
    xs, cmf, cfb = dev.shift_fields(xi, big)
    close(xs[0], d[tag + 'xs_clip'])
    close(cmf[0], d[tag + 'cmf_clip'])
    close(cfb[0], d[tag + 'cfb_clip'])
    # --- HS potential
    xs_ref = numpy.array([ref.shift_fields(xi[i], xbar[i], model.mf_shift, model.sqrt_dt)[0] for i in range(nw)])
    vhs = dev.vhs(xs_ref)
    gold = d[tag + 'VHS']
    close(vhs[0] if gold.ndim == 3 else vhs[0, 0], gold)
    # --- Taylor propagator
    dev.set(L.F_PHI, phis)
    dev.apply_exponential(vhs)
    close(dev.get(L.F_PHI)[0], d[tag + 'phi_exp'])
    # --- one-body propagator
    dev.set(L.F_PHI, phis)
    dev.kinetic()
    close(dev.get(L.F_PHI)[0], d[tag + 'phi_kin'])
    # --- local energy
    dev.set(L.F_PHI, phis)
    dev.greens(want_G=True)
    E = dev.local_energy()
    close(E[0], d[tag + 'energy'])
    close(E, numpy.array([model.local_energy(r[2], r[1]) for r in refs]))
    # --- one full propagation step incl. weight update
    if tag + 'step_phi' in d:
        dev.set(L.F_PHI, phis)
        ots = numpy.array([ref.calc_overlap(p, model.psi, na, nb) for p in phis])
        dev.set(L.F_OT, ots)
        dev.set(L.F_HYBRID_ENERGY, numpy.full(nw, 0.25 + 0.1j))
        dev.set(L.F_WEIGHT, numpy.ones(nw))
        dev.propagate(xi, 0.3)
        close(dev.get(L.F_PHI)[0], d[tag + 'step_phi'])
        close(dev.get(L.F_WEIGHT)[0], d[tag + 'step_weight'])
        close(dev.get(L.F_OT)[0], d[tag + 'step_ot'])
        close(dev.get(L.F_HYBRID_ENERGY)[0], d[tag + 'step_ehyb'])
        for i in range(1, nw):
            w = ref.new_walker(model, phis[i])
            w['hybrid_energy'] = 0.25 + 0.1j
            ref.propagate_walker_phaseless(model, w, xi[i], 0.3)
            close(dev.get(L.F_PHI, i, 1)[0], w['phi'])
            close(dev.get(L.F_WEIGHT, i, 1)[0], w['weight'])
            close(dev.get(L.F_HYBRID_ENERGY, i, 1)[0], w['hybrid_energy'])
    # --- re-orthogonalisation
    dev.set(L.F_PHI, phis)
    dev.set(L.F_OT, numpy.ones(nw, dtype=complex))
    detR = dev.reortho()
    close(dev.get(L.F_PHI)[0], d[tag + 'phi_qr'])
    close(detR[0], d[tag + 'detR'])
    close(dev.get(L.F_OT)[0], 1.0 / d[tag + 'detR'])
    dev.close()


def test_mfma_tile_orientation():
    """Identity-like phi, asymmetric one-body matrix: catches a transposed C/D fragment map."""
    M, na, nb = 20, 3, 3
    rng = numpy.random.RandomState(0)
    h1 = rng.rand(M, M)
    chol = rng.rand(M * M, 4) * 0.01
    psi = numpy.linalg.qr(rng.rand(M, M))[0][:, :na + nb].astype(complex)
    B = numpy.zeros((2, M, M), dtype=complex)
    B[0] = numpy.arange(M * M).reshape(M, M) + 1j * numpy.arange(M * M)[::-1].reshape(M, M)
    B[1] = -B[0].T
    rchol = numpy.zeros(((na + nb) * M, 4), dtype=complex)
    m = ref.RefModel('generic', M, na, nb, psi, B, numpy.zeros(4), 0.01, hs_pot=chol, rchol=rchol,
                     H1=numpy.array([h1, h1]), ecore=0.0)
    dev = make_device(m, 2)
    phi = numpy.array([numpy.eye(M)[:, :na + nb] + 0j, numpy.eye(M)[:, ::-1][:, :na + nb] * (1 + 2j)])
    dev.set(L.F_PHI, phi)
    dev.kinetic()
    out = dev.get(L.F_PHI)
    for w in range(2):
        expect = phi[w].copy()
        ref.kinetic_real(expect, B, na)
        close(out[w], expect, 1e-13)
    dev.close()


def test_generic_ops_real_trial(golden):
    d = golden('generic_ops.npz')
    check_ops(d, 'A_', generic_model(d, 'A_'))


def test_generic_ops_complex_trial(golden):
    d = golden('generic_ops.npz')
    check_ops(d, 'B_', generic_model(d, 'B_'))


def test_generic_energy_known_answer(golden):
    """estimators/tests/test_generic.py:62-64 through the HIP energy kernel."""
    d = golden('generic_ops.npz')
    m = generic_model(d, 'A_')
    dev = make_device(m, 1)
    dev.set(L.F_PHI, m.psi[None])
    dev.greens(want_G=True)
    E = dev.local_energy()[0]
    assert E.real == pytest.approx((20.6826247016273, 23.0173528796140, -2.3347281779866), rel=1e-10)
    dev.close()


def test_exchange_energy_algorithms_agree(golden):
    """estimators/generic.py:198-216 by the two device algorithms (afq_set_exchange_algorithm): the T-intermediate
    MFMA kernel and the quadratic form g^T Atil g -- real trial with n_alpha != n_beta, complex trial, odd walker
    counts; both against the oracle and against each other."""
    d = golden('generic_ops.npz')
    for tag, nw in (('A_', 5), ('B_', 19)):
        m = generic_model(d, tag)
        phis = perturbed(d[tag + 'phi'], nw)
        E = {}
        for mode in (1, 2):
            dev = make_device(m, nw)
            dev.set_exchange_algorithm(mode)
            assert dev.exchange_algorithm() == mode
            dev.set(L.F_PHI, phis)
            dev.greens(want_G=False)
            E[mode] = dev.local_energy()
            dev.local_energy()                                   # second evaluation reuses the operands
            dev.close()
        close(E[1], E[2], 1e-12)
        for w in range(nw):
            _, gh, G = ref.greens_function(phis[w], m.psi, m.na, m.nb)
            close(E[2][w], numpy.array(m.local_energy(G, gh)), 1e-10)


def test_hubbard_ops(golden):
    d = golden('hubbard_ops.npz')
    check_ops(d, 'C_', hubbard_model(d, 'C_', 'hubbard'))
    check_ops(d, 'S_', hubbard_model(d, 'S_', 'hubbard_spin'))


def test_hubbard_spin_known_answer(golden):
    d = golden('hubbard_ops.npz')
    m = hubbard_model(d, 'S_', 'hubbard_spin')
    dev = make_device(m, 1)
    dev.set(L.F_PHI, m.psi[None])
    dev.set(L.F_OT, numpy.array([ref.calc_overlap(m.psi, m.psi, m.na, m.nb)]))
    dev.propagate(d['pin_xi'][None], 0.0)
    ot = dev.get(L.F_OT)[0]
    assert ot.real == pytest.approx(0.765551499039435, rel=1e-10)
    assert abs(ot.imag) < 1e-11
    dev.close()


def test_ueg_ops(golden):
    d = golden('ueg_ops.npz')
    m = ueg_model(d, 'U_')
    check_ops(d, 'U_', m)
    dev = make_device(m, 1)
    dev.set(L.F_PHI, d['pw_phi'][None])
    dev.greens(want_G=True)
    fb = dev.force_bias()[0]
    assert numpy.linalg.norm(fb) == pytest.approx(0.16660828645573392, rel=1e-10)
    vhs = dev.vhs((d['pw_xi'] - fb)[None])[0, 0]
    assert numpy.linalg.norm(vhs) == pytest.approx(0.1467322554815581, rel=1e-10)
    dev.close()


def test_full_g_energy(golden):
    """estimators/generic.py:398-434 on the device: the trial's Green's function of
    estimators/tests/test_generic.py:34-64 (pinned 20.6826247016273, 23.0173528796140, -2.3347281779866)
    and random walkers' Green's functions against the oracle; equality with the half-rotated form."""
    from pauxy_amd.estimators.mixed import local_energy as device_local_energy
    d = golden('generic_ops.npz')
    m = generic_model(d, 'A_')
    dev = make_device(m, 3)
    E = dev.local_energy_full_g(d['A_trialG'][None])[0]
    assert E.real == pytest.approx((20.6826247016273, 23.0173528796140, -2.3347281779866), rel=1e-10)
    close(E, d['A_e_full'])
    rng = numpy.random.RandomState(4)
    phis = numpy.array([m.psi + 0.1 * (rng.rand(m.M, m.na + m.nb) + 1j * rng.rand(m.M, m.na + m.nb))
                        for _ in range(3)])
    refs = [ref.greens_function(p, m.psi, m.na, m.nb) for p in phis]
    G = numpy.array([r[2] for r in refs])
    Efull = dev.local_energy_full_g(G)
    close(Efull, numpy.array([ref.local_energy_generic_cholesky(m.H1, m.ecore, g, m.hs_pot) for g in G]))
    close(Efull, numpy.array([m.local_energy(r[2], r[1]) for r in refs]), 1e-9)
    # the reference's free function without Ghalf
    e = device_local_energy(None, G[1], device=dev)
    close(numpy.array(e), Efull[1])
    dev.close()


@pytest.mark.parametrize("charge", [False, True])
def test_hirsch_single_step(golden, charge):
    """propagation/hubbard.py:148-225,285-312 step by step against the oracle: kinetic importance sampling,
    the site loop (chosen fields must match exactly), second kinetic step, eshift factor; one dead walker."""
    from pauxy_amd.device import AfqDevice
    d = golden('traj_hubbard_hirsch.npz')
    na, nb = [int(x) for x in d['nelec']]
    m = ref.HirschModel(d['T'], float(d['U']), d['psi'], na, nb, 0.01, charge)
    nw, M = 6, m.M
    rng = numpy.random.RandomState(9)
    dev = AfqDevice(0)
    dev.set_system_hubbard(m.H1, m.U, na, nb)
    dev.set_trial(m.psi)
    dev.set_propagator_hirsch(m.bt2, 0.01, charge)
    dev.walkers_alloc(nw)
    phis = numpy.array([m.psi + 0.1 * (rng.rand(M, na + nb) + 1j * rng.rand(M, na + nb)) for _ in range(nw)])
    w0 = numpy.array([1.0, 0.7, 0.0, 1.3, 1.0, 0.4])
    dev.set(L.F_PHI, phis)
    dev.set(L.F_WEIGHT, w0)
    ot0 = dev.calc_overlap()
    dev.set(L.F_OT, ot0)
    walkers = [ref.new_walker(m, p, weight=w) for p, w in zip(phis, w0)]
    u = rng.rand(nw, M)
    dev.hirsch_kinetic()
    for wk in walkers:
        if abs(wk['weight']) > 1e-8:
            ref.hirsch_kinetic_importance_sampling(m, wk)
    close(dev.get(L.F_WEIGHT), numpy.array([wk['weight'] for wk in walkers]))
    close(dev.get(L.F_OT), numpy.array([wk['ot'] for wk in walkers]))
    fields, used = dev.hirsch_two_body(u)
    for iw, wk in enumerate(walkers):
        if abs(w0[iw]) > 1e-8 and abs(wk['weight']) > 0:
            it = iter(u[iw])
            f = ref.hirsch_two_body_single_site(m, wk, lambda: next(it))
            assert list(fields[iw]) == f and used[iw] == M
        else:
            assert used[iw] == 0 and numpy.all(fields[iw] == -1)
    close(dev.get(L.F_PHI), numpy.array([wk['phi'] for wk in walkers]))
    close(dev.get(L.F_WEIGHT), numpy.array([wk['weight'] for wk in walkers]))
    close(dev.get(L.F_OT), numpy.array([wk['ot'] for wk in walkers]))
    dev.hirsch_finish(-0.3)
    for iw, wk in enumerate(walkers):
        if abs(w0[iw]) > 1e-8:
            if abs(numpy.real(wk['weight'])) > 0:
                ref.hirsch_kinetic_importance_sampling(m, wk)
            wk['weight'] *= numpy.exp(0.01 * -0.3)
    close(dev.get(L.F_PHI), numpy.array([wk['phi'] for wk in walkers]))
    close(dev.get(L.F_WEIGHT), numpy.array([wk['weight'] for wk in walkers]))
    close(dev.get(L.F_OT), numpy.array([wk['ot'] for wk in walkers]))
    live = numpy.nonzero(dev.get(L.F_WEIGHT) != 0)[0]            # ot is only refreshed while the walker survives
    close(dev.get(L.F_OT)[live], dev.calc_overlap()[live])      # the tracked overlap is the true overlap
    # device-RNG step: same kernels, Philox uniforms
    dev.rng_seed(5, 0)
    before = dev.get(L.F_PHI)
    dev.propagate_hirsch(0.0)
    after, wt = dev.get(L.F_PHI), dev.get(L.F_WEIGHT)
    assert numpy.all(numpy.isfinite(wt)) and wt[2] == 0.0 and numpy.array_equal(after[2], before[2])
    assert numpy.abs(after[0] - before[0]).max() > 1e-6
    live = numpy.nonzero(wt != 0)[0]
    close(dev.get(L.F_OT)[live], dev.calc_overlap()[live])
    dev.close()


def test_greens_function_free_functions():
    """pauxy/estimators/greens_function.py:5-160 by name, on the device, against the oracle's restatements."""
    from pauxy_amd.estimators import greens_function as gf
    rng = numpy.random.RandomState(2)
    M, na, nb = 14, 5, 3
    A = rng.rand(M, na + nb) + 1j * rng.rand(M, na + nb)
    B = rng.rand(M, na + nb) + 1j * rng.rand(M, na + nb)
    close(gf.gab(A[:, :na], B[:, :na]), ref.gab(A[:, :na], B[:, :na]))
    G, Gh = gf.gab_mod(A[:, :na], B[:, :na])
    Gr, Ghr = ref.gab_mod(A[:, :na], B[:, :na])
    close(G, Gr)
    close(Gh, Ghr)
    Gs, Ghs = gf.gab_spin(A, B, na, nb)
    close(Gs[0], Gr)
    close(Gs[1], ref.gab_mod(A[:, na:], B[:, na:])[0])
    close(Ghs[1], ref.gab_mod(A[:, na:], B[:, na:])[1])
    G, Gh, inv = gf.gab_mod_ovlp(A[:, :na], B[:, :na])
    close(inv, numpy.linalg.inv(numpy.dot(B[:, :na].T, A[:, :na].conj())))
    As = numpy.array([A[:, :na], A[:, :na] + 0.1 * rng.rand(M, na), A[:, :na] + 0.2j * rng.rand(M, na)])
    co = numpy.array([0.6 + 0.1j, 0.3, -0.2j])
    Gi = numpy.array([ref.gab(a_, B[:, :na]).T for a_ in As])
    ov = numpy.array([numpy.linalg.det(a_.conj().T.dot(B[:, :na])) for a_ in As])
    close(gf.gab_multi_det(As, B[:, :na], co), numpy.einsum('i,ijk,i->jk', co, Gi, ov) / numpy.dot(co, ov))
    gf.release()
