"""Multi-determinant trial on the device against the golden vectors of the
reference's own propagation tests (propagation/tests/test_generic.py:52-92) and the CPU oracle."""
import numpy
import pytest

from oracle import afqmc_ref as ref
from pauxy_amd import _lib as L
from tests.helpers import make_device, msd_model

pytestmark = pytest.mark.gpu
TOL = 1e-10


def close(a, b, tol=TOL):
    a, b = numpy.asarray(a), numpy.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(numpy.max(numpy.abs(b))))
    err = float(numpy.max(numpy.abs(a - b))) / scale
    assert err <= tol, err


def run_steps(d, tag, hybrid, nw=3, fb_mode=0):
    """Walker 0 replays the reference's run; the other walkers start from perturbed states and are
    checked against the oracle.  fb_mode: afq_set_msd_force_bias (1 one contraction per determinant, 2 through the
    determinant-averaged Green's function, propagation/generic.py:154-157)."""
    m = msd_model(d, tag)
    dev = make_device(m, nw, hybrid=hybrid)
    dev.set_msd_force_bias(fb_mode)
    if fb_mode:
        # (the averaged-G contraction runs on the symmetric-packed hs_pot: non-symmetric L_n keep algorithm 1)
        hs = numpy.asarray(m.hs_pot).reshape(m.M, m.M, -1)
        sym = bool(numpy.array_equal(hs, hs.transpose(1, 0, 2)))
        assert dev.msd_force_bias() == (fb_mode if (fb_mode == 1 or sym) else 1)
    rng = numpy.random.RandomState(3)
    phi0 = numpy.array([d[tag + 'phi0']] + [d[tag + 'phi0'] + 0.05 * (rng.rand(m.M, m.na + m.nb) +
                                                                       1j * rng.rand(m.M, m.na + m.nb))
                                            for _ in range(nw - 1)])
    dev.set(L.F_PHI, phi0)
    # ---- per-operation checks
    ot = dev.greens()
    refs = [m.greens(p) for p in phi0]
    close(ot, numpy.array([r[0] for r in refs]))
    close(ot[0], d[tag + 'ot0'])
    close(dev.det_weights(), numpy.array([r[1] for r in refs]))
    close(dev.det_weights()[0], d[tag + 'weights0'])
    close(dev.calc_overlap(), numpy.array([m.overlap(p) for p in phi0]))
    xbar = dev.force_bias()
    close(xbar, numpy.array([m.force_bias(r[1], r[2]) for r in refs]))
    close(xbar[0], d[tag + 'xbar0'])
    dev.greens()
    E = dev.local_energy()
    close(E, numpy.array([m.local_energy(r[2], r[1]) for r in refs]))
    close(E[0], d[tag + 'energy0'])
    # ---- the reference's 10 propagation steps
    dev.set(L.F_OT, ot)
    walkers = [ref.new_walker(m, p) for p in phi0]
    eshift = complex(d[tag + 'eshift'])
    xi_rec = d[tag + 'xi']
    for i in range(xi_rec.shape[0]):
        xi = numpy.array([xi_rec[i]] + [rng.normal(size=m.nfields) for _ in range(nw - 1)])
        dev.propagate(xi, eshift)
        for w, x in zip(walkers, xi):
            ref.propagate_walker_phaseless(m, w, x, eshift, hybrid=hybrid)
        phi, wt, o = dev.get(L.F_PHI), dev.get(L.F_WEIGHT), dev.get(L.F_OT)
        close(phi, numpy.array([w['phi'] for w in walkers]), 1e-9)
        close(wt, numpy.array([w['weight'] for w in walkers]), 1e-9)
        close(o, numpy.array([w['ot'] for w in walkers]), 1e-9)
        close(phi[0], d[tag + 'step_phi'][i], 1e-9)
        close(wt[0], d[tag + 'step_weight'][i], 1e-9)
        close(o[0], d[tag + 'step_ot'][i], 1e-9)
        if hybrid:
            close(dev.get(L.F_HYBRID_ENERGY)[0], d[tag + 'step_ehyb'][i], 1e-9)
        else:
            close(dev.get(L.F_ELOC)[0], d[tag + 'step_eloc'][i], 1e-9)
    wfinal = float(dev.get(L.F_WEIGHT)[0])
    detR = dev.reortho()
    close(dev.get(L.F_PHI)[0], d[tag + 'phi_qr'], 1e-9)
    close(detR[0], d[tag + 'detR'], 1e-9)
    dev.close()
    return wfinal


@pytest.mark.parametrize("fb_mode", [0, 1, 2])
def test_phmsd_local_energy_weights(golden, fb_mode):
    w = run_steps(golden('msd_ops.npz'), 'PL_', False, fb_mode=fb_mode)
    assert w == pytest.approx(0.68797524675701, rel=1e-9)           # propagation/tests/test_generic.py:72


@pytest.mark.parametrize("fb_mode", [0, 1, 2])
def test_phmsd_hybrid(golden, fb_mode):
    w = run_steps(golden('msd_ops.npz'), 'PH_', True, fb_mode=fb_mode)
    assert w == pytest.approx(0.7430443466368197, rel=1e-9)         # propagation/tests/test_generic.py:92


@pytest.mark.parametrize("fb_mode", [0, 1, 2])
def test_nomsd(golden, fb_mode):
    run_steps(golden('msd_ops.npz'), 'N_', True, fb_mode=fb_mode)


def test_nomsd_many_walkers_averaged_g(golden):
    """40 walkers: the automatic choice (more than 32 walkers, cost model) and both forced algorithms agree."""
    run_steps(golden('msd_ops.npz'), 'N_', True, nw=40, fb_mode=0)
    run_steps(golden('msd_ops.npz'), 'N_', True, nw=40, fb_mode=2)


@pytest.mark.parametrize("fb_mode,na,nb,M", [(1, 3, 3, 12), (2, 3, 3, 12), (1, 50, 50, 110), (2, 50, 50, 110)])
def test_numerically_dead_determinant_is_skipped_like_the_reference(fb_mode, na, nb, M):
    """walkers/multi_det.py:209,218: a determinant whose overlap with the walker is below 1e-16 is skipped by the
    reference's Green's function -- no Gi, no weight, not in the total.  Here: a NOMSD trial of three determinants, one of
    them EXACTLY orthogonal to some of the walkers (an orbital on a basis function those walkers do not occupy: its
    overlap matrix has a zero row, the inverse does not exist).  Determinant weights, total overlap, force bias (both
    algorithms) and energy against the oracle, which restates the skip (a fresh walker: zero weight, zero Gi); then one
    full step.  calc_overlap (multi_det.py:135-162) skips nothing -- the zero overlap just adds zero."""
    from pauxy_amd import systems, trial as trial_mod
    from pauxy_amd.propagation import setup
    K, dt, nw = 9, 0.005, 5
    nt = na + nb
    s = systems.synthetic_generic(M, K, (na, nb), seed=7)
    t0 = trial_mod.rhf_trial_generic(s)
    rng = numpy.random.RandomState(5)
    dets = numpy.array([t0.psi + 0.05 * (rng.rand(M, nt) + 1j * rng.rand(M, nt)) for _ in range(3)])
    dets[:, M - 1, :] = 0.0                                 # nobody occupies the last basis function ...
    dets[1][:, 1] = 0.0
    dets[1][M - 1, 1] = 1.0                                 # ... except one alpha orbital of determinant 1
    coeffs = numpy.array([0.7 + 0.1j, 0.4 - 0.2j, 0.3 + 0.05j])
    t = trial_mod.MultiDetTrial(s, (coeffs, dets), init=t0.psi)
    BH1, mf = setup.generic_propagator_arrays(s, t, dt)
    model = ref.RefModel('generic_msd', M, na, nb, dets, BH1, mf, dt, coeffs=coeffs, hs_pot=s.hs_pot,
                         H1=numpy.array([s.H1[0], s.H1[1]]).astype(complex), ecore=s.ecore)
    phis = t0.psi[None] + 0.05 * (rng.rand(nw, M, nt) + 1j * rng.rand(nw, M, nt))
    phis[:3, M - 1, :] = 0.0                                # walkers 0..2: <D_1|phi> = 0 exactly; 4: ordinary
    # walker 3: the ALPHA determinant with D_1 alone is below the threshold (1e-17 times a minor of order one) while the
    # product of both spins is far above it (beta columns scaled up): the reference tests the alpha factor first (:209)
    phis[3, M - 1, :na] = 0.0
    phis[3, M - 1, 0] = 1e-17
    phis[3, :, na:] *= 1e3
    dev = make_device(model, nw)
    dev.set_msd_force_bias(fb_mode)
    dev.set(L.F_PHI, phis)
    tot = dev.greens()
    wts = dev.det_weights()
    refs = [model.greens(p) for p in phis]
    for w in range(4):
        assert refs[w][1][1] == 0.0 and wts[w][1] == 0.0   # skipped by the oracle (as by the reference) and by the device
    assert abs(refs[4][1][1]) > 0.0
    da = numpy.linalg.det(phis[3][:, :na].T @ dets[1][:, :na].conj())
    db = numpy.linalg.det(phis[3][:, na:].T @ dets[1][:, na:].conj())
    assert abs(da) < 1e-16 and abs(da * db) > 1e-12        # the alpha test, not the product test, is what skips walker 3
    close(wts, numpy.array([r[1] for r in refs]))
    close(tot, numpy.array([r[0] for r in refs]))
    xbar = dev.force_bias()
    assert numpy.all(numpy.isfinite(xbar))
    close(xbar, numpy.array([model.force_bias(r[1], r[2]) for r in refs]))
    # (calc_overlap refreshes the determinant weights WITHOUT the skip, as multi_det.py:160 does: walker 3's weight of
    #  determinant 1 comes back -- hence after the force bias, and a fresh Green's function before the energy)
    close(dev.calc_overlap(), numpy.array([model.overlap(p) for p in phis]))
    dev.greens()
    E = dev.local_energy()
    assert numpy.all(numpy.isfinite(E))
    close(E, numpy.array([model.local_energy(r[2], r[1]) for r in refs]), 1e-9)
    # one full step; the propagated walkers occupy every basis function again
    dev.set(L.F_OT, tot)
    walkers = [ref.new_walker(model, p) for p in phis]
    xi = rng.normal(size=(nw, K))
    dev.propagate(xi, -0.3)
    for w, x in zip(walkers, xi):
        ref.propagate_walker_phaseless(model, w, x, -0.3)
    close(dev.get(L.F_PHI), numpy.array([w['phi'] for w in walkers]), 1e-9)
    close(dev.get(L.F_WEIGHT), numpy.array([w['weight'] for w in walkers]), 1e-9)
    close(dev.get(L.F_OT), numpy.array([w['ot'] for w in walkers]), 1e-9)
    dev.close()
