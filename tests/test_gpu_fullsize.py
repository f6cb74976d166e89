"""BASELINE-size parity on the GPU.

At the full sizes of BASELINE configs[1..3] the oracle still finishes one walker
in well under a second, so a sample of walkers of each full-size batch is checked
DIRECTLY against the oracle (one complete propagation step + local energy), and
every walker is checked through size-independent properties:
  * the Green's function is a rank-N projector: G.G = G, trace G_s = N_s
  * re-orthogonalisation: columns orthonormal, ot * detR invariant
  * zero HS potential leaves phi unchanged by the Taylor propagator
  * dead walkers are not touched
"""
import numpy
import pytest

from oracle import afqmc_ref as ref
from pauxy_amd import _lib as L
from pauxy_amd import systems, trial as trial_mod
from pauxy_amd.propagation import setup
from tests.helpers import make_device

pytestmark = pytest.mark.gpu


def close(a, b, tol):
    a, b = numpy.asarray(a), numpy.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(numpy.max(numpy.abs(b))))
    err = float(numpy.max(numpy.abs(a - b))) / scale
    assert err <= tol, err


def generic_c3(nw):
    M, K, N, dt = 100, 500, 25, 0.005
    s = systems.synthetic_generic(M, K, (N, N), seed=7)
    t = trial_mod.rhf_trial_generic(s)
    BH1, mf = setup.generic_propagator_arrays(s, t, dt)
    model = ref.RefModel('generic', M, N, N, t.psi, BH1, mf, dt, hs_pot=s.hs_pot, rchol=t._rchol,
                         H1=s.H1.astype(complex), ecore=0.0)
    return model


def hubbard_c4():
    s = systems.Hubbard(16, 16, 128, 128, 8.0)
    t = trial_mod.uhf_trial_hubbard(s, ueff=0.4)
    dt = 0.005
    BH1, mf = setup.hubbard_propagator_arrays(s, t, dt, True)
    return ref.RefModel('hubbard', 256, 128, 128, t.psi, BH1, mf, dt, U=8.0, H1=s.T.astype(complex))


def ueg_c2():
    s = systems.UEG(2.0, 7, 7, 4.0)
    assert s.nbasis == 93 and s.nchol == 750
    t = trial_mod.hartree_fock_ueg(s)
    dt = 0.005
    BH1, mf = setup.ueg_propagator_arrays(s, t, dt)
    H1diag = numpy.array([numpy.diag(s.H1[0]), numpy.diag(s.H1[1])])
    return ref.RefModel('ueg', s.nbasis, 7, 7, t.psi, BH1, mf, dt, iA=s.iA, iB=s.iB, H1diag=H1diag,
                        vqvec=s.vqvec, vol=s.vol, ikpq_i=s.ikpq_i, ikpq_kpq=s.ikpq_kpq, ipmq_i=s.ipmq_i,
                        ipmq_pmq=s.ipmq_pmq, ecore=s.ecore)


def run_fullsize(model, nw, sample, tol_op=1e-10, tol_step=1e-9):
    rng = numpy.random.RandomState(5)
    M, na, nb, K = model.M, model.na, model.nb, model.nfields
    dev = make_device(model, nw)
    phis = model.psi[None] + 0.05 * (rng.rand(nw, M, na + nb) + 1j * rng.rand(nw, M, na + nb))
    dev.set(L.F_PHI, phis)
    # ---- projector property of the Green's function, all walkers
    det = dev.greens(want_G=True)
    G = dev.get(L.F_G)
    for w in range(0, nw, max(1, nw // 16)):
        for s_, n in ((0, na), (1, nb)):
            close(G[w, s_] @ G[w, s_], G[w, s_], 1e-9)
            assert abs(numpy.trace(G[w, s_]) - n) < 1e-8 * n
    E = dev.local_energy()
    # ---- sampled walkers against the oracle: Green's function, energy
    for w in sample:
        d, gh, Gr = ref.greens_function(phis[w], model.psi, na, nb)
        close(det[w], d, tol_op)
        close(G[w], Gr, tol_op)
        close(E[w], numpy.array(model.local_energy(Gr, gh)), tol_op)
    # ---- one full step, some walkers dead
    xi = rng.normal(size=(nw, K))
    w0 = numpy.ones(nw)
    w0[1::5] = 0.0
    dev.set(L.F_WEIGHT, w0)
    dev.set(L.F_OT, det)
    dev.propagate(xi, 0.0)
    out_phi, out_w, out_e, out_ot = (dev.get(L.F_PHI), dev.get(L.F_WEIGHT), dev.get(L.F_HYBRID_ENERGY),
                                     dev.get(L.F_OT))
    dead = numpy.nonzero(w0 == 0.0)[0]
    assert numpy.array_equal(out_phi[dead], phis[dead]) and numpy.all(out_w[dead] == 0.0)
    for w in sample:
        if w0[w] == 0.0:
            continue
        wk = ref.new_walker(model, phis[w])
        ref.propagate_walker_phaseless(model, wk, xi[w], 0.0)
        close(out_phi[w], wk['phi'], tol_step)
        close(out_w[w], wk['weight'], tol_step)
        close(out_e[w], wk['hybrid_energy'], tol_step)
        close(out_ot[w], wk['ot'], tol_step)
    # ---- zero HS potential: Taylor propagator is the identity
    dev.set(L.F_PHI, phis)
    dev.set(L.F_WEIGHT, numpy.ones(nw))
    dev.apply_exponential(numpy.zeros((nw, dev.nv, M, M), dtype=complex))
    assert numpy.array_equal(dev.get(L.F_PHI), phis)
    # ---- re-orthogonalisation invariants, all walkers
    ot0 = dev.calc_overlap()
    dev.set(L.F_OT, ot0)
    detR = dev.reortho()
    q = dev.get(L.F_PHI)
    ot1 = dev.calc_overlap()
    close(ot1 * detR, ot0, 1e-9)                      # detR * ot == ovlp (walkers/tests/test_single_det.py)
    close(dev.get(L.F_OT), ot0 / detR, 1e-12)
    for w in range(0, nw, max(1, nw // 8)):
        for sl in (slice(0, na), slice(na, na + nb)):
            close(q[w][:, sl].conj().T @ q[w][:, sl], numpy.eye(sl.stop - sl.start), 1e-11)
    for w in sample[:2]:
        p = phis[w].copy()
        d = ref.reortho(p, na, nb)
        close(q[w], p, 1e-9)
        close(detR[w], d, 1e-9)
    dev.close()


def test_c3_generic_256_walkers():
    run_fullsize(generic_c3(256), 256, [0, 100, 255])


def position_independence(model, n0, rep, tol=1e-11):
    """rep copies of n0 distinct walkers with the same fields: Green's function, force bias, one full step (hybrid weights,
    some walkers dead), energies and the re-orthogonalisation bit-equal between the copies, and the first copy equal to an
    n0-walker run of the same walkers to rounding (contractions are split differently for the two populations)."""
    rng = numpy.random.RandomState(21)
    M, nt, K = model.M, model.na + model.nb, model.nfields
    phis0 = model.psi[None] + 0.05 * (rng.rand(n0, M, nt) + 1j * rng.rand(n0, M, nt))
    xi0 = rng.normal(size=(n0, K))
    w0 = numpy.ones(n0)
    w0[3::7] = 0.0
    out = {}
    for nw in (n0 * rep, n0):
        r = nw // n0
        dev = make_device(model, nw)
        dev.set(L.F_PHI, numpy.tile(phis0, (r, 1, 1)))
        det = dev.greens(want_G=(model.kind == 'ueg'))
        xbar = dev.force_bias()
        dev.set(L.F_WEIGHT, numpy.tile(w0, r))
        dev.set(L.F_OT, det)
        dev.propagate(numpy.tile(xi0, (r, 1)), -0.3)
        if model.kind == 'ueg':
            dev.greens(want_G=True)
        E = dev.local_energy()
        res = dict(det=det, xbar=xbar, phi=dev.get(L.F_PHI), weight=dev.get(L.F_WEIGHT), ot=dev.get(L.F_OT),
                   ehyb=dev.get(L.F_HYBRID_ENERGY), E=E)
        res['detR'] = dev.reortho()
        res['q'] = dev.get(L.F_PHI)
        dev.close()
        out[nw] = res
    big, small = out[n0 * rep], out[n0]
    for key, v in big.items():
        v = numpy.asarray(v)
        assert numpy.all(numpy.isfinite(v.view(float))), key
        for c in range(1, rep):
            assert numpy.array_equal(v[c * n0:(c + 1) * n0], v[:n0]), (key, c)
        close(v[:n0], small[key], tol)


def test_c3_large_population_is_position_independent():
    """2048 walkers at the BASELINE configs[2] size (the single-GPU leg of the strong-scaling bench) made of 8 copies of 256
    distinct walkers: what a walker gets must not depend on where in the population it sits."""
    position_independence(generic_c3(256), 256, 8)


@pytest.mark.parametrize("which", ["C2-ueg", "C4-hubbard16x16", "C5-single-determinant"])
def test_other_configs_are_position_independent(which):
    """The same for the other BASELINE sizes: 1024 walkers of the electron gas, 512 of the 16 x 16 lattice, 256 at M = 400."""
    if which == "C2-ueg":
        position_independence(ueg_c2(), 256, 4)
    elif which == "C4-hubbard16x16":
        position_independence(hubbard_c4(), 256, 2, 1e-10)
    else:
        position_independence(generic_c5(), 128, 2, 1e-10)


def test_c3_exchange_algorithms_agree():
    """BASELINE configs[2] size, 256 walkers: quadratic-form exchange energy (the default there, K = 5 M) against the
    T-intermediate kernel for every walker, and a sample against the oracle."""
    model = generic_c3(256)
    rng = numpy.random.RandomState(9)
    nw, M, nt = 256, model.M, model.na + model.nb
    phis = model.psi[None] + 0.05 * (rng.rand(nw, M, nt) + 1j * rng.rand(nw, M, nt))
    E = {}
    for mode in (1, 2):
        dev = make_device(model, nw)
        dev.set_exchange_algorithm(mode)
        dev.set(L.F_PHI, phis)
        dev.greens()
        E[mode] = dev.local_energy()
        if mode == 2:
            assert dev.exchange_algorithm() == 2
        dev.close()
    close(E[1], E[2], 1e-12)
    dev = make_device(model, nw)
    assert dev.exchange_algorithm() == 2                            # what bench.py runs
    dev.close()
    for w in (0, 17, 255):
        _, gh, Gr = ref.greens_function(phis[w], model.psi, model.na, model.nb)
        close(E[2][w], numpy.array(model.local_energy(Gr, gh)), 1e-10)


def test_c4_hubbard_16x16():
    run_fullsize(hubbard_c4(), 32, [0, 31])


def test_c4_hubbard_16x16_256_walkers():
    """BASELINE configs[3] at its stated population per GPU (2048 walkers over 8 GPUs): batch <-> XCD maps, 64-walker
    GEMM tiles and the split-K slice counts all branch on the walker count."""
    run_fullsize(hubbard_c4(), 256, [0, 129, 255])


def test_hubbard_large_n_ragged():
    """45 < N <= 128 with na != nb and no dimension a multiple of 16: the GEMM + register-resident
    Gauss-Jordan Green's function (k_bigdet.hip), every walker against the oracle."""
    s = systems.Hubbard(9, 8, 50, 47, 4.0)
    t = trial_mod.uhf_trial_hubbard(s, ueff=0.4)
    dt = 0.01
    BH1, mf = setup.hubbard_propagator_arrays(s, t, dt, True)
    model = ref.RefModel('hubbard', 72, 50, 47, t.psi, BH1, mf, dt, U=4.0, H1=s.T.astype(complex))
    run_fullsize(model, 11, list(range(11)))


def test_large_n_reortho_breakdown_fallback():
    """Cholesky-QR2 breaks down on a numerically rank-deficient walker; the flagged walker must be
    redone by the Gram-Schmidt kernel and the healthy walkers must be unaffected."""
    s = systems.Hubbard(9, 8, 50, 47, 4.0)
    t = trial_mod.uhf_trial_hubbard(s, ueff=0.4)
    BH1, mf = setup.hubbard_propagator_arrays(s, t, 0.01, True)
    model = ref.RefModel('hubbard', 72, 50, 47, t.psi, BH1, mf, 0.01, U=4.0, H1=s.T.astype(complex))
    rng = numpy.random.RandomState(11)
    nw = 5
    dev = make_device(model, nw)
    phis = model.psi[None] + 0.05 * (rng.rand(nw, 72, 97) + 1j * rng.rand(nw, 72, 97))
    phis[2][:, 7] = phis[2][:, 3] * (1.0 + 1e-9) + 1e-10 * rng.rand(72)     # cond ~ 1e9
    dev.set(L.F_PHI, phis)
    dev.set(L.F_OT, numpy.ones(nw, dtype=complex))
    detR = dev.reortho()
    q = dev.get(L.F_PHI)
    assert numpy.all(numpy.isfinite(detR)) and numpy.all(numpy.isfinite(q.view(float)))
    for w in range(nw):
        for sl in (slice(0, 50), slice(50, 97)):
            close(q[w][:, sl].conj().T @ q[w][:, sl], numpy.eye(sl.stop - sl.start), 1e-10)
        if w != 2:
            p = phis[w].copy()
            d = ref.reortho(p, 50, 47)
            close(q[w], p, 1e-9)
            close(detR[w], d, 1e-9)
    # span preserved for the repaired walker: Q Q^H phi = phi
    qa = q[2][:, :50]
    close(qa @ (qa.conj().T @ phis[2][:, :50]), phis[2][:, :50], 1e-9)
    dev.close()


def test_c2_ueg_93_planewaves():
    run_fullsize(ueg_c2(), 64, [0, 63])


def test_c2_ueg_256_walkers():
    """BASELINE configs[1] at its stated 256 walkers."""
    run_fullsize(ueg_c2(), 256, [0, 100, 255])


def generic_c5():
    M, K, N, dt = 400, 2000, 50, 0.005
    s = systems.synthetic_generic(M, K, (N, N), seed=7)
    t = trial_mod.rhf_trial_generic(s)
    BH1, mf = setup.generic_propagator_arrays(s, t, dt)
    return ref.RefModel('generic', M, N, N, t.psi, BH1, mf, dt, hs_pot=s.hs_pot, rchol=t._rchol,
                        H1=s.H1.astype(complex), ecore=0.0)


def test_c5_sizes_single_determinant_256_walkers():
    """BASELINE configs[4] sizes (M=400, K=2000, 50+50) with a single determinant at 256 walkers per GPU: three walkers
    against the oracle (Green's function, energy, one full step, re-orthogonalisation), properties on all."""
    run_fullsize(generic_c5(), 256, [0, 130, 255])


def test_c5_sizes_closed_shell_walkers_take_the_alpha_columns_through_the_gemm_chain():
    """The same sizes with walkers whose spin blocks are bitwise equal (what the RHF run of `bench.py --config C5sd` holds): the
    GEMM chain multiplies their alpha columns only and copies them over beta (afq_counters_ext [7]), the exchange energy evaluates
    one spin ([4]).  Three walkers against the oracle for the energy and for one full step (propagation/continuous.py:232-262),
    dead walkers untouched, live ones closed bit for bit afterwards."""
    model = generic_c5()
    nw, N, M, K = 256, model.na, model.M, model.nfields
    assert numpy.array_equal(model.psi[:, :N], model.psi[:, N:]) and numpy.array_equal(model.BH1[0], model.BH1[1])
    rng = numpy.random.RandomState(8)
    half = model.psi[None, :, :N] + 0.05 * (rng.rand(nw, M, N) + 1j * rng.rand(nw, M, N))
    phis = numpy.concatenate([half, half], axis=2)
    dev = make_device(model, nw)
    dev.set(L.F_PHI, phis)
    det = dev.greens(want_G=False)
    dev.counters(reset=True, n=8)
    E = dev.local_energy()
    assert int(dev.counters(n=8)[4]) == (nw if dev.exchange_algorithm() == 2 else 0)
    sample = [0, 130, 255]
    for w in sample:
        d, gh, Gr = ref.greens_function(phis[w], model.psi, N, N)
        close(det[w], d, 1e-10)
        close(E[w], numpy.array(model.local_energy(Gr, gh)), 1e-10)
    xi = rng.normal(size=(nw, K))
    w0 = numpy.ones(nw)
    w0[1::5] = 0.0
    dev.set(L.F_WEIGHT, w0)
    dev.set(L.F_OT, det)
    dev.counters(reset=True, n=8)
    dev.propagate(xi, 0.0)
    assert int(dev.counters(n=8)[7]) == int((w0 > 0).sum())
    out_phi, out_w = dev.get(L.F_PHI), dev.get(L.F_WEIGHT)
    dead = numpy.nonzero(w0 == 0.0)[0]
    assert numpy.array_equal(out_phi[dead], phis[dead]) and numpy.all(out_w[dead] == 0.0)
    assert numpy.array_equal(out_phi[:, :, :N], out_phi[:, :, N:])
    for w in sample:
        wk = ref.new_walker(model, phis[w])
        ref.propagate_walker_phaseless(model, wk, xi[w], 0.0)
        close(out_phi[w], wk['phi'], 1e-9)
        close(out_w[w], wk['weight'], 1e-9)
    dev.close()


@pytest.mark.parametrize("M,K,N,nw", [(400, 2000, 50, 32), (130, 24, 20, 33), (100, 20, 45, 9), (40, 16, 33, 65)],
                         ids=["C5", "M130-20e", "M100-45e", "M40-33e"])
def test_c5_sizes_multi_determinant_consistency(M, K, N, nw):
    """BASELINE configs[4] sizes (M=400, K=2000, 50+50 electrons), and smaller shapes that take the other dispatch paths
    of the determinant kernels with several determinants (the GEMM + register Gauss-Jordan path below 45 electrons, the
    LDS Gauss-Jordan of the one-work-group kernel).  Size-independent property of the
    multi-determinant path: a trial made of three IDENTICAL determinants with coefficients c_d must
    reproduce the single-determinant force bias, energy and propagated walkers, with overlap
    sum_d conj(c_d) times the single-determinant overlap.  One walker is also checked directly against
    the oracle (Green's function, force bias, half-rotated energy)."""
    dt = 0.005
    s = systems.synthetic_generic(M, K, (N, N), seed=7)
    t = trial_mod.rhf_trial_generic(s)
    BH1, mf = setup.generic_propagator_arrays(s, t, dt)
    model = ref.RefModel('generic', M, N, N, t.psi, BH1, mf, dt, hs_pot=s.hs_pot, rchol=t._rchol,
                         H1=s.H1.astype(complex), ecore=0.0)
    rng = numpy.random.RandomState(5)
    phis = model.psi[None] + 0.05 * (rng.rand(nw, M, 2 * N) + 1j * rng.rand(nw, M, 2 * N))
    xi = rng.normal(size=(nw, K))
    # ---- single determinant
    dev = make_device(model, nw)
    dev.set(L.F_PHI, phis)
    det = dev.greens()
    xbar = dev.force_bias()
    E = dev.local_energy()
    d, gh, Gr = ref.greens_function(phis[3], model.psi, N, N)
    close(det[3], d, 1e-9)
    close(xbar[3], model.force_bias(gh, Gr), 1e-9)
    close(E[3], numpy.array(model.local_energy(Gr, gh)), 1e-9)
    dev.set(L.F_OT, det)
    dev.propagate(xi, 0.0)
    phi_sd, w_sd = dev.get(L.F_PHI), dev.get(L.F_WEIGHT)
    dev.close()
    # ---- three copies of the same determinant
    coeffs = numpy.array([0.7 + 0.1j, 0.2 - 0.3j, 0.4 + 0j])
    per = M * 2 * N
    dev = AfqDeviceForMsd(model, coeffs, t._rchol[:per], nw)
    dev.set(L.F_PHI, phis)
    det3 = dev.greens()
    close(det3, det * numpy.sum(coeffs.conj()), 1e-10)
    close(dev.det_weights(), det[:, None] * coeffs.conj()[None, :], 1e-10)
    close(dev.force_bias(), xbar, 1e-10)
    dev.greens()
    close(dev.local_energy(), E, 1e-10)
    dev.set(L.F_OT, det3)
    dev.propagate(xi, 0.0)
    close(dev.get(L.F_PHI), phi_sd, 1e-10)
    close(dev.get(L.F_WEIGHT), w_sd, 1e-10)
    dev.close()


def AfqDeviceForMsd(model, coeffs, rchol0, nw):
    from pauxy_amd.device import AfqDevice
    dev = AfqDevice(0)
    ndet = len(coeffs)
    dev.set_system_generic(model.hs_pot, rchol0, model.H1, model.ecore, model.na, model.nb)
    dev.set_trial_multi(numpy.array([model.psi] * ndet), coeffs, numpy.concatenate([rchol0] * ndet))
    dev.set_propagator(model.BH1, model.mf_shift, model.dt)
    dev.walkers_alloc(nw)
    return dev


@pytest.mark.parametrize("nw", [8, 256])
def test_c5_sizes_distinct_complex_determinants(nw):
    """BASELINE configs[4] sizes (M=400, K=2000, 50+50) with a NOMSD trial of three DISTINCT complex determinants
    (complex half-rotated Cholesky vectors: the complex exchange path).  Per walker: determinant weights
    conj(c_d) <D_d|phi>, the weighted force bias (propagation/generic.py:154-157) and the energy
    sum_d w_d E[G_d] / sum_d w_d (estimators/mixed.py:439-448) with every E[G_d] evaluated by the oracle's
    half-rotated form (estimators/generic.py:156-221, equal to the reference's full-G form; the full-G intermediate
    would need 5 GB per walker here); both exchange algorithms of the device against each other for all walkers."""
    M, K, N, dt = 400, 2000, 50, 0.005
    s = systems.synthetic_generic(M, K, (N, N), seed=7)
    t0 = trial_mod.rhf_trial_generic(s)
    rng = numpy.random.RandomState(3)
    dets = numpy.array([t0.psi + (0.0 if d == 0 else 0.05) * (rng.rand(M, 2 * N) + 1j * rng.rand(M, 2 * N)) for d in range(3)])
    coeffs = numpy.array([0.8 + 0.1j, 0.3 - 0.2j, 0.2 + 0.05j])
    t = trial_mod.MultiDetTrial(s, (coeffs, dets), init=t0.psi)
    BH1, mf = setup.generic_propagator_arrays(s, t, dt)
    phis = t0.psi[None] + 0.05 * (rng.rand(nw, M, 2 * N) + 1j * rng.rand(nw, M, 2 * N))
    per = M * 2 * N
    E = {}
    for mode in (1, 2):
        from pauxy_amd.device import AfqDevice
        dev = AfqDevice(0)
        dev.set_exchange_algorithm(mode)
        dev.set_system_generic(s.hs_pot, t._rchol[:per], s.H1.astype(complex), s.ecore, N, N)
        dev.set_trial_multi(dets, coeffs, t._rchol)
        dev.set_propagator(BH1, mf, dt)
        dev.walkers_alloc(nw)
        dev.set(L.F_PHI, phis)
        tot = dev.greens()
        wts = dev.det_weights()
        xbar = dev.force_bias()
        # (256 walkers: the automatic choice is the contraction with the determinant-averaged G, propagation/generic.py:154-157;
        #  8 walkers: one contraction per determinant) -- both algorithms against each other
        assert dev.msd_force_bias() == (2 if nw > 32 else 1)
        xb = {}
        for fbm in (1, 2):
            dev.set_msd_force_bias(fbm)
            xb[fbm] = dev.force_bias()
        close(xb[1], xb[2], 1e-11)
        dev.set_msd_force_bias(0)
        dev.greens()
        E[mode] = dev.local_energy()
        assert dev.exchange_algorithm() == mode
        dev.close()
    close(E[1], E[2], 1e-11)
    H1 = s.H1.astype(complex)
    for w in (0, nw - 3):
        ws, Es, Gsum = [], [], 0.0
        for d in range(3):
            ov, gh, G = ref.greens_function(phis[w], dets[d], N, N)
            ws.append(numpy.conj(coeffs[d]) * ov)
            Es.append(numpy.array(ref.local_energy_generic_cholesky_opt(H1, s.ecore, G, gh, t._rchol[d * per:(d + 1) * per], N, N)))
            Gsum = Gsum + ws[-1] * (G[0] + G[1])
        ws = numpy.array(ws)
        close(wts[w], ws, 1e-10)
        close(tot[w], ws.sum(), 1e-10)
        close(E[2][w], (ws[:, None] * numpy.array(Es)).sum(0) / ws.sum(), 1e-10)
        vbias = s.hs_pot.T.dot((Gsum / ws.sum()).ravel())                      # propagation/generic.py:154-157
        close(xbar[w], -dt ** 0.5 * (1j * vbias - mf), 1e-10)


def test_c5_sizes_back_propagation_window():
    """A back-propagation window at M=400, K=2000 (single-determinant trial, as the reference requires for it):
    three recorded steps through the unfused propagator path (M > 104), afq_bp_update against the oracle."""
    M, K, N, dt, nw, nbp = 400, 2000, 50, 0.005, 4, 3
    s = systems.synthetic_generic(M, K, (N, N), seed=7)
    t = trial_mod.rhf_trial_generic(s)
    BH1, mf = setup.generic_propagator_arrays(s, t, dt)
    model = ref.RefModel('generic', M, N, N, t.psi, BH1, mf, dt, hs_pot=s.hs_pot, rchol=t._rchol,
                         H1=s.H1.astype(complex), ecore=0.0)
    rng = numpy.random.RandomState(11)
    phis = model.psi[None] + 0.05 * (rng.rand(nw, M, 2 * N) + 1j * rng.rand(nw, M, 2 * N))
    dev = make_device(model, nw)
    dev.set(L.F_PHI, phis)
    dev.set(L.F_OT, dev.calc_overlap())
    dev.bp_configure(nbp)
    walkers = [ref.new_walker(model, p) for p in phis]
    for w in walkers:
        w['bp'] = ref.bp_new(K, nbp)
        w['phi_old'] = w['phi'].copy()
    for step in range(nbp):
        xi = rng.normal(size=(nw, K))
        dev.propagate(xi, 0.1)
        for w, x in zip(walkers, xi):
            ref.propagate_walker_phaseless(model, w, x, 0.1)
    close(dev.get(L.F_WEIGHT), numpy.array([w['weight'] for w in walkers]), 1e-9)
    est = numpy.zeros(4 + 2 * M * M, dtype=complex)
    ref.bp_update(model, walkers, 2, est, 'partial')
    _, denom, G = dev.bp_update(model.psi, 2, 'partial')
    close(denom, est[3], 1e-9)
    close(G, est[4:].reshape(2, M, M), 1e-8)
    dev.close()


def run_hirsch_lattice(nx, ny, na, nb, charge=False, nw=4):
    """Kinetic importance sampling, site loop (chosen fields exact), against the oracle on one lattice."""
    from pauxy_amd.device import AfqDevice
    s = systems.Hubbard(nx, ny, na, nb, 4.0)
    t = trial_mod.uhf_trial_hubbard(s, ueff=0.4)
    m = ref.HirschModel(s.T.astype(complex), 4.0, t.psi, na, nb, 0.01, charge)
    M = m.M
    rng = numpy.random.RandomState(9)
    dev = AfqDevice(0)
    dev.set_system_hubbard(m.H1, m.U, na, nb)
    dev.set_trial(m.psi)
    dev.set_propagator_hirsch(m.bt2, 0.01, charge)
    dev.walkers_alloc(nw)
    phis = numpy.array([m.psi + 0.02 * (rng.rand(M, na + nb) + 1j * rng.rand(M, na + nb)) for _ in range(nw)])
    w0 = numpy.resize(numpy.array([1.0, 0.7, 0.0, 1.3]), nw)
    dev.set(L.F_PHI, phis)
    dev.set(L.F_WEIGHT, w0)
    dev.set(L.F_OT, dev.calc_overlap())
    walkers = [ref.new_walker(m, p, weight=w) for p, w in zip(phis, w0)]
    u = rng.rand(nw, M)
    dev.hirsch_kinetic()
    for wk in walkers:
        if abs(wk['weight']) > 1e-8:
            ref.hirsch_kinetic_importance_sampling(m, wk)
    close(dev.get(L.F_WEIGHT), numpy.array([wk['weight'] for wk in walkers]), 1e-9)
    fields, used = dev.hirsch_two_body(u)
    for iw, wk in enumerate(walkers):
        if abs(w0[iw]) > 1e-8 and abs(wk['weight']) > 0:
            it = iter(u[iw])
            f = ref.hirsch_two_body_single_site(m, wk, lambda: next(it))
            assert list(fields[iw]) == f and used[iw] == M
        else:
            assert used[iw] == 0
    close(dev.get(L.F_PHI), numpy.array([wk['phi'] for wk in walkers]), 1e-9)
    close(dev.get(L.F_WEIGHT), numpy.array([wk['weight'] for wk in walkers]), 1e-8)
    close(dev.get(L.F_OT), numpy.array([wk['ot'] for wk in walkers]), 1e-8)
    dev.close()


@pytest.mark.parametrize("nx,ny,ne", [(8, 8, 30), (10, 10, 45), (16, 16, 128)])
def test_hirsch_large_lattices(nx, ny, ne):
    """The discrete Hirsch step (propagation/hubbard.py:148-225,285-312) beyond N = 45: 8x8 with 30+30 electrons
    (inverse overlaps still in LDS) and the BASELINE configs[3] lattice 16x16 with 128+128 (inverse overlaps from
    the register-resident Gauss-Jordan kernel, updated in place in global memory).  Kinetic importance sampling,
    site loop (chosen fields exact), second kinetic step, against the oracle."""
    run_hirsch_lattice(nx, ny, ne, ne)
