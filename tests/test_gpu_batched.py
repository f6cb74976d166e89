"""The benchmarked loop is the tested loop: ``AFQMC.run_batched`` (what ``bench.py`` times -- batched device
calls only, in-kernel weight cap, comb without read-back, estimator sums kept on the device for a whole block)
replays the golden trajectories of the genuine reference with the same tolerances as the per-walker loop of
``tests/test_gpu_traj.py`` (1e-8 relative, comb decisions exact, block rows), and at the BASELINE configs[2]
size it is compared with ``AFQMC.run`` on identical host-drawn fields.  Reference loop: qmc/afqmc.py:223-255."""
import numpy
import pytest

from pauxy_amd import _lib as L
from pauxy_amd import systems, trial as trial_mod
from pauxy_amd.context import release_context
from pauxy_amd.qmc.afqmc import AFQMC
from tests.test_gpu_traj import close, replay, run_bp, run_hirsch

pytestmark = pytest.mark.gpu


def generic_from(d):
    na, nb = [int(x) for x in d['nelec']]
    return systems.Generic((na, nb), numpy.array([d['h1e'], d['h1e']]), d['chol'], float(d['ecore']))


def test_batched_traj_generic(golden, monkeypatch):
    d = golden('traj_generic.npz')
    s = generic_from(d)
    t = trial_mod.SingleDetTrial(s, d['psi'])
    est = replay(d, s, t, {}, monkeypatch, batched=True)
    assert est[2].real == pytest.approx(3.8763193646854273, rel=1e-8)        # qmc/tests/test_afqmc.py:227


def test_batched_traj_hubbard_c1(golden, monkeypatch):
    d = golden('traj_hubbard_c1.npz')
    s = systems.Hubbard(4, 4, 8, 8, float(d['U']))
    t = trial_mod.SingleDetTrial(s, d['psi'], name='UHF')
    replay(d, s, t, {'hubbard_stratonovich': 'continuous'}, monkeypatch, batched=True)


def test_batched_traj_hubbard(golden, monkeypatch):
    d = golden('traj_hubbard.npz')
    na, nb = [int(x) for x in d['nelec']]
    s = systems.Hubbard(4, 4, na, nb, float(d['U']))
    t = trial_mod.SingleDetTrial(s, d['psi'], name='UHF')
    est = replay(d, s, t, {'hubbard_stratonovich': 'continuous'}, monkeypatch, batched=True)
    assert est[2].real == pytest.approx(-152.91937839611, rel=1e-8)          # qmc/tests/test_afqmc.py:186


def test_batched_traj_ueg(golden, monkeypatch):
    d = golden('traj_ueg.npz')
    s = systems.UEG(float(d['sys_rs']), 7, 7, float(d['sys_ecut']))
    t = trial_mod.hartree_fock_ueg(s)
    est = replay(d, s, t, {}, monkeypatch, batched=True)
    assert est[2].real == pytest.approx(16.33039729324558, rel=1e-8)         # qmc/tests/test_afqmc.py:87


def test_batched_traj_msd(golden, monkeypatch):
    d = golden('traj_msd.npz')
    s = generic_from(d)
    t = trial_mod.MultiDetTrial(s, (d['coeffs'], d['psi']), init=d['phi0'][0])
    replay(d, s, t, {}, monkeypatch, batched=True)


def test_batched_traj_free_projection_and_local_energy(golden, monkeypatch):
    for name, opts in (('traj_hubbard_fp.npz', {'free_projection': True}), ('traj_hubbard_le.npz', {'hybrid': False})):
        d = golden(name)
        s = systems.Hubbard(4, 4, 8, 8, float(d['U']))
        t = trial_mod.SingleDetTrial(s, d['psi'], name='UHF')
        replay(d, s, t, dict(opts, hubbard_stratonovich='continuous'), monkeypatch, batched=True)


def test_batched_traj_back_propagation(golden, monkeypatch):
    """run_batched drives the back-propagation estimator at the reference cadence (estimators/handler.py:156-162)."""
    rdm = run_bp(golden, monkeypatch, 'traj_bp.npz', None, batched=True)
    assert rdm[11, 0, 1, 3].real == pytest.approx(-0.121883381144845, rel=1e-7)   # qmc/tests/test_afqmc.py:276


def test_batched_traj_hirsch(golden, monkeypatch):
    est = run_hirsch(golden, monkeypatch, 'traj_hubbard_hirsch.npz', batched=True)
    assert est[2].real == pytest.approx(-152.68468568462666, rel=1e-8)


# ----------------------------------------------------------------------------- BASELINE configs[2] size
C3 = dict(M=100, K=500, N=25, nw=256)


def build_afqmc(s, t, nw, device_rng=False, prop=None, dt=0.005):
    options = {'qmc': {'timestep': dt, 'num_steps': 10, 'blocks': 2, 'stabilise_freq': 10, 'pop_control_freq': 5,
                       'num_walkers': nw, 'rng_seed': 7},
               'propagator': dict({'device_rng': device_rng, 'rng_seed': 7}, **(prop or {})),
               'estimators': {'mixed': {'verbose': False}}}
    return AFQMC(options=options, system=s, trial=t), s, t


def c3_afqmc(device_rng=False):
    s = systems.synthetic_generic(C3['M'], C3['K'], (C3['N'], C3['N']), seed=7)
    t = trial_mod.rhf_trial_generic(s)
    return build_afqmc(s, t, C3['nw'], device_rng)


def c2_afqmc(device_rng=False):
    """BASELINE configs[1]: UEG rs=2, 14 electrons, 93 plane waves, 256 walkers."""
    s = systems.UEG(2.0, 7, 7, 4.0)
    return build_afqmc(s, trial_mod.hartree_fock_ueg(s), 256, device_rng)


def c4_afqmc(device_rng=False):
    """BASELINE configs[3] per GPU: 16x16 Hubbard U=8, 128+128 electrons, UHF trial, 256 walkers."""
    s = systems.Hubbard(16, 16, 128, 128, 8.0)
    t = trial_mod.uhf_trial_hubbard(s, ueff=0.4)
    return build_afqmc(s, t, 256, device_rng, {'hubbard_stratonovich': 'continuous'})


def c4_small_step_afqmc(device_rng=False):
    """The same lattice with a ten times smaller time step: the zero energy shift of the first block (E ~ +700) then
    costs a factor 0.03 in weight over the block instead of 1e-15, and every walker stays above the 1e-8 threshold."""
    s = systems.Hubbard(16, 16, 128, 128, 8.0)
    t = trial_mod.uhf_trial_hubbard(s, ueff=0.4)
    return build_afqmc(s, t, 256, device_rng, {'hubbard_stratonovich': 'continuous'}, dt=0.0005)


def run_c3(batched, fetch, ride=False, make=c3_afqmc):
    afqmc, s, t = make()
    nw = afqmc.psi.nw
    # ride: the estimator terms of the plain steps ride on their weight update (afq_estimates_fuse_next) -- the same
    # additions in another association, so the block sums (hence the shift, hence the weights) move by rounding
    afqmc.ride_estimates = ride
    numpy.random.seed(1234)
    # walkers that all start on the trial keep weights within a few per cent of each other for the first blocks and
    # the comb would clone nobody: start from a spread population
    afqmc.psi.dev.set(L.F_WEIGHT, numpy.exp(0.6 * numpy.random.RandomState(5).normal(size=nw)))
    afqmc.psi._invalidate()
    rec = dict(weight=[], ot=[], ehyb=[], pix=[])

    def on_step(step, psi):
        rec['weight'].append(psi._mirror('weight').copy())
        rec['ot'].append(psi._mirror('ot').copy())
        rec['ehyb'].append(psi._mirror('hybrid_energy').copy())
        if fetch and step % afqmc.qmc.npop_control == 0:
            rec['pix'].append(psi.last_parent_ix.copy())

    if batched:
        afqmc.run_batched(on_step=on_step, fetch_popcontrol=fetch)
    else:
        afqmc.run(on_step=on_step)
    blocks = numpy.array(afqmc.estimators.estimators['mixed'].blocks)
    phi = afqmc.psi.dev.get(L.F_PHI)
    release_context(s, t)
    return {k: numpy.array(v) for k, v in rec.items()}, blocks, phi


@pytest.mark.parametrize("make", [c2_afqmc, c4_afqmc, c4_small_step_afqmc],
                         ids=["C2-ueg-256", "C4-hubbard16x16-256", "C4-hubbard16x16-256-all-alive"])
def test_c2_c4_run_batched_equals_run(make):
    """20 steps of BASELINE configs[1] (UEG, 93 plane waves) and configs[3] (16x16 Hubbard, 128+128 electrons: the
    GEMM + register-resident Gauss-Jordan Green's function, the unfused propagator) at their stated 256 walkers
    per GPU, host-drawn fields: the batched loop bench-style (no read-back at the comb) against the per-walker loop
    -- same kernels on the same data, walkers bit-equal -- and with the comb read back, identical comb decisions."""
    a, blocks_a, phi_a = run_c3(False, True, make=make)
    b, blocks_b, phi_b = run_c3(True, True, make=make)
    c, blocks_c, phi_c = run_c3(True, False, make=make)
    assert numpy.all(numpy.isfinite(phi_a.view(float))) and numpy.all(numpy.isfinite(a['weight']))
    # the two batched variants (comb read back or not) queue identical work: bit-equal
    for key in ('weight', 'ot', 'ehyb'):
        assert numpy.array_equal(b[key], c[key]), key
    assert numpy.array_equal(phi_b, phi_c)
    for other, blocks, phi in ((b, blocks_b, phi_b), (c, blocks_c, phi_c)):
        for key in ('weight', 'ot', 'ehyb'):
            if make is c2_afqmc or make is c4_small_step_afqmc:
                # every walker alive: Slater matrices and overlaps bit-equal over all 20 steps; the block sums of the two
                # loops are the same terms in another association (the per-walker loop adds every step on the host), so
                # the shift derived from block 1 may differ in its last bit (E ~ +700 on the lattice) and with it, from step 11
                # on, the weights and the hybrid energies (one ulp).  The electron gas was bit-equal
                # throughout up to round 4 and differs by one ulp in three weights since the round-5 Green's function kernel
                # for at most 8 electrons per spin (another summation order in the energies that make the shift)
                assert numpy.array_equal(a[key][:10], other[key][:10]), key
                if key == 'ot':
                    assert numpy.array_equal(a[key], other[key]), key
                close(a[key], other[key], 1e-13)
            else:
                # C4 at the time step of the other configurations: bit-equal through the first re-orthogonalisation +
                # energy evaluation (step 10).  The zero shift of the
                # first block (E ~ +700 on this lattice) takes every walker below the 1e-8 threshold from step 8 to 10: the
                # per-walker loop then has nothing to propagate and makes no call, the batched loop queues its step, whose
                # closing Green's function evaluation sees the freshly re-orthogonalised walkers (Ghalf of phi R^-1
                # instead of the one kept across the QR: 1e-15); the comb revives the
                # population and the difference spreads at rounding level
                assert numpy.array_equal(a[key][:10], other[key][:10]), key
                close(a[key], other[key], 1e-11)
        if make is c4_afqmc:
            close(phi, phi_a, 1e-11)
        else:
            assert numpy.array_equal(phi_a, phi)
        close(blocks[:, 1:10], blocks_a[:, 1:10], 1e-12)
    assert numpy.array_equal(a['pix'], b['pix'])
    assert a['pix'].shape == (4, 256) and a['pix'].max() >= 2             # the comb did clone walkers


def test_c3_run_batched_equals_run():
    """20 steps at Nbasis=100, Nchol=500, 256 walkers with host-drawn fields: the batched loop (weight cap inside the
    weight-update kernel, comb with and without read-back, per-block estimator fetch) against the per-walker
    loop.  The two loops queue the same kernels on the same data, so everything the walkers carry is bit-equal;
    the block rows differ only in where the step-0 row is added (1e-13)."""
    a, blocks_a, phi_a = run_c3(False, True)
    b, blocks_b, phi_b = run_c3(True, True)
    c, blocks_c, phi_c = run_c3(True, False)         # nothing read back at the comb (bench.py's mode)
    for other, blocks, phi in ((b, blocks_b, phi_b), (c, blocks_c, phi_c)):
        for key in ('weight', 'ot', 'ehyb'):
            assert numpy.array_equal(a[key], other[key]), key
        assert numpy.array_equal(phi_a, phi)
        close(blocks[:, 1:10], blocks_a[:, 1:10], 1e-12)
    assert numpy.array_equal(a['pix'], b['pix'])
    assert a['pix'].shape == (4, C3['nw']) and a['pix'].max() >= 2         # the comb did clone walkers
    # bench.py's mode proper: estimator terms riding on the weight update as well -- equal to rounding
    d, blocks_d, phi_d = run_c3(True, False, ride=True)
    for key in ('weight', 'ot', 'ehyb'):
        close(d[key], a[key], 1e-11)
    close(phi_d, phi_a, 1e-11)
    close(blocks_d[:, 1:10], blocks_a[:, 1:10], 1e-12)


def test_collapsed_population_is_reported():
    """walkers/handler.py:236-241: total weight < 1e-8 stops the run; on the asynchronous comb the flag surfaces at
    the next afq_estimates_get (AFQ_EWEIGHT) and nothing was cloned or reset."""
    afqmc, s, t = c3_afqmc()
    dev = afqmc.psi.dev
    dev.set(L.F_WEIGHT, numpy.full(dev.nw, 1e-12))
    dev.popcontrol_comb(0.3, dev.nw, fetch=False)
    with pytest.raises(L.AfqError) as e:
        dev.estimates_get(zero=True)
    assert e.value.code == L.AFQ_EWEIGHT
    assert numpy.all(dev.get(L.F_WEIGHT) == 1e-12)
    release_context(s, t)
