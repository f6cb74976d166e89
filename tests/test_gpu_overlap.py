"""run_batched with the block boundary overlapped (the head of the first step of a block is enqueued before the host
waits for the sums of the block just ended: afq_propagate_begin / _finish, afq_estimates_get_begin / _end) against the
plain order: the energy shift enters a step only in its weight update, so every number must be IDENTICAL -- walkers,
weights, block estimates -- including across re-orthogonalisations and combs that fall on a block boundary."""
import numpy
import pytest

from pauxy_amd import systems, trial as trial_mod
from pauxy_amd.context import release_context
from pauxy_amd.qmc.afqmc import AFQMC

pytestmark = pytest.mark.gpu


def drive(overlap, nsteps_block, total, ride=True):
    s = systems.synthetic_generic(20, 30, (5, 4), seed=9)
    t = trial_mod.rhf_trial_generic(s)
    opts = {'qmc': {'timestep': 0.01, 'num_steps': nsteps_block, 'blocks': 10 ** 6, 'stabilise_freq': 4,
                    'pop_control_freq': 3, 'num_walkers': 24, 'rng_seed': 5},
            'propagator': {'device_rng': True, 'rng_seed': 11},
            'estimators': {'mixed': {'verbose': False, 'energy_eval_freq': 2}}}
    afqmc = AFQMC(options=opts, system=s, trial=t)
    afqmc.ride_estimates = ride
    eshift = afqmc.run_batched(total // 2, first_step=1, eshift=0.0, overlap_blocks=overlap)
    # a second call continues the run: the boundary logic restarts cleanly with nothing half done
    eshift = afqmc.run_batched(total - total // 2, first_step=total // 2 + 1, eshift=eshift, overlap_blocks=overlap)
    mixed = afqmc.estimators.estimators['mixed']
    out = dict(blocks=numpy.array(mixed.blocks), eshift=eshift, weight=afqmc.psi._mirror('weight').copy(),
               phi=numpy.array([w.phi for w in afqmc.psi.walkers]), ot=afqmc.psi._mirror('ot').copy())
    release_context(s, t)
    return out


@pytest.mark.parametrize("nsteps_block,total", [(4, 24), (6, 36), (1, 8)])
def test_overlapped_block_boundary_is_bit_identical(nsteps_block, total):
    a = drive(False, nsteps_block, total)
    b = drive(True, nsteps_block, total)
    assert a['blocks'].shape == b['blocks'].shape and a['blocks'].shape[0] >= total // nsteps_block
    # column 0 = step, the last column = wall time of the block
    assert numpy.array_equal(a["blocks"][:, :-1], b["blocks"][:, :-1], equal_nan=True)   # blocks without an energy evaluation: 0 / 0
    assert a['eshift'] == b['eshift'] and a['eshift'] != 0.0
    for k in ('weight', 'phi', 'ot'):
        assert numpy.array_equal(a[k], b[k]), k


def test_estimator_terms_riding_on_the_weight_update_equal_the_separate_launch():
    """afq_estimates_fuse_next: on the steps that neither comb nor evaluate the energy nor end a block the weight update
    adds every walker's estimator terms to per-walker accumulators, folded into the sums by the next summation launch or
    fetch (another association of the same additions: equal to rounding).  Blocks of 6 steps with comb / 3 and energy / 2
    leave ride steps 1 and 5 of every block; blocks of 4 exercise the fold at a fetch that follows a comb step."""
    for nsteps_block, total in ((6, 36), (7, 28)):
        a = drive(True, nsteps_block, total, ride=False)
        b = drive(True, nsteps_block, total, ride=True)
        assert a['blocks'].shape == b['blocks'].shape
        x, y = a['blocks'][:, 1:-1], b['blocks'][:, 1:-1]
        ok = numpy.isfinite(x)
        assert numpy.array_equal(ok, numpy.isfinite(y))
        assert numpy.max(numpy.abs(x[ok] - y[ok]) / (1e-300 + numpy.abs(x[ok]) + numpy.abs(y[ok]))) < 1e-12
        # the shift derived from the sums differs by rounding, hence the weights
        assert abs(a['eshift'] - b['eshift']) < 1e-10 * max(1.0, abs(a['eshift']))
        assert numpy.max(numpy.abs(a['weight'] - b['weight'])) < 1e-9
        assert numpy.max(numpy.abs(a['phi'] - b['phi'])) < 1e-9


def test_per_spin_ghalf_not_stored_on_announced_steps_is_recomputed_on_demand():
    """afq_estimates_fuse_next announces a step behind which nothing but the next force bias reads the Green's function:
    the kernel then leaves the per-spin Ghalf unwritten (RHF-type trial: the force bias contracts the spin sum).  Asking
    for the energy or for Ghalf afterwards must give what a handle without the announcement gives, and the walkers of
    both handles must stay bit-identical over further steps."""
    from oracle import afqmc_ref as ref
    from pauxy_amd import _lib as L
    from pauxy_amd.propagation.setup import generic_propagator_arrays
    from tests.helpers import make_device
    M, K, N, nw = 24, 30, 5, 64
    s = systems.synthetic_generic(M, K, (N, N), seed=11)
    t = trial_mod.rhf_trial_generic(s)
    BH1, mf = generic_propagator_arrays(s, t, 0.01)
    model = ref.RefModel('generic', M, N, N, t.psi, BH1, mf, 0.01, hs_pot=s.hs_pot, rchol=t._rchol,
                         H1=s.H1.astype(complex), ecore=s.ecore)
    rng = numpy.random.RandomState(3)
    phi = numpy.array([t.psi + 0.05 * (rng.rand(M, 2 * N) + 1j * rng.rand(M, 2 * N)) for _ in range(nw)])
    devs = [make_device(model, nw) for _ in range(2)]
    for d in devs:
        d.set(L.F_PHI, phi)
        d.set(L.F_OT, d.calc_overlap())
    for step in range(4):
        xi = rng.normal(size=(nw, K))
        devs[0].estimates_fuse_next()                       # announced: per-spin Ghalf not stored
        devs[0].propagate(xi, 0.1)
        devs[1].propagate(xi, 0.1)
        devs[1].estimates_update(False)
        if step == 1:
            assert numpy.array_equal(devs[0].local_energy(), devs[1].local_energy())
        if step == 2:
            assert numpy.array_equal(devs[0].get(L.F_GHALF), devs[1].get(L.F_GHALF))
        if step == 3:
            # the same through afq_walkers_device_ptr (a caller that maps the buffer, e.g. a custom estimator): the pointer
            # must hand out the CURRENT walkers' per-spin Ghalf, not an earlier step's
            import ctypes
            hip = ctypes.CDLL('libamdhip64.so')
            got = []
            for d in devs:
                ptr, per = ctypes.c_void_p(), ctypes.c_int64()
                assert d.lib.afq_walkers_device_ptr(d.h, L.F_GHALF, ctypes.byref(ptr), ctypes.byref(per)) == 0
                d.sync()
                buf = numpy.empty((nw, 2 * N, M), dtype=numpy.complex128)
                assert per.value == buf[0].nbytes
                assert hip.hipMemcpy(buf.ctypes.data_as(ctypes.c_void_p), ptr, ctypes.c_size_t(buf.nbytes), 2) == 0
                got.append(buf)
            assert numpy.array_equal(got[0], got[1])
            assert numpy.array_equal(got[0], devs[1].get(L.F_GHALF))
        for f in (L.F_PHI, L.F_WEIGHT, L.F_OT, L.F_HYBRID_ENERGY):
            assert numpy.array_equal(devs[0].get(f), devs[1].get(f)), (step, f)
    assert numpy.allclose(devs[0].estimates_get(), devs[1].estimates_get(), rtol=1e-12, atol=0)
    for d in devs:
        d.close()


def test_update_that_hands_the_block_over_equals_update_plus_fetch():
    """afq_estimates_update_publish: the summation launch of the update writes the block's sums into the mapped host buffer
    itself (estimators/mixed.py:211-225 + :261-273 at the step that ends a block).  Against afq_estimates_update followed by
    afq_estimates_get_begin on a second handle: the same sums bit for bit, zeroed alike, and a second hand-over while one is
    in flight is refused (AFQ_ESTATE)."""
    from oracle import afqmc_ref as ref
    from pauxy_amd import _lib as L
    from pauxy_amd.propagation.setup import generic_propagator_arrays
    from tests.helpers import make_device
    M, K, N, nw = 24, 30, 5, 64
    s = systems.synthetic_generic(M, K, (N, N), seed=13)
    t = trial_mod.rhf_trial_generic(s)
    BH1, mf = generic_propagator_arrays(s, t, 0.01)
    model = ref.RefModel('generic', M, N, N, t.psi, BH1, mf, 0.01, hs_pot=s.hs_pot, rchol=t._rchol,
                         H1=s.H1.astype(complex), ecore=s.ecore)
    rng = numpy.random.RandomState(4)
    phi = numpy.array([t.psi + 0.05 * (rng.rand(M, 2 * N) + 1j * rng.rand(M, 2 * N)) for _ in range(nw)])
    devs = [make_device(model, nw) for _ in range(2)]
    for d in devs:
        d.set(L.F_PHI, phi)
        d.set(L.F_OT, d.calc_overlap())
    for block in range(3):
        for step in range(3):
            xi = rng.normal(size=(nw, K))
            last = step == 2
            if not last:
                devs[0].estimates_fuse_next()                # (the ride steps' accumulators are folded in by the same launch)
                devs[1].estimates_fuse_next()
            for d in devs:
                d.propagate(xi, 0.1)
            if last:
                devs[0].estimates_update_publish(True, zero=True)
                with pytest.raises(L.AfqError) as err:
                    devs[0].estimates_update_publish(True, zero=True)
                assert err.value.code == -2                  # AFQ_ESTATE
                devs[1].estimates_update(True)
                devs[1].estimates_get_begin(zero=True)
        a, b = devs[0].estimates_get_end(), devs[1].estimates_get_end()
        assert numpy.array_equal(a, b) and numpy.abs(a).max() > 0.0, block
    for d in devs:
        d.close()
