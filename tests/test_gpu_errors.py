"""Error behaviour of the C ABI (status codes + afq_last_error text surfaced as AfqError): wrong call order,
out-of-range arguments, unsupported configurations.  The reference raises Python exceptions / sys.exit at the
same places ("conventions at this boundary")."""
import numpy
import pytest

from pauxy_amd import _lib as L
from pauxy_amd import systems, trial as trial_mod
from pauxy_amd.device import AfqDevice
from pauxy_amd.propagation import setup

pytestmark = pytest.mark.gpu


def small_generic():
    s = systems.synthetic_generic(12, 10, (3, 3), seed=1)
    t = trial_mod.rhf_trial_generic(s)
    BH1, mf = setup.generic_propagator_arrays(s, t, 0.01)
    return s, t, BH1, mf


def test_call_order_and_ranges():
    s, t, BH1, mf = small_generic()
    dev = AfqDevice(0)
    with pytest.raises(L.AfqError) as e:                    # nothing set yet
        dev.walkers_alloc(4)
    assert e.value.code == -2
    dev.set_system_generic(s.hs_pot, t._rchol, s.H1.astype(complex), 0.0, 3, 3)
    dev.walkers_alloc(4)
    with pytest.raises(L.AfqError) as e:                    # no trial, no propagator
        dev.propagate(numpy.zeros((4, 10)), 0.0)
    assert e.value.code == -2 and 'must be set' in str(e.value)
    dev.set_trial(t.psi)
    with pytest.raises(L.AfqError) as e:
        dev.propagate(numpy.zeros((4, 10)), 0.0)
    assert e.value.code == -2 and 'propagator' in str(e.value)
    dev.set_propagator(BH1, mf, 0.01)
    with pytest.raises(L.AfqError) as e:                    # walker range
        dev.set(L.F_WEIGHT, numpy.ones(2), first=3)
    assert e.value.code == -1
    with pytest.raises(L.AfqError) as e:
        dev.copy_walker(0, 9)
    assert e.value.code == -1
    with pytest.raises(L.AfqError) as e:                    # back-propagation not configured
        dev.bp_update(t.psi, 5)
    assert e.value.code == -2
    with pytest.raises(L.AfqError) as e:                    # multi-determinant trial after the walkers exist
        dev.set_trial_multi(numpy.array([t.psi, t.psi]), numpy.ones(2), numpy.concatenate([t._rchol, t._rchol]))
    assert e.value.code == -2
    with pytest.raises(L.AfqError) as e:                    # Hirsch transformation needs a Hubbard system
        dev.set_propagator_hirsch(BH1, 0.01)
    assert e.value.code == -2
    # the handle is still usable after the errors
    dev.set(L.F_PHI, numpy.array([t.psi] * 4))
    dev.propagate(numpy.zeros((4, 10)), 0.0)
    assert numpy.all(numpy.isfinite(dev.get(L.F_WEIGHT)))
    # a step in two calls: finish without begin, another step / a re-orthogonalisation inside a half-done step
    with pytest.raises(L.AfqError) as e:
        dev.propagate_finish(0.0)
    assert e.value.code == -2
    dev.propagate_begin(numpy.zeros((4, 10)))
    for call in (lambda: dev.propagate_begin(numpy.zeros((4, 10))), lambda: dev.reortho(),
                 lambda: dev.propagate(numpy.zeros((4, 10)), 0.0)):
        with pytest.raises(L.AfqError) as e:
            call()
        assert e.value.code == -2 and 'half done' in str(e.value)
    dev.set_weight_cap(0.0)                                  # allowed in between
    dev.propagate_finish(0.0)
    # asynchronous fetch of the estimator sums: one in flight, end needs a begin
    with pytest.raises(L.AfqError) as e:
        dev.estimates_get_end()
    assert e.value.code == -2
    dev.estimates_get_begin()
    with pytest.raises(L.AfqError) as e:
        dev.estimates_get_begin()
    assert e.value.code == -2
    assert numpy.all(numpy.isfinite(dev.estimates_get_end()))
    assert numpy.all(numpy.isfinite(dev.get(L.F_WEIGHT)))
    dev.close()


def test_unsupported_sizes():
    s = systems.Hubbard(12, 12, 130, 130, 4.0)              # N = 130 > 128: beyond the discrete Hirsch propagator
    M = 144
    q = numpy.linalg.qr(numpy.random.RandomState(1).rand(M, 130))[0]
    psi = numpy.hstack([q, q]).astype(complex)
    dev = AfqDevice(0)
    dev.set_system_hubbard(s.T.astype(complex), 4.0, 130, 130)
    dev.set_trial(psi)
    dev.set_propagator_hirsch(numpy.array([numpy.eye(M), numpy.eye(M)], dtype=complex), 0.01)
    dev.walkers_alloc(2)
    dev.set(L.F_PHI, numpy.array([psi] * 2))
    with pytest.raises(L.AfqError) as e:
        dev.hirsch_kinetic()
    assert e.value.code == -5 and 'N <= 128' in str(e.value)
    with pytest.raises(L.AfqError) as e:                    # continuous step on a handle configured for Hirsch
        dev.propagate(numpy.zeros((2, M)), 0.0)
    assert e.value.code == -2
    dev.close()


def test_back_propagation_hubbard_needs_the_discrete_fields():
    """estimators/back_propagation.py:117-125: for a Hubbard system the reference back-propagates with
    propagation/hubbard.py:568-672, which reads the history as 0 / 1 fields; a continuous propagator is refused, and
    so are restored weights with the discrete one (FieldConfig.push records no factors, walkers/stack.py:35-49)."""
    s = systems.Hubbard(4, 4, 7, 7, 4.0)
    M = 16
    q = numpy.linalg.qr(numpy.random.RandomState(2).rand(M, 7))[0]
    psi = numpy.hstack([q, q]).astype(complex)
    dev = AfqDevice(0)
    dev.set_system_hubbard(s.T.astype(complex), 4.0, 7, 7)
    dev.set_trial(psi)
    dev.walkers_alloc(4)
    dev.set(L.F_PHI, numpy.array([psi] * 4))
    eye = numpy.array([numpy.eye(M), numpy.eye(M)], dtype=complex)
    dev.set_propagator(eye, numpy.zeros(M), 0.01)
    with pytest.raises(L.AfqError) as e:
        dev.bp_configure(4)
    assert e.value.code == -5 and 'discrete' in str(e.value)
    dev.set_propagator_hirsch(eye, 0.01)
    dev.bp_configure(4)
    with pytest.raises(L.AfqError) as e:
        dev.bp_update(psi, 5, restore_weights='full')
    assert e.value.code == -5 and 'restore_weights' in str(e.value)
    assert numpy.array_equal(dev.bp_steps(), numpy.zeros(4, dtype=numpy.int32))
    dev.propagate_hirsch(0.0)
    assert numpy.array_equal(dev.bp_steps(), numpy.ones(4, dtype=numpy.int32))
    dev.close()
