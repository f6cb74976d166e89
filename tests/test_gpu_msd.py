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


def run_steps(d, tag, hybrid, nw=3):
    """Walker 0 replays the reference's run; the other walkers start from perturbed states and are
    checked against the oracle."""
    m = msd_model(d, tag)
    dev = make_device(m, nw, hybrid=hybrid)
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


def test_phmsd_local_energy_weights(golden):
    w = run_steps(golden('msd_ops.npz'), 'PL_', False)
    assert w == pytest.approx(0.68797524675701, rel=1e-9)           # propagation/tests/test_generic.py:72


def test_phmsd_hybrid(golden):
    w = run_steps(golden('msd_ops.npz'), 'PH_', True)
    assert w == pytest.approx(0.7430443466368197, rel=1e-9)         # propagation/tests/test_generic.py:92


def test_nomsd(golden):
    run_steps(golden('msd_ops.npz'), 'N_', True)
