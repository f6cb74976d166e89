"""Files either side of the device path, end to end on the GPU:
a run fed from QMCPACK-format files equals the run fed from the arrays; the estimator file holds the
blocks the driver produced (estimators/utils.py:279-327 layout); a restart file written every
``write_freq`` steps restores the walkers (walkers/handler.py:432-485)."""
import json

import numpy
import pytest

from pauxy_amd import systems, trial as trial_mod
from pauxy_amd import _lib as L
from pauxy_amd.context import release_context
from pauxy_amd.qmc.afqmc import AFQMC
from pauxy_amd.utils import io as aio

pytestmark = pytest.mark.gpu


def options(tmp_path=None, blocks=4, walkers=None, est=None):
    opt = {'qmc': {'timestep': 0.01, 'num_steps': 5, 'blocks': blocks, 'stabilise_freq': 5, 'pop_control_freq': 5,
                   'num_walkers': 12, 'rng_seed': 11},
           'propagator': {},
           'estimators': {'mixed': {'energy_eval_freq': 5, 'verbose': False}}}
    opt['estimators'].update(est or {})
    if walkers:
        opt['walkers'] = walkers
    return opt


def run(system, trial, opt):
    a = AFQMC(options=opt, system=system, trial=trial)
    a.run(verbose=False)
    a.finalise()
    return a


def test_run_from_files_equals_run_from_arrays(tmp_path):
    s0 = systems.synthetic_generic(12, 30, (3, 3), seed=5)
    psi = trial_mod.rhf_trial_generic(s0).psi
    psi[numpy.abs(psi) < 1e-8] = 0.0                       # write_nomsd drops such elements (io.py:472)
    t0 = trial_mod.SingleDetTrial(s0, psi)
    ham, wfn = str(tmp_path / 'ham.h5'), str(tmp_path / 'wfn.h5')
    aio.write_qmcpack_dense(s0.H1[0], s0.chol_vecs, s0.nelec, s0.nbasis, enuc=s0.ecore, filename=ham)
    aio.write_qmcpack_wfn(wfn, (numpy.array([1.0 + 0j]), t0.psi[None].copy()), 'uhf', s0.nelec, s0.nbasis)
    a0 = run(s0, t0, options())
    blocks0 = numpy.array(a0.estimators.estimators['mixed'].blocks)
    phi0 = a0.psi.dev.get(L.F_PHI)
    release_context(s0, t0)
    s1 = systems.get_system({'name': 'Generic', 'integrals': ham, 'nup': 3, 'ndown': 3})
    t1 = trial_mod.get_trial_wavefunction(s1, {'filename': wfn})
    assert numpy.array_equal(s1.chol_vecs, s0.chol_vecs) and numpy.array_equal(s1.H1, s0.H1) and s1.ecore == s0.ecore
    assert numpy.array_equal(t1.psi, t0.psi)
    a1 = run(s1, t1, options())
    blocks1 = numpy.array(a1.estimators.estimators['mixed'].blocks)
    assert numpy.array_equal(blocks1[:, 1:10], blocks0[:, 1:10])          # same inputs, same seed: bit-identical
    assert numpy.array_equal(a1.psi.dev.get(L.F_PHI), phi0)
    release_context(s1, t1)


def test_estimator_file_and_restart(tmp_path):
    s = systems.synthetic_generic(10, 24, (2, 2), seed=3)
    t = trial_mod.rhf_trial_generic(s)
    base = str(tmp_path / 'estimates')
    restart = str(tmp_path / 'restart.h5')
    est = {'basename': base, 'flush_every': 3,
           'back_propagation': {'tau_bp': 0.05, 'one_rdm': True, 'evaluate_energy': True}}
    a = run(s, t, options(est=est, walkers={'write_freq': 10, 'write_file': restart}))
    mixed = a.estimators.estimators['mixed']
    bp = a.estimators.estimators['back_prop']
    assert a.estimators.filename == base + '.0.h5'
    with aio.h5.File(a.estimators.filename, 'r') as f:
        assert [x.decode() for x in f['basic/headers'][:]] == mixed.header
        names = f['basic/energies'].keys()
        assert names == ['%09d' % i for i in range(len(mixed.blocks))] and len(names) == 4
        for i, row in enumerate(mixed.blocks):
            got = f['basic/energies/' + names[i]][:]
            assert got.dtype == numpy.complex128 and numpy.array_equal(got, row)
        meta = json.loads(f['metadata'][()])
        assert meta['qmc']['dt'] == 0.01 and meta['system']['nbasis'] == 10 and meta['estimators']['nbp'] == 5
        assert [x.decode() for x in f['back_propagated/headers'][:]] == bp.header
        nwin = len(bp.denominator)
        assert nwin >= 3 and f['back_propagated/one_rdm_5'].keys() == ['%09d' % i for i in range(nwin)]
        for i in range(nwin):
            assert numpy.array_equal(f['back_propagated/one_rdm_5/%09d' % i][:], bp.one_rdm[i])
            assert numpy.array_equal(f['back_propagated/denominator_5/%09d' % i][:], [bp.denominator[i]])
            assert numpy.array_equal(f['back_propagated/energies_5/%09d' % i][:], bp.energies[i])
    # restart file: the state after the last write step (step 20 = the final step)
    dev = a.psi.dev
    phi, w, ot, ph = dev.get(L.F_PHI), dev.get(L.F_WEIGHT), dev.get(L.F_OT), dev.get(L.F_PHASE)
    with aio.h5.File(restart, 'r') as f:
        assert len(f) == 12
        for i in range(12):
            row = f['walker_%d' % i][:]
            assert row.shape == (3 + phi[i].size,)
            assert row[0] == w[i] and row[1] == ph[i] and row[2] == ot[i] and numpy.array_equal(row[3:], phi[i].ravel())
    release_context(s, t)
    # a new population started from the file: same walkers; walkers the file does not hold keep the initial state
    s2 = systems.synthetic_generic(10, 24, (2, 2), seed=3)
    t2 = trial_mod.rhf_trial_generic(s2)
    with aio.h5.File(restart, 'a') as f:
        del f['walker_7']
    b = AFQMC(options=options(walkers={'read_file': restart}), system=s2, trial=t2)
    phi2, w2, ot2 = b.psi.dev.get(L.F_PHI), b.psi.dev.get(L.F_WEIGHT), b.psi.dev.get(L.F_OT)
    keep = [i for i in range(12) if i != 7]
    assert numpy.array_equal(phi2[keep], phi[keep]) and numpy.array_equal(w2[keep], w[keep])
    assert numpy.array_equal(ot2[keep], ot[keep])
    assert numpy.array_equal(phi2[7], numpy.asarray(t2.init)) and w2[7] == 1.0
    assert b.psi.walkers[3].weight == w[3] and numpy.array_equal(b.psi.walkers[3].phi, phi[3])
    release_context(s2, t2)


def test_no_file_without_a_name(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    s = systems.synthetic_generic(8, 10, (2, 2), seed=1)
    t = trial_mod.rhf_trial_generic(s)
    a = run(s, t, options(blocks=1))
    assert a.estimators.filename is None and list(tmp_path.iterdir()) == []
    release_context(s, t)
