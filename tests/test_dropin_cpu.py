"""The drop-in boundary where there is no GPU: the plug-in classes of pauxy_amd over the test-only numpy stand-in for
AfqDevice (tests/oracle_device.py, oracle arithmetic).

What the build container adds on top (``python tests/golden/make_golden.py dropin``, needs /root/reference): the same
classes, on the same stand-in, driven by the GENUINE pauxy/qmc/afqmc.py with its three plug-in imports switched
 -- constructor, to_json/serialise, run, finalise -- reproducing the golden trajectories the
genuine classes produced, and recording what that driver touches into tests/golden/dropin_trace.json.  The tests here
hold this build's classes to that record and run the same trajectories through the restated loop."""
import json
import os

import numpy
import pytest

from pauxy_amd import systems, trial as trial_mod
from pauxy_amd.context import release_context
from pauxy_amd.estimators.mixed import local_energy
from pauxy_amd.utils import io as pio
from tests import dropin_checks, oracle_device, serialise_walk
from tests.test_gpu_traj import replay, run_hirsch


@pytest.fixture
def standin(monkeypatch):
    oracle_device.install(monkeypatch)
    return oracle_device.OracleDevice


def test_object_graph_survives_the_reference_serialise(golden, standin, monkeypatch, tmp_path):
    """Continuous.propagator._driver was a cycle and utils/misc.py:72-135 has no cycle guard."""
    monkeypatch.delenv('AFQ_ESTIMATES_FILE', raising=False)
    monkeypatch.chdir(tmp_path)
    shell, comm = dropin_checks.build_like_the_driver(golden('traj_hubbard_c1.npz'))
    text = dropin_checks.check_serialisable(shell)
    assert shell.estimators.filename == 'estimates.0.h5'                  # estimators/handler.py:62-64: the default
    with pio.h5.File('estimates.0.h5', 'r') as f:
        meta = f['metadata'][()]
    meta = meta.decode() if isinstance(meta, bytes) else str(meta)
    assert json.loads(meta) == json.loads(text)
    release_context(shell.system, shell.trial)


def test_plugin_objects_hold_no_handles_in_their_dict(golden, standin):
    """What serialise sees is data: no device handle, context, communicator or back reference in any __dict__."""
    shell, comm = dropin_checks.build_like_the_driver(golden('traj_hubbard_c1.npz'), {'write_file': False})
    seen = []

    def visit(obj, path):
        for k, v in vars(obj).items():
            assert not isinstance(v, (standin, type(comm))), path + '.' + k
            assert type(v).__name__ not in ('Context', 'AfqDevice', 'Walkers', 'Continuous'), path + '.' + k
            if serialise_walk.is_object(v):
                seen.append(path + '.' + k)
                visit(v, path + '.' + k)
    for name in ('propagators', 'estimators', 'psi'):
        visit(getattr(shell, name), name)
    assert 'propagators.propagator' in seen
    assert shell.psi.walkers[0]._h is shell.psi
    release_context(shell.system, shell.trial)


def test_surface_the_genuine_driver_touches(golden, standin):
    shell, comm = dropin_checks.build_like_the_driver(golden('traj_hubbard_c1.npz'), {'write_file': False})
    dropin_checks.check_surface(shell)
    release_context(shell.system, shell.trial)


def test_surface_with_the_back_propagated_estimator(golden, standin):
    """The same for the objects of a run with `estimators: {back_propagated: ...}` (Generic system): BackPropagation is
    built by Estimators, its nmax reaches Walkers as nbp / nprop_tot, everything survives the serialise walk."""
    d = golden('traj_bp.npz')
    s = systems.Generic((3, 3), numpy.array([d['h1e'], d['h1e']]), d['chol'], float(d['ecore']))
    t = trial_mod.SingleDetTrial(s, d['psi'])
    shell, comm = dropin_checks.build_shell(s, t, {'timestep': 0.005, 'num_steps': 10, 'blocks': 10, 'rng_seed': 8}, {},
                                            {'back_propagated': {'tau_bp': 0.025, 'one_rdm': True},
                                             'mixed': {'energy_eval_freq': 1}, 'write_file': False})
    assert shell.estimators.nbp == 5 and shell.psi.nbp == 5
    dropin_checks.check_surface(shell)
    tree = serialise_walk.walk(shell)
    json.dumps(tree)
    want = dropin_checks.trace_fixture()['cases']['generic_bp']['serialised']['estimators']['estimators']['back_prop']
    got = serialise_walk.kinds(tree)['estimators']['estimators']['back_prop']
    want = dict(want, output='null')                       # (this shell writes no file)
    assert got == want, dropin_checks._diff(got, want)
    release_context(s, t)


def test_trace_fixture_is_what_the_dropin_run_recorded():
    doc = dropin_checks.trace_fixture()
    assert doc['driver_imports_switched'] == [
        "from pauxy_amd.estimators.handler import Estimators",
        "from pauxy_amd.propagation.continuous import get_propagator_driver",
        "from pauxy_amd.walkers.handler import Walkers"]
    assert set(doc['cases']) == {'hubbard_c1', 'generic', 'ueg', 'generic_bp', 'hubbard_hirsch', 'hubbard_le', 'hubbard_fp'}
    for case in doc['cases'].values():
        assert max(case['max_rel_err'].values()) < 1e-8
    t = doc['trace']
    assert 'propagate_walker' in t['Propagator']['called'] and 'pop_control' in t['Walkers']['called']
    assert 'weight' in t['Walker']['read'] and 'weight' in t['Walker']['written']
    assert '__dict__' in t['Propagator']['read']                         # serialise walked it


CASES = [('traj_hubbard_c1.npz', {}), ('traj_hubbard_le.npz', {'hybrid': False}),
         ('traj_hubbard_fp.npz', {'free_projection': True}), ('traj_generic.npz', None), ('traj_ueg.npz', 'ueg')]


@pytest.mark.parametrize('name,extra', CASES, ids=[c[0][5:-4] for c in CASES])
def test_plugin_classes_over_the_standin_reproduce_the_golden_trajectory(golden, standin, monkeypatch, name, extra):
    """The host logic of the plug-in classes (mirrors, dirty flags, one launch per sweep, lazy Green's functions, block
    rows) with the oracle's arithmetic underneath, through the restated per-walker loop."""
    d = golden(name)
    if extra == 'ueg':
        s = systems.UEG(2.44, 7, 7, 2.0)
        t = trial_mod.hartree_fock_ueg(s)
        opts = {}
    elif extra is None:
        na, nb = [int(x) for x in d['nelec']]
        s = systems.Generic((na, nb), numpy.array([d['h1e'], d['h1e']]), d['chol'], float(d['ecore']))
        t = trial_mod.SingleDetTrial(s, d['psi'])
        opts = {}
    else:
        s = systems.Hubbard(4, 4, 8, 8, float(d['U']))
        t = trial_mod.SingleDetTrial(s, d['psi'], name='UHF')
        opts = dict({'hubbard_stratonovich': 'continuous'}, **extra)
    replay(d, s, t, opts, monkeypatch)
    dev = standin.instances[-1]
    assert dev.calls.count('propagate') == int(d['nsteps']) * int(d['nblocks'])   # one batched launch per step


@pytest.mark.parametrize('name,kw', [('traj_hubbard_hirsch.npz', {}), ('traj_hubbard_hirsch_charge.npz', {}),
                                     ('traj_hirsch_direct.npz', {'direct': True}),
                                     ('traj_hirsch_bp.npz', {'bp': {'tau_bp': 0.04, 'one_rdm': True}})],
                         ids=['single-site', 'charge', 'direct', 'back-propagation'])
def test_hirsch_plugin_class_over_the_standin(golden, standin, monkeypatch, name, kw):
    """The discrete-field propagator class (batched site loop with the reference's stream of uniforms, the half step that
    decides who still draws) over the stand-in, through the restated per-walker loop."""
    run_hirsch(golden, monkeypatch, name, **kw)


def test_back_propagation_plugin_class_over_the_standin(golden, standin, monkeypatch):
    """BackPropagation (window bookkeeping, split lengths, output rows) with the oracle's field history underneath."""
    d = golden('traj_bp.npz')
    s = systems.Generic((3, 3), numpy.array([d['h1e'], d['h1e']]), d['chol'], float(d['ecore']))
    t = trial_mod.SingleDetTrial(s, d['psi'])
    out = {}
    replay(d, s, t, {}, monkeypatch, est_extra={'back_propagated': {'tau_bp': 0.025, 'one_rdm': True}}, out=out)
    est = out['afqmc'].estimators.estimators['back_prop']
    numpy.testing.assert_allclose(numpy.array(est.denominator), d['bp_denominator'], rtol=1e-9)
    numpy.testing.assert_allclose(numpy.array(est.one_rdm), d['bp_one_rdm'], rtol=1e-8, atol=1e-10)
    assert abs(est.rdm()[11, 0, 1, 3].real - (-0.121883381144845)) < 1e-9


def test_default_output_file_is_the_reference_s(golden, standin, monkeypatch, tmp_path):
    """estimators/handler.py:60-71: estimates.<index>.h5 in the working directory unless told otherwise."""
    monkeypatch.delenv('AFQ_ESTIMATES_FILE', raising=False)
    monkeypatch.chdir(tmp_path)
    d = golden('traj_hubbard_c1.npz')
    s = systems.Hubbard(4, 4, 8, 8, float(d['U']))
    t = trial_mod.SingleDetTrial(s, d['psi'], name='UHF')
    replay(d, s, t, {'hubbard_stratonovich': 'continuous'}, monkeypatch)
    assert os.path.isfile('estimates.0.h5')
    data = pio.extract_mixed_estimates('estimates.0.h5')
    assert len(data['ETotal']) == int(d['nblocks'])
    numpy.testing.assert_allclose(data['ETotal'], d['blocks'][:, 5].real, rtol=1e-8)
    # opt-out
    from pauxy_amd.estimators.handler import Estimators
    from pauxy_amd.qmc.options import QMCOpts
    est = Estimators({'write_file': False}, True, QMCOpts({}, s), s, t, None)
    assert est.filename is None and est.basename == 'estimates'


def test_local_energy_by_name_needs_no_device_argument(golden, standin):
    """pauxy.estimators.mixed.local_energy(system, G, Ghalf) (mixed.py:383-385): call-compatible."""
    shell, comm = dropin_checks.build_like_the_driver(golden('traj_hubbard_c1.npz'), {'write_file': False})
    w0 = shell.psi.walkers[0]
    want = w0.local_energy(shell.system)
    before = shell.psi.dev.get(9, 0, 1).copy()                             # F_G of walker 0
    got = local_energy(shell.system, w0.G, Ghalf=w0.Gmod)
    numpy.testing.assert_allclose(numpy.array(got), numpy.array(want), rtol=1e-12)
    assert numpy.array_equal(shell.psi.dev.get(9, 0, 1), before)           # the population was not touched
    other = systems.Hubbard(2, 2, 2, 2, 4.0)
    with pytest.raises(ValueError, match='no device context'):
        local_energy(other, w0.G)
    release_context(shell.system, shell.trial)
