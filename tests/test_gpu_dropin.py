"""The drop-in boundary on the GPU: the checks of tests/test_dropin_cpu.py with the plug-in classes over
libafqmc_hip.so instead of the numpy stand-in -- the object graph the reference's ``serialise`` walks, the attribute
surface the genuine driver touched (tests/golden/dropin_trace.json), ``local_energy`` by the reference's signature and
the trial energies the driver asks for at set-up."""
import json

import numpy
import pytest

from oracle import afqmc_ref as ref
from pauxy_amd import systems, trial as trial_mod
from pauxy_amd.context import release_context
from pauxy_amd.estimators.mixed import local_energy
from pauxy_amd.utils import io as pio
from tests import dropin_checks

pytestmark = pytest.mark.gpu


def test_object_graph_survives_the_reference_serialise(golden, monkeypatch, tmp_path):
    monkeypatch.delenv('AFQ_ESTIMATES_FILE', raising=False)
    monkeypatch.chdir(tmp_path)
    shell, comm = dropin_checks.build_like_the_driver(golden('traj_hubbard_c1.npz'))
    text = dropin_checks.check_serialisable(shell)
    with pio.h5.File('estimates.0.h5', 'r') as f:
        meta = f['metadata'][()]
    meta = meta.decode() if isinstance(meta, bytes) else str(meta)
    assert json.loads(meta) == json.loads(text)
    # the same walk after the objects have worked: lazily created members must not bring handles into a __dict__
    shell.psi.orthogonalise(shell.trial, False)
    numpy.random.seed(3)
    for w in shell.psi.walkers:
        shell.propagators.propagate_walker(w, shell.system, shell.trial, 0.0)
    shell.psi.pop_control(comm)
    shell.estimators.update(shell.system, shell.qmc, shell.trial, shell.psi, 1, False)
    dropin_checks.check_serialisable(shell, fresh=False)
    release_context(shell.system, shell.trial)


def test_surface_the_genuine_driver_touches(golden):
    shell, comm = dropin_checks.build_like_the_driver(golden('traj_hubbard_c1.npz'), {'write_file': False})
    e = dropin_checks.check_surface(shell)
    d = golden('traj_hubbard_c1.npz')
    m = ref.RefModel('hubbard', 16, 8, 8, d['psi'], d['BH1'], d['mf_shift'], 0.01, U=float(d['U']), H1=d['T'])
    _, gh, G = m.greens(d['phi0'][0])
    numpy.testing.assert_allclose(numpy.array(e), numpy.array(m.local_energy(G, gh)), rtol=1e-10)
    release_context(shell.system, shell.trial)


def test_local_energy_by_name_needs_no_device_argument(golden):
    """pauxy.estimators.mixed.local_energy(system, G, Ghalf) (mixed.py:383-385) and the trial energy the driver asks
    for before anything else exists (qmc/afqmc.py:147), Generic system."""
    d = golden('traj_generic.npz')
    na, nb = [int(x) for x in d['nelec']]
    s = systems.Generic((na, nb), numpy.array([d['h1e'], d['h1e']]), d['chol'], float(d['ecore']))
    t = trial_mod.SingleDetTrial(s, d['psi'])
    H1 = numpy.array([d['h1e'], d['h1e']], dtype=complex)
    want = ref.local_energy_generic_cholesky_opt(H1, float(d['ecore']), t.G, t.GH, t._rchol, na, nb)
    t.calculate_energy(s)                              # uploads system + trial, evaluates on a scratch handle
    numpy.testing.assert_allclose([t.energy, t.e1b, t.e2b], numpy.array(want), rtol=1e-10)
    rng = numpy.random.RandomState(5)
    phi = d['psi'] + 0.1 * (rng.rand(*d['psi'].shape) + 1j * rng.rand(*d['psi'].shape))
    _, gh, G = ref.greens_function(phi, d['psi'], na, nb)
    want = ref.local_energy_generic_cholesky_opt(H1, float(d['ecore']), G, gh, t._rchol, na, nb)
    numpy.testing.assert_allclose(numpy.array(local_energy(s, G, Ghalf=gh)), numpy.array(want), rtol=1e-10)
    numpy.testing.assert_allclose(numpy.array(local_energy(s, G)), numpy.array(want), rtol=1e-10)     # full-G form
    with pytest.raises(ValueError, match='no device context'):
        local_energy(systems.Hubbard(2, 2, 2, 2, 4.0), G)
    release_context(s, t)


def test_multi_determinant_trial_energy(golden):
    """variational_energy_multi_det (estimators/mixed.py:292-343) through the device full-G energy, against the same sum
    evaluated with the oracle's full-G Cholesky energy."""
    d = golden('msd_ops.npz')
    tag = 'N_'                                          # the non-orthogonal expansion (get_random_nomsd)
    na, nb = [int(x) for x in d['nelec']]
    h1e = d['h1e']
    s = systems.Generic((na, nb), numpy.array([h1e, h1e]), d['chol'], float(d['ecore']))
    coeffs, psi = d[tag + 'coeffs'][:3], d[tag + 'psi'][:3]
    t = trial_mod.MultiDetTrial(s, (coeffs, psi))
    t.calculate_energy(s)
    H1 = numpy.array([h1e, h1e], dtype=complex)
    num, den = numpy.zeros(3, dtype=complex), 0.0
    for ci, Di in zip(coeffs, psi):
        for cj, Dj in zip(coeffs, psi):
            Oa = Di[:, :na].conj().T.dot(Dj[:, :na])
            Ob = Di[:, na:].conj().T.dot(Dj[:, na:])
            ov = numpy.linalg.det(Oa) * numpy.linalg.det(Ob)
            G = numpy.array([ref.gab_mod(Di[:, :na], Dj[:, :na])[0], ref.gab_mod(Di[:, na:], Dj[:, na:])[0]])
            e = numpy.array(ref.local_energy_generic_cholesky(H1, float(d['ecore']), G, d['chol']))
            num += ci.conj() * cj * ov * e
            den += ci.conj() * cj * ov
    numpy.testing.assert_allclose([t.energy, t.e1b, t.e2b], num / den, rtol=1e-9)
    release_context(s, t)


def _walk_and_dump(shell):
    from tests import serialise_walk
    tree = serialise_walk.walk(shell)
    assert json.loads(json.dumps(tree, indent=4)).keys() == tree.keys()
    return tree


def test_every_plugin_variant_survives_the_serialise_walk(golden):
    """The other objects the driver can be handed: the discrete Hirsch propagator with the back-propagated estimator,
    a Generic system with back-propagation and the mixed one-body RDM, the UEG -- before and after they have worked."""
    d = golden('traj_hubbard_c1.npz')
    s = systems.Hubbard(4, 4, 8, 8, float(d['U']))
    t = trial_mod.SingleDetTrial(s, d['psi'], name='UHF')
    qmc = {'timestep': 0.01, 'num_steps': 5, 'blocks': 2, 'num_walkers': 6, 'pop_control_freq': 5}
    variants = [(s, t, {'hubbard_stratonovich': 'discrete'},
                 {'mixed': {'energy_eval_freq': 1}, 'back_propagated': {'tau_bp': 0.04, 'one_rdm': True}}, {})]
    g = golden('traj_generic.npz')
    na, nb = [int(x) for x in g['nelec']]
    sg = systems.Generic((na, nb), numpy.array([g['h1e'], g['h1e']]), g['chol'], float(g['ecore']))
    tg = trial_mod.SingleDetTrial(sg, g['psi'])
    variants.append((sg, tg, {}, {'mixed': {'energy_eval_freq': 1, 'one_rdm': True},
                                  'back_propagated': {'tau_bp': 0.04, 'one_rdm': True}}, {'use_log_shift': True}))
    su = systems.UEG(2.44, 7, 7, 2.0)
    variants.append((su, trial_mod.hartree_fock_ueg(su), {}, {'mixed': {'energy_eval_freq': 1}}, {}))
    for (sys_, tr, prop, est, wopt) in variants:
        est = dict(est, write_file=False)
        shell, comm = dropin_checks.build_shell(sys_, tr, qmc, prop, est, wopt)
        tree = _walk_and_dump(shell)
        assert {'propagators', 'estimators', 'psi'} <= set(tree)
        numpy.random.seed(11)
        for step in range(1, 6):
            for w in shell.psi.walkers:
                if abs(w.weight) > 1e-8:
                    shell.propagators.propagate_walker(w, sys_, tr, 0.0)
            if step == 5:
                shell.psi.pop_control(comm)
            shell.estimators.update(sys_, shell.qmc, tr, shell.psi, step, False)
            shell.estimators.print_step(comm, comm.size, step)
        _walk_and_dump(shell)
        release_context(sys_, tr)
