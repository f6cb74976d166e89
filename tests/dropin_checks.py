"""Checks of the plug-in classes against what the GENUINE driver does to them (tests/golden/dropin_trace.json, recorded
by ``make_golden.py dropin`` while pauxy/qmc/afqmc.py drove these classes).  Shared by the CPU suite (classes over the
oracle stand-in) and the GPU suite (classes over libafqmc_hip.so)."""
import json
import os

import numpy

from pauxy_amd import systems, trial as trial_mod
from pauxy_amd.comm import FakeComm
from pauxy_amd.estimators.handler import Estimators
from pauxy_amd.propagation.continuous import get_propagator_driver
from pauxy_amd.qmc.options import QMCOpts
from pauxy_amd.walkers.handler import Walkers
from tests import serialise_walk

HERE = os.path.dirname(os.path.abspath(__file__))


def trace_fixture():
    with open(os.path.join(HERE, 'golden', 'dropin_trace.json')) as f:
        return json.load(f)


class DriverShell(object):
    """The attributes qmc/afqmc.py:86-198 gives the driver object before it serialises itself."""


def build_like_the_driver(d, est_opts=None):
    """qmc/afqmc.py:149-182 for BASELINE configs[0] (the run behind traj_hubbard_c1.npz): the three constructor calls
    in the driver's order with the driver's arguments, on this package's own system / trial classes."""
    s = systems.Hubbard(4, 4, 8, 8, float(d['U']))
    t = trial_mod.SingleDetTrial(s, d['psi'], name='UHF')
    return build_shell(s, t, {'timestep': 0.01, 'num_steps': 10, 'blocks': 10, 'rng_seed': 8, 'num_walkers': 10,
                              'pop_control_freq': 5}, {'hubbard_stratonovich': 'continuous'}, est_opts)


def build_shell(s, t, qmc_opts, prop_opts, est_opts=None, walker_opts=None):
    qmc = QMCOpts(qmc_opts, s)
    comm = FakeComm()
    shell = DriverShell()
    shell.verbosity = 0
    shell.root, shell.rank = True, 0
    shell.system, shell.qmc, shell.trial = s, qmc, t
    shell.propagators = get_propagator_driver(s, t, qmc, options=prop_opts, verbose=False)
    est = dict(est_opts or {'mixed': {'energy_eval_freq': 1}})
    est['stack_size'] = 1                                                        # afqmc.py:160
    shell.estimators = Estimators(est, True, qmc, s, t, shell.propagators.BT_BP, False)
    qmc.nwalkers = int(qmc.nwalkers / comm.size)
    qmc.ntot_walkers = qmc.nwalkers * comm.size
    shell.psi = Walkers(s, t, qmc, walker_opts=walker_opts or {}, verbose=False, nprop_tot=shell.estimators.nprop_tot,
                        nbp=shell.estimators.nbp, comm=comm)
    return shell, comm


def check_serialisable(shell, fresh=True):
    """AFQMC.__init__'s last act (afqmc.py:192-195): the walk terminates, the result dumps as JSON, and the three plug-in
    sub-trees are what the genuine serialise produced for them, key by key and kind by kind."""
    tree = serialise_walk.walk(shell)
    text = json.dumps(tree, sort_keys=False, indent=4)
    assert json.loads(text).keys() == tree.keys()
    want = trace_fixture()['cases']['hubbard_c1']['serialised']
    got = serialise_walk.kinds(tree)
    for key in ('propagators', 'estimators', 'psi'):
        if fresh:       # as built, like AFQMC.__init__ sees them
            assert got[key] == want[key], (key, _diff(got[key], want[key]))
        else:           # after they have worked: more mirrors may exist, nothing may have gone or changed kind
            lost = [x for x in _diff(got[key], want[key]) if not x.endswith(' extra')]
            assert not lost, (key, lost)
    shell.estimators.json_string = text
    shell.estimators.dump_metadata()
    return text


def _diff(a, b, path=''):
    if isinstance(a, dict) and isinstance(b, dict):
        out = []
        for k in sorted(set(a) | set(b)):
            if k not in a or k not in b:
                out.append(path + '/' + k + (' missing' if k not in a else ' extra'))
            else:
                out.extend(_diff(a[k], b[k], path + '/' + k))
        return out
    return [] if a == b else ['%s: %r != %r' % (path, a, b)]


def check_surface(shell):
    """Every attribute the reference's code read, wrote or called on a plug-in object exists on this build's object of
    the same role (hasattr for reads and calls; callable for calls; assignable for writes)."""
    trace = trace_fixture()['trace']
    objs = {'Propagator': shell.propagators, 'Continuous.propagator': shell.propagators.propagator,
            'Walkers': shell.psi, 'Walker': shell.psi.walkers[0], 'Estimators': shell.estimators,
            'Mixed': shell.estimators.estimators['mixed'], 'system': shell.system, 'trial': shell.trial,
            'BackPropagation': shell.estimators.estimators.get('back_prop')}
    assert set(trace) <= set(objs), set(trace) - set(objs)
    for label, ent in trace.items():
        obj = objs[label]
        if obj is None:                  # this shell was built without that estimator
            continue
        for name in ent['read'] + ent['called']:
            assert hasattr(obj, name), (label, name)
        for name in ent['called']:
            assert callable(getattr(obj, name)), (label, name)
        for name in ent['written']:
            setattr(obj, name, getattr(obj, name, ''))
    # and the call signatures the driver uses (afqmc.py:214-250), on the objects themselves
    w0 = shell.psi.walkers[0]
    e = w0.local_energy(shell.system, rchol=shell.trial._rchol, eri=shell.trial._eri, UVT=shell.trial._UVT)
    assert len(e) == 3 and numpy.isfinite(complex(e[0]).real)
    return e
