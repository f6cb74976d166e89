"""One device context (AfqDevice + upload state) per (system, trial) pair.

PAUXY's driver builds the propagator, the estimators and the walkers as three
independent objects from the same ``system`` / ``trial`` (qmc/afqmc.py:149-182).
Here all three must talk to ONE library handle on one GPU; this module hands
that shared handle out, keyed by the identity of the system and trial objects.
"""
import os
import weakref

import numpy

from pauxy_amd.device import AfqDevice

_contexts = {}


class hidden(object):
    """Data descriptor that keeps a member OUT of the instance ``__dict__``.

    The genuine driver describes itself at the end of ``AFQMC.__init__`` (qmc/afqmc.py:193 -> utils/io.py:44-48 ->
    utils/misc.py:72-135 ``serialise``): it walks ``__dict__`` of every attribute that has one, without a cycle guard,
    and dumps the result as JSON.  The plug-in objects therefore keep in their ``__dict__`` only what the reference's
    own classes keep there (numbers, strings, arrays, option flags); handles to the device, the shared context, the
    communicator and back references (walker -> population) are class-level ``hidden()`` members, stored per
    instance in a WeakKeyDictionary.  Values must not refer back to their owner (an entry whose value keeps its key
    alive is never collected)."""

    def __init__(self):
        self._store = weakref.WeakKeyDictionary()
        self._name = '?'

    def __set_name__(self, owner, name):
        self._name = name

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        try:
            return self._store[obj]
        except KeyError:
            raise AttributeError(self._name)

    def __set__(self, obj, value):
        self._store[obj] = value

    def __delete__(self, obj):
        self._store.pop(obj, None)


def local_device_id():
    for key in ("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID"):
        if key in os.environ:
            return int(os.environ[key])
    return 0


def trial_psi(trial):
    """trial.psi is [ndet, M, ne] until the walker handler strips the leading
    axis for ndets == 1 (walkers/handler.py:61); accept both."""
    psi = numpy.asarray(trial.psi)
    if psi.ndim == 3 and psi.shape[0] == 1:
        psi = psi[0]
    return psi


class Context(object):
    def __init__(self, system, trial, device_id=None):
        self.system = system
        self.trial = trial
        self.dev = AfqDevice(local_device_id() if device_id is None else device_id)
        self._upload_system()
        psi = trial_psi(trial)
        if psi.ndim == 3:
            # multi-determinant expansion: per-determinant half-rotated Cholesky vectors, stacked as in
            # trial_wavefunction/multi_slater.py:370-409
            if system.name != "Generic":
                raise NotImplementedError("multi-determinant trials need a Generic system")
            self.dev.set_trial_multi(psi, numpy.asarray(trial.coeffs), numpy.asarray(trial._rchol))
        else:
            self.dev.set_trial(psi)
        self.propagator_set = False

    def _upload_system(self):
        s, t, dev = self.system, self.trial, self.dev
        na, nb = s.nup, s.ndown
        if s.name == "Generic":
            rchol = getattr(t, '_rchol', None)
            if rchol is None:
                raise ValueError("Generic system needs half-rotated Cholesky vectors (trial.half_rotate)")
            hs = numpy.asarray(s.hs_pot)
            if numpy.iscomplexobj(hs):
                if numpy.abs(hs.imag).max() > 0:
                    raise NotImplementedError("complex Cholesky vectors are not supported on the device path")
                hs = hs.real
            M = s.nbasis
            dev.set_system_generic(hs, numpy.asarray(rchol)[:(na + nb) * M], numpy.asarray(s.H1, dtype=complex),
                                   s.ecore, na, nb)
        elif s.name == "Hubbard":
            dev.set_system_hubbard(numpy.asarray(s.T, dtype=complex), s.U, na, nb)
        elif s.name == "UEG":
            H1diag = numpy.array([numpy.diag(s.H1[0]).real, numpy.diag(s.H1[1]).real])
            dev.set_system_ueg(s.iA, s.iB, s.ikpq_i, s.ikpq_kpq, s.ipmq_i, s.ipmq_pmq, s.vqvec, s.vol, H1diag,
                               s.ecore, na, nb)
        else:
            raise NotImplementedError("system %r has no device path" % s.name)

    def close(self):
        self.dev.close()


def get_context(system, trial, device_id=None):
    key = (id(system), id(trial))
    ctx = _contexts.get(key)
    if ctx is None or ctx.system is not system or ctx.trial is not trial:
        ctx = Context(system, trial, device_id)
        _contexts[key] = ctx
    return ctx


def release_context(system, trial):
    ctx = _contexts.pop((id(system), id(trial)), None)
    if ctx is not None:
        ctx.close()
    ent = _scratch.pop(id(system), None)
    if ent is not None:
        ent[1].close()


_scratch = {}


def release_scratch(system):
    """Closes the scratch handle of ``system`` (its second copy of the Hamiltonian on the device); the next
    ``local_energy(system, ...)`` without a device builds a new one.  ``trial.calculate_energy`` -- the one call the
    driver makes through it, at set-up -- releases it again, so a run does not carry two copies of hs_pot / rchol."""
    ent = _scratch.pop(id(system), None)
    if ent is not None:
        ent[1].close()


def scratch_device(system):
    """One-walker handle holding ``system`` for the reference's free functions that name no walker
    (``pauxy.estimators.mixed.local_energy(system, G, Ghalf)``, mixed.py:383-437): a second handle with the same
    uploads as the context the propagator / walkers / estimators of ``system`` share, so that evaluating an energy
    for a bare Green's function never touches the population's arrays or the library's caches of them."""
    ent = _scratch.get(id(system))
    if ent is not None and ent[0] is system:
        return ent[1].dev
    if ent is not None:                 # id() of a dead system object reused by this one: the old handle is nobody's
        _scratch.pop(id(system))[1].close()
    for ctx in list(_contexts.values()):
        if ctx.system is system:
            scratch = Context(system, ctx.trial, ctx.dev.device_id)
            scratch.dev.walkers_alloc(1)
            _scratch[id(system)] = (system, scratch)
            return scratch.dev
    raise ValueError("local_energy(system, ...): no device context holds this system yet -- build the propagator or "
                     "the walkers first (they upload it), or pass device=")
