"""Continuous Hubbard-Stratonovich propagator driver on the device.

Mirrors pauxy/propagation/continuous.py:10-318 (``Continuous``) and the three
system back-ends it wraps (propagation/generic.py:10-179 ``GenericContinuous``,
propagation/planewave.py:11-112 ``PlaneWave``, propagation/hubbard.py:346-480
``HubbardContinuous`` / ``HubbardContinuousSpin``): same constructor signature,
same attributes (``free_projection``, ``hybrid``, ``force_bias``, ``BT_BP``,
``propagator.BH1`` / ``.mf_shift`` / ``.mf_core``, ``nfb_trig``, ``nhe_trig``)
and the same ``propagate_walker(walker, system, trial, eshift)`` entry point.

``propagate_walker`` is what the unchanged driver calls once per live walker
(qmc/afqmc.py:231-234).  The first call of a sweep draws the auxiliary fields of
every live walker from numpy's global stream -- in walker order, exactly the
numbers the reference would draw at continuous.py:133 -- and launches ONE batched
device step for the whole population; the remaining calls of the sweep find
their walker already propagated and return immediately.
"""
import numpy

from pauxy_amd.context import get_context, hidden
from pauxy_amd.propagation import setup


class _SystemPropagator(object):
    """The ``Continuous.propagator`` member: constants + per-walker test hooks."""
    _dev = hidden()         # no back reference to the driver object: the reference's serialise has no cycle guard

    def __init__(self, dev, system, trial, qmc, options):
        self._dev = dev
        self.dt = qmc.dt
        self.sqrt_dt = qmc.dt ** 0.5
        self.isqrt_dt = 1j * self.sqrt_dt
        name = system.name
        if name == "Generic":
            self.BH1, self.mf_shift = setup.generic_propagator_arrays(system, trial, qmc.dt)
            self.mf_core = system.ecore + 0.5 * numpy.dot(self.mf_shift, self.mf_shift)   # generic.py:49
        elif name == "Hubbard":
            charge = options.get('charge_decomposition', True)                              # continuous.py:344
            self.charge = charge
            self.BH1, self.mf_shift = setup.hubbard_propagator_arrays(system, trial, qmc.dt, charge)
            self.mf_core = 0.5 * numpy.dot(self.mf_shift, self.mf_shift)                   # hubbard.py:384
        elif name == "UEG":
            self.BH1, self.mf_shift = setup.ueg_propagator_arrays(system, trial, qmc.dt)
            self.mf_core = 0
            self.num_vplus = system.nfields // 2
        else:
            raise NotImplementedError("no continuous propagator for system %r" % name)
        self.nstblz = qmc.nstblz
        self.ebound = (2.0 / self.dt) ** 0.5

    def construct_one_body_propagator(self, system, dt):
        pass    # built in the constructor (reference: continuous.py:58)

    def construct_force_bias(self, system, walker, trial):
        """<system>.construct_force_bias(system, walker, trial) -> xbar[K] of this walker."""
        h = walker._h
        h._ensure_greens(want_G=h.dev.kind == 'ueg')
        return h.dev.force_bias()[walker._i]

    def construct_VHS(self, system, xshifted):
        """<system>.construct_VHS(system, xshifted) -> [M,M] (or [2,M,M] for spin HS)."""
        dev = self._dev
        xs = numpy.zeros((dev.nw, dev.K), dtype=numpy.complex128)
        xs[0] = xshifted
        v = dev.vhs(xs)[0]
        return v if dev.nv == 2 else v[0]


class Continuous(object):
    ctx = hidden()
    dev = hidden()

    def __init__(self, system, trial, qmc, options={}, verbose=False, device_id=None):
        self.free_projection = options.get('free_projection', False)
        self.hybrid = options.get('hybrid', True)
        self.force_bias = options.get('force_bias', True)
        if options.get('stochastic_ri', False) or options.get('control_variate', False):
            raise NotImplementedError("stochastic_ri / control_variate are outside the device hot path")
        if self.free_projection:
            self.force_bias = False                          # continuous.py:30-33
        self.exp_nmax = options.get('expansion_order', 6)
        self.device_rng = options.get('device_rng', False)
        self.dt = qmc.dt
        self.sqrt_dt = qmc.dt ** 0.5
        self.isqrt_dt = 1j * self.sqrt_dt
        self.ctx = get_context(system, trial, device_id)
        self.dev = self.ctx.dev
        self.propagator = _SystemPropagator(self.dev, system, trial, qmc, options)
        self.dev.set_propagator(self.propagator.BH1, self.propagator.mf_shift, qmc.dt,
                                exp_order=self.exp_nmax, hybrid=self.hybrid, force_bias=self.force_bias,
                                free_projection=self.free_projection,
                                hubbard_spin=(system.name == "Hubbard" and not self.propagator.charge))
        self.ctx.propagator_set = True
        self.BT_BP = self.propagator.BH1
        self.nstblz = qmc.nstblz
        self.ebound = (2.0 / self.dt) ** 0.5
        self.verbose = verbose
        if self.device_rng:
            self.dev.rng_seed(options.get('rng_seed', getattr(qmc, 'rng_seed', 0) or 0),
                              options.get('rng_stream', self.dev.device_id))
        self.propagate_walker = (self.propagate_walker_free if self.free_projection
                                 else self.propagate_walker_phaseless)

    @property
    def nfb_trig(self):
        return int(self.dev.counters()[0])

    @property
    def nhe_trig(self):
        return int(self.dev.counters()[1])

    # ------------------------------------------------------------ batched step
    def propagate_walkers(self, psi, system, trial, eshift):
        """One propagation step for every live walker of ``psi`` (the batched
        form of continuous.py:232-262 / :175-200)."""
        psi._flush()
        dev = self.dev
        if self.device_rng:
            xi = None
        else:
            alive = numpy.abs(psi._mirror('weight')) > 1e-8          # qmc/afqmc.py:232
            xi = numpy.zeros((dev.nw, dev.K))
            for iw in numpy.nonzero(alive)[0]:
                xi[iw] = numpy.random.normal(0.0, 1.0, dev.K)         # continuous.py:133
        dev.propagate(xi, eshift)
        psi.phi_version += 1
        psi._invalidate('weight', 'ot', 'hybrid_energy', 'phase', 'eloc')

    def propagate_walkers_begin(self, psi):
        """The part of propagate_walkers that does not read the energy shift (device stream of fields only): everything
        up to the propagated Slater matrices.  propagate_walkers_finish(psi, eshift) completes the step."""
        if not self.device_rng:
            raise NotImplementedError("split step: device stream of auxiliary fields only")
        psi._flush()
        self.dev.propagate_begin(None)

    def propagate_walkers_finish(self, psi, eshift):
        self.dev.propagate_finish(eshift)
        psi.phi_version += 1
        psi._invalidate('weight', 'ot', 'hybrid_energy', 'phase', 'eloc')

    def _propagate_walker(self, walker, system, trial, eshift):
        if walker._pending:
            walker._pending = False
            return
        psi = walker._h
        alive = numpy.abs(psi._mirror('weight')) > 1e-8
        self.propagate_walkers(psi, system, trial, eshift)
        for iw in numpy.nonzero(alive)[0]:
            psi.walkers[iw]._pending = (iw != walker._i)

    def propagate_walker_phaseless(self, walker, system, trial, eshift):
        self._propagate_walker(walker, system, trial, eshift)

    def propagate_walker_free(self, walker, system, trial, eshift):
        self._propagate_walker(walker, system, trial, eshift)


def get_propagator_driver(system, trial, qmc, options={}, verbose=False):
    """pauxy/propagation/utils.py:8-13 for continuous HS transformations."""
    hs = options.get('hubbard_stratonovich', 'continuous')
    if 'discrete' in hs:                                       # propagation/utils.py:10-11,34-37
        if system.name != "Hubbard":
            raise NotImplementedError("discrete HS transformation: Hubbard model only")
        from pauxy_amd.propagation.hubbard import Hirsch
        return Hirsch(system, trial, qmc, options=options, verbose=verbose)
    return Continuous(system, trial, qmc, options=options, verbose=verbose)
