"""Discrete Hirsch Hubbard-Stratonovich propagator on the device.

Mirrors pauxy/propagation/hubbard.py:12-343 (``Hirsch``) for RHF/UHF-type single-determinant
trials with the classic single-site update or, with ``single_site_update: False``, the dynamic-force-bias
update of all sites at once (``two_body_direct``, :222-275) (constrained path, and ``free_projection: True`` =
``propagate_walker_free``, :303-343): same constructor signature, the same constants
(``bt2``, ``BT_BP``, ``gamma``, ``auxf``, ``aux_wfac``, ``delta``), ``hybrid == False`` and the
``propagate_walker(walker, system, trial, eshift)`` entry point the driver calls per walker.

Batched execution with the reference's random stream: the first ``propagate_walker`` call of a
sweep runs the kinetic half step of the whole population, reads the weights back (the
reference draws the site uniforms of a walker only if its weight is still non-zero after that
half step, hubbard.py:306-309), draws ``numpy.random.random()`` M times per surviving walker in
walker order, and launches the site loop and the second half step.
"""
import numpy
import scipy.linalg

from pauxy_amd import _lib as L
from pauxy_amd.context import get_context, hidden


class Hirsch(object):
    ctx = hidden()
    dev = hidden()

    def __init__(self, system, trial, qmc, options={}, verbose=False, device_id=None):
        if getattr(trial, 'type', '') == 'GHF' or getattr(trial, 'name', '') == 'multi_determinant':
            raise NotImplementedError("device Hirsch propagator: RHF/UHF-type single-determinant trials")
        self.single_site = options.get('single_site_update', True)                       # hubbard.py:49-57
        if options.get('ffts', False):
            raise NotImplementedError("k-space kinetic propagation is not on the device path")
        self.free_projection = options.get('free_projection', False)
        self.bt2 = numpy.array([scipy.linalg.expm(-0.5 * qmc.dt * system.T[0]),
                                scipy.linalg.expm(-0.5 * qmc.dt * system.T[1])])          # hubbard.py:36-37
        self.BT_BP = self.bt2
        self.nstblz = qmc.nstblz
        self.dt = qmc.dt
        self.hs_type = 'discrete'
        self.charge_decomp = options.get('charge_decomposition', False)
        if self.charge_decomp:                                                            # :66-82
            self.gamma = numpy.arccosh(numpy.exp(-0.5 * qmc.dt * system.U + 0j))
            self.auxf = numpy.array([[numpy.exp(self.gamma), numpy.exp(self.gamma)],
                                     [numpy.exp(-self.gamma), numpy.exp(-self.gamma)]])
            self.aux_wfac = numpy.exp(0.5 * qmc.dt * system.U) * numpy.array([numpy.exp(-self.gamma),
                                                                              numpy.exp(self.gamma)])
        else:
            self.gamma = numpy.arccosh(numpy.exp(0.5 * qmc.dt * system.U))
            self.auxf = numpy.array([[numpy.exp(self.gamma), numpy.exp(-self.gamma)],
                                     [numpy.exp(-self.gamma), numpy.exp(self.gamma)]])
            self.aux_wfac = numpy.array([1.0, 1.0])
        self.auxf = self.auxf * numpy.exp(-0.5 * qmc.dt * system.U)
        self.delta = self.auxf - 1
        self.hybrid = False
        self.device_rng = options.get('device_rng', False)
        self.ctx = get_context(system, trial, device_id)
        self.dev = self.ctx.dev
        self.dev.set_propagator_hirsch(self.bt2, qmc.dt, self.charge_decomp)
        self.ctx.propagator_set = True
        if self.device_rng:
            self.dev.rng_seed(options.get('rng_seed', getattr(qmc, 'rng_seed', 0) or 0),
                              options.get('rng_stream', self.dev.device_id))
        if not self.single_site:
            self.dev.hirsch_single_site(False)                                            # two_body_direct, :222-275
        if self.free_projection:                                                          # hubbard.py:84-89
            self.dev.hirsch_free_projection(True)
        self.propagate_walker = self.propagate_walker_free if self.free_projection else self.propagate_walker_constrained
        self.nfb_trig = 0
        self.nhe_trig = 0
        self.last_fields = None

    # ------------------------------------------------------------ batched step
    def propagate_walkers(self, psi, system, trial, eshift):
        """hubbard.py:285-312 for every walker with |weight| > 1e-8."""
        psi._flush()
        dev = self.dev
        if self.free_projection:
            # hubbard.py:303-343: M uniforms per propagated walker, in site order (:326)
            u = None
            if not self.device_rng:
                live0 = numpy.abs(psi._mirror('weight')) > 1e-8
                u = numpy.zeros((dev.nw, dev.M))
                for iw in numpy.nonzero(live0)[0]:
                    for i in range(dev.M):
                        u[iw, i] = numpy.random.random()
            self.last_fields = dev.propagate_hirsch_free(u, eshift, fetch_fields=not self.device_rng)
            psi.phi_version += 1
            psi._invalidate('weight', 'ot', 'phase')
            return
        if self.device_rng:
            dev.propagate_hirsch(eshift)
        else:
            dev.hirsch_kinetic()
            weight = dev.get(L.F_WEIGHT)
            live0 = numpy.abs(psi._mirror('weight')) > 1e-8
            u = numpy.zeros((dev.nw, dev.M))
            rows = [iw for iw in range(dev.nw) if live0[iw] and abs(weight[iw]) > 0]
            for iw in rows:
                for i in range(dev.M):
                    u[iw, i] = numpy.random.random()                       # hubbard.py:202
            fields, used = dev.hirsch_two_body(u)
            if any(used[iw] != dev.M for iw in rows):
                raise RuntimeError("a walker's field probabilities both vanished inside the site loop: the "
                                   "reference stops drawing uniforms for it (hubbard.py:222-224) and the batched "
                                   "draw above has diverged from its random stream")
            self.last_fields = fields
            dev.hirsch_finish(eshift)
        psi.phi_version += 1
        psi._invalidate('weight', 'ot')

    def propagate_walker_free(self, walker, system, trial, eshift=0):
        self.propagate_walker_constrained(walker, system, trial, eshift)

    def propagate_walker_constrained(self, walker, system, trial, eshift):
        if walker._pending:
            walker._pending = False
            return
        psi = walker._h
        alive = numpy.abs(psi._mirror('weight')) > 1e-8
        self.propagate_walkers(psi, system, trial, eshift)
        for iw in numpy.nonzero(alive)[0]:
            psi.walkers[iw]._pending = (iw != walker._i)
