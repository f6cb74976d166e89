"""Step loop over the device objects.

This is the *caller* of the hot path, restated from pauxy/qmc/afqmc.py:200-255
so that tests and ``bench.py`` can drive Propagator / Walkers / Estimators in
exactly the order the reference driver does (orthogonalise -> propagate every
live walker -> weight cap -> population control -> estimators -> energy shift).
PAUXY's own ``AFQMC`` can drive the same three objects unchanged.

Two loops are provided:
  * ``run``          -- the reference's per-walker loop, line for line (parity mode);
  * ``run_batched``  -- the same sequence with every per-walker Python loop
                        replaced by one batched device call and no host round
                        trip inside a step except at population control.
"""
import json
import sys
import time

import numpy

from pauxy_amd import _lib
from pauxy_amd.comm import FakeComm
from pauxy_amd.estimators.handler import Estimators
from pauxy_amd.propagation.continuous import get_propagator_driver
from pauxy_amd.qmc.options import QMCOpts
from pauxy_amd.walkers.handler import Walkers


class AFQMC(object):
    def __init__(self, comm=None, options=None, system=None, trial=None, verbose=0):
        self.comm = comm if comm is not None else FakeComm()
        options = options or {}
        self.system = system
        self.trial = trial
        self.qmc = QMCOpts(options.get('qmc', options.get('qmc_options', {})), system)
        if self.qmc.rng_seed is not None:
            # qmc/utils.py:3-16: one global stream per rank, seed + rank
            numpy.random.seed(self.qmc.rng_seed + self.comm.rank)
        prop_opt = dict(options.get('propagator', {}))
        prop_opt.setdefault('hubbard_stratonovich', 'continuous')
        # device Philox stream = GLOBAL rank (the host path seeds seed + rank, qmc/utils.py:14): two ranks
        # must never draw the same auxiliary fields
        prop_opt.setdefault('rng_stream', self.comm.rank)
        self.propagators = get_propagator_driver(system, trial, self.qmc, options=prop_opt, verbose=verbose)
        est_opts = options.get('estimators', options.get('estimates', options.get('estimator', {})))
        self.estimators = Estimators(est_opts, self.comm.rank == 0, self.qmc, system, trial,
                                     self.propagators.BT_BP, verbose)
        self.qmc.nwalkers = max(1, int(self.qmc.nwalkers / self.comm.size))       # afqmc.py:167-176
        self.qmc.ntot_walkers = self.qmc.nwalkers * self.comm.size
        self.psi = Walkers(system, trial, self.qmc, walker_opts=options.get('walkers', {}), comm=self.comm,
                           nprop_tot=self.estimators.nprop_tot, nbp=self.estimators.nbp)       # afqmc.py:177-182
        self.setup_timers()
        if self.comm.rank == 0:                                                     # afqmc.py:192-195
            self.estimators.json_string = self.to_json()
            self.estimators.dump_metadata()

    def to_json(self):
        """Run description stored as ``metadata`` in the estimator file (utils/io.py:44-48 serialises
        the whole driver object; here: the scalar options of every component)."""
        def scalars(obj):
            out = {}
            for k, v in sorted(vars(obj).items()):
                if k.startswith('_'):
                    continue
                if isinstance(v, (bool, int, float, str)) or v is None:
                    out[k] = v
                elif isinstance(v, (numpy.integer, numpy.floating)):
                    out[k] = v.item()
                elif isinstance(v, complex):
                    out[k] = [v.real, v.imag]
                elif isinstance(v, (tuple, list)) and all(isinstance(x, (int, float)) for x in v):
                    out[k] = list(v)
                elif isinstance(v, numpy.ndarray) and v.size <= 64 and v.dtype.kind in 'iuf':
                    out[k] = [v.tolist()]                       # utils/misc.py serialise() nests arrays this way
            return out
        est = {'mixed': scalars(self.estimators.estimators['mixed'])}
        if 'back_prop' in self.estimators.estimators:
            est['back_prop'] = scalars(self.estimators.estimators['back_prop'])
        doc = {'system': scalars(self.system), 'qmc': scalars(self.qmc), 'trial': scalars(self.trial),
               'propagators': scalars(self.propagators),
               'estimators': {'filename': self.estimators.filename, 'estimators': est,
                              'nbp': self.estimators.nbp},
               'nprocs': self.comm.size}
        return json.dumps(doc, sort_keys=False, indent=4)

    def setup_timers(self):
        self.tortho = self.tprop = self.testim = self.tpopc = self.tstep = 0.0

    # ------------------------------------------------------------- parity loop
    def run(self, psi=None, comm=None, verbose=False, on_step=None):
        """qmc/afqmc.py:200-255."""
        comm = comm or self.comm
        if psi is not None:
            self.psi = psi
        self.setup_timers()
        self.psi.dev.set_weight_cap(0.0)          # this loop caps on the host mirrors, like the reference driver
        mixed = self.estimators.estimators['mixed']
        eshift = 0
        mixed.update(self.system, self.qmc, self.trial, self.psi, 0, self.propagators.free_projection)
        if verbose:
            mixed.print_step(comm, comm.size, 0, 1)
        for step in range(1, self.qmc.total_steps + 1):
            start_step = time.time()
            if step % self.qmc.nstblz == 0:
                start = time.time()
                self.psi.orthogonalise(self.trial, self.propagators.free_projection)
                self.tortho += time.time() - start
            start = time.time()
            for w in self.psi.walkers:
                if abs(w.weight) > 1e-8:
                    self.propagators.propagate_walker(w, self.system, self.trial, eshift)
                if (abs(w.weight) > w.total_weight * 0.10) and step > 1:
                    w.weight = w.total_weight * 0.10
            self.tprop += time.time() - start
            if step % self.qmc.npop_control == 0:
                start = time.time()
                self.psi.pop_control(comm)
                self.tpopc += time.time() - start
            start = time.time()
            self.estimators.update(self.system, self.qmc, self.trial, self.psi, step,
                                   self.propagators.free_projection)
            self.testim += time.time() - start
            if on_step is not None:
                on_step(step, self.psi)
            self.estimators.print_step(comm, comm.size, step)
            if self.psi.write_restart and step % self.psi.write_freq == 0:       # afqmc.py:249-250
                self.psi.write_walkers(comm)
            if step < self.qmc.neqlb:
                eshift = mixed.get_shift(self.propagators.hybrid)
            else:
                eshift += (mixed.get_shift() - eshift)
            self.tstep += time.time() - start_step

    # ------------------------------------------------------------ batched loop
    def step_batched_begin(self, step):
        """The head of step_batched(step, ...) that does not depend on the energy shift: re-orthogonalisation when due and
        the propagation up to the new Slater matrices.  run_batched enqueues it for the first step of a block BEFORE it
        waits for the estimators of the block just ended (which give that block's shift): the device then works through
        the block boundary instead of idling while the host reduces, prints and derives the shift."""
        psi, dev = self.psi, self.psi.dev
        if step % self.qmc.nstblz == 0:
            dev.reortho(fetch=False)
            psi.phi_version += 1
            psi._invalidate('ot', 'detR', 'weight')
        self.propagators.propagate_walkers_begin(psi)

    def step_batched(self, step, eshift, fetch_popcontrol=False, begun=False, publish=False):
        """One step of run() with batched device calls only (qmc/afqmc.py:223-246).  ``begun``: step_batched_begin(step)
        has been called already."""
        psi, dev = self.psi, self.psi.dev
        if not begun and step % self.qmc.nstblz == 0:
            dev.reortho(fetch=False)
            psi.phi_version += 1
            psi._invalidate('ot', 'detR', 'weight')
        single = self.comm is None or self.comm.size == 1
        hirsch = getattr(self.propagators, 'hs_type', '') == 'discrete'
        # the total weight of the last comb stays on the device whenever the comb itself ran there
        # (single rank, or the library-owned communicator of afq_comm_init)
        on_device = single or getattr(psi, 'device_comm', False)
        if not hirsch:
            # the weight cap of afqmc.py:235-236 rides on the weight-update kernel of the propagation
            dev.set_weight_cap(0.10 if step > 1 else 0.0, -1.0 if on_device else psi.total_weight)
        mixed = self.estimators.estimators['mixed']
        do_energy = step % mixed.energy_eval_freq == 0
        # a step that neither combs nor evaluates the energy nor ends a block: its estimator terms ride on its weight update
        ride = (getattr(self, 'ride_estimates', True) and not hirsch and step % self.qmc.npop_control != 0
                and not do_energy and step % self.qmc.nsteps != 0 and not mixed.calc_one_rdm)
        if ride:
            dev.estimates_fuse_next()
        if begun:
            self.propagators.propagate_walkers_finish(psi, eshift)
        else:
            self.propagators.propagate_walkers(psi, self.system, self.trial, eshift)
        if hirsch and step > 1:
            dev.cap_weights(0.10, -1.0 if on_device else psi.total_weight)
            psi._invalidate('weight')
        if step % self.qmc.npop_control == 0:
            psi._invalidate()
            psi.pop_control(self.comm, fetch=fetch_popcontrol or not on_device)
        if do_energy and not mixed.eval_energy:
            mixed.update(self.system, self.qmc, self.trial, psi, step, self.propagators.free_projection)
        elif not ride and publish:
            dev.estimates_update_publish(do_energy, zero=True)   # (the block's sums leave with this launch: run_batched)
        elif not ride:
            dev.estimates_update(do_energy)

    def run_batched(self, nsteps_total=None, first_step=1, eshift=0.0, on_step=None, fetch_popcontrol=False,
                    overlap_blocks=True):
        """Steps first_step .. first_step+nsteps_total-1 of run() with no host round trip inside a step
        (the estimator sums stay on the device until the end of a block); returns the final eshift.
        ``first_step == 1`` starts with the step-0 estimator pass of qmc/afqmc.py:214-221, like run().
        ``on_step(step, psi)`` (tests) is called after every step; ``fetch_popcontrol`` reads the comb
        decisions back (``psi.last_parent_ix``).  ``overlap_blocks``: at the end of a block the head of the next step
        (step_batched_begin) is enqueued before the host waits for the block's sums -- same numbers, the device does not
        idle while the host reduces / prints / derives the shift (one rank, or several ranks on the library-owned
        communicator, whose block reduction is queued on the stream like everything else; continuous fields from the
        device stream, mixed estimator only, no per-step callback; otherwise the plain order is used)."""
        mixed = self.estimators.estimators['mixed']
        others = [e for k, e in self.estimators.estimators.items() if k != 'mixed']
        n = self.qmc.total_steps if nsteps_total is None else nsteps_total
        ns = mixed.names
        dev = self.psi.dev
        fp = self.propagators.free_projection
        dcomm = getattr(self.psi, 'device_comm', False)
        if dcomm and not mixed.eval_energy:
            raise NotImplementedError("evaluate_energy: False with the device communicator")
        if dcomm:
            from pauxy_amd.comm import DeviceComm, ReducedComm
            block_comm, other_comm = ReducedComm(self.comm), DeviceComm(dev, self.comm)
        else:
            block_comm = other_comm = self.comm
        mixed.arm_rdm(dev)
        if first_step == 1:
            if dcomm:       # stays in the device accumulators and is reduced with the first block
                self.psi._end_sweep()
                self.psi._flush()
                dev.estimates_update(True)
            else:
                mixed.update(self.system, self.qmc, self.trial, self.psi, 0, fp)
        hirsch = getattr(self.propagators, 'hs_type', '') == 'discrete'
        overlap = (overlap_blocks and not others and (dcomm or self.comm.size == 1) and not hirsch and on_step is None
                   and getattr(self.propagators, 'device_rng', False) and mixed.eval_energy and not mixed.calc_one_rdm
                   and not self.psi.write_restart)
        begun = False
        try:
            for step in range(first_step, first_step + n):
                # one rank, a block ends here and the next step's head is enqueued before the host waits: the update's summation
                # launch hands the sums over itself (afq_estimates_update_publish)
                publish = (overlap and not dcomm and step % self.qmc.nsteps == 0 and step + 1 < first_step + n
                           and not (step % mixed.energy_eval_freq == 0 and not mixed.eval_energy))
                self.step_batched(step, eshift, fetch_popcontrol, begun=begun, publish=publish)
                begun = False
                for est in others:                      # back-propagation: estimators/handler.py:156-162
                    est.update(self.system, self.qmc, self.trial, self.psi, step, fp)
                if on_step is not None:
                    on_step(step, self.psi)
                if step % self.qmc.nsteps == 0:
                    if dcomm:
                        dev.estimates_allreduce()       # mixed.py:261 on the device, 20 doubles over RCCL
                    if overlap and step + 1 < first_step + n:
                        if not publish:
                            dev.estimates_get_begin(zero=True)
                        self.step_batched_begin(step + 1)
                        begun = True
                        est = dev.estimates_get_end()
                    else:
                        est = dev.estimates_get(zero=True)  # also reports a collapsed population (AFQ_EWEIGHT)
                    mixed.estimates[:ns.time] += est[:ns.time]
                    if mixed.calc_one_rdm:
                        mixed.rdm_acc += dev.estimates_rdm_get(zero=True)
                    mixed.print_step(block_comm, self.comm.size, step)
                    self.psi.tune_exchange_capacity()
                for est in others:
                    est.print_step(other_comm, self.comm.size, step)
                if self.psi.write_restart and step % self.psi.write_freq == 0:
                    self.psi.write_walkers(self.comm)
                if step < self.qmc.neqlb:
                    eshift = mixed.get_shift(self.propagators.hybrid)
                else:
                    eshift += (mixed.get_shift() - eshift)
        except _lib.AfqError as e:
            if e.code == _lib.AFQ_EWEIGHT:              # walkers/handler.py:236-241
                print("# Warning: total weight is below 1e-8.  Something is seriously wrong.")
                sys.exit()
            raise
        finally:
            dev.set_weight_cap(0.0)                     # the cap must not stay armed on the shared handle
        return eshift

    def finalise(self, verbose=False):
        """afqmc.py:257-275 prints timings; here: write whatever estimator blocks are still queued."""
        self.estimators.flush()
