"""Mixed estimator on the device, behind PAUXY's ``Mixed`` surface.

Mirrors pauxy/estimators/mixed.py:33-371 (``Mixed``: ``update``, ``print_step``,
``get_shift``, ``names``, ``estimates``) and the free function
``local_energy`` (:383-437).  ``update`` accumulates the ten estimators of
:460-469 with one device launch (plus the batched Green's function + local
energy launches on energy steps) instead of a Python loop over walkers.

Output: the per-block rows are kept in ``self.blocks`` (and printed when
``verbose``); when the container was given a file name the root rank also pushes
them to ``basic/energies/<block>`` with ``basic/headers`` (mixed.py:278,368-371;
layout in pauxy_amd/estimators/utils.py).
"""
import time

import numpy


class _Enum(dict):
    __getattr__ = dict.__getitem__


def get_estimator_enum(thermal=False):
    keys = ['uweight', 'weight', 'enumer', 'edenom', 'eproj', 'e1b', 'e2b', 'ehyb', 'ovlp', 'time']
    return _Enum((k, v) for v, k in enumerate(keys))


class Mixed(object):
    def __init__(self, mixed, system, root, filename, qmc, trial, dtype=complex):
        self.eval_energy = mixed.get('evaluate_energy', True)
        self.calc_one_rdm = mixed.get('one_rdm', False)
        if mixed.get('two_rdm', None) is not None:
            raise NotImplementedError("two_rdm accumulation is outside the device hot path")
        self.G = numpy.zeros((2, system.nbasis, system.nbasis))          # mixed.py:99
        self.rdm_acc = numpy.zeros_like(self.G)
        self.one_rdm = []                                               # per block, as pushed to 'one_rdm'
        self._rdm_armed = False
        self.energy_eval_freq = mixed.get('energy_eval_freq', None)
        if self.energy_eval_freq is None:
            self.energy_eval_freq = qmc.nsteps                      # mixed.py:78-80
        self.verbose = mixed.get('verbose', True)
        self.nsteps = qmc.nsteps
        self.header = ['Iteration', 'WeightFactor', 'Weight', 'ENumer', 'EDenom', 'ETotal', 'E1Body',
                       'E2Body', 'EHybrid', 'Overlap', 'Time']
        self.nreg = len(self.header[1:])
        self.dtype = dtype
        self.names = get_estimator_enum()
        self.estimates = numpy.zeros(self.nreg, dtype=dtype)
        self.estimates[self.names.time] = time.time()
        self.global_estimates = numpy.zeros(self.nreg, dtype=dtype)
        self.eshift = numpy.array([0, 0])
        self.key = {                                                    # mixed.py:113-126
            'Iteration': "Simulation iteration. iteration*dt = tau.",
            'WeightFactor': "Rescaling Factor from population control.",
            'Weight': "Total walker weight.",
            'E_num': "Numerator for projected energy estimator.",
            'E_denom': "Denominator for projected energy estimator.",
            'ETotal': "Projected energy estimator.",
            'E1Body': "Mixed one-body energy estimator.",
            'E2Body': "Mixed two-body energy estimator.",
            'EHybrid': "Hybrid energy.",
            'Overlap': "Walker average overlap.",
            'Nav': "Average number of electrons.",
            'Time': "Time per processor to complete one iteration.",
        }
        self.blocks = []
        self.root = root
        self.flush_every = mixed.get('flush_every', None)
        self.output = None
        if root and filename is not None:
            self.setup_output(filename)

    def update(self, system, qmc, trial, psi, step, free_projection=False):
        """mixed.py:133-233: importance-sampling branch (:210-225) or, when the propagator was
        built with free_projection, the complex wfac = weight*ot*phase accumulation of :151-175
        (the device handle knows which from afq_set_propagator)."""
        psi._end_sweep()
        psi._flush()
        dev = psi.dev
        self.arm_rdm(dev)
        do_energy = (step % self.energy_eval_freq == 0)
        if do_energy and not self.eval_energy:
            # E, T, V = 0 but the denominator still accumulates (mixed.py:215-221)
            dev.estimates_update(False)
            est = dev.estimates_get(zero=True)
            est[self.names.edenom] += est[self.names.weight]
        else:
            dev.estimates_update(do_energy)
            est = dev.estimates_get(zero=True)
        self.estimates[:self.names.time] += est[:self.names.time]
        if self.calc_one_rdm:
            self.rdm_acc += dev.estimates_rdm_get(zero=True)            # mixed.py:226-229
        if do_energy:
            psi._greens_version = psi.phi_version           # the launch refreshed Ghalf

    def arm_rdm(self, dev):
        """one_rdm: True -> the device accumulates weight * walker.G.real with every estimator update."""
        if self.calc_one_rdm and not self._rdm_armed:
            dev.estimates_rdm(True)
            self._rdm_armed = True

    def print_step(self, comm, nprocs, step, nsteps=None, free_projection=False):
        """mixed.py:235-289."""
        if step % self.nsteps != 0:
            return
        if nsteps is None:
            nsteps = self.nsteps
        es = self.estimates
        ns = self.names
        es[ns.time] = (time.time() - es[ns.time]) / (1 if getattr(comm, 'already_reduced', False) else nprocs)
        es[ns.uweight:ns.weight + 1] /= nsteps
        es[ns.ehyb:ns.time + 1] /= nsteps
        comm.Reduce(es, self.global_estimates, op=None)
        gs = self.global_estimates
        # torch.distributed has no rooted reduce worth its latency: TorchComm.Reduce leaves the sums on every
        # rank, so every rank derives the shift itself and the broadcast of mixed.py:273 is not needed
        everywhere = getattr(comm, 'reduce_is_allreduce', False)
        if comm.rank == 0 or everywhere:
            gs[ns.eproj] = gs[ns.enumer]
            # A block without an energy evaluation (energy_eval_freq larger than the block) has edenom = 0: the reference
            # divides all the same (mixed.py:268: a RuntimeWarning and a NaN row).  The row is the same NaN here -- the
            # file stays comparable with the reference's -- without the warning
            with numpy.errstate(divide='ignore', invalid='ignore'):
                gs[ns.eproj:ns.e2b + 1] = gs[ns.eproj:ns.e2b + 1] / gs[ns.edenom]
            gs[ns.ehyb] /= gs[ns.weight]
            gs[ns.ovlp] /= gs[ns.weight]
            eshift = numpy.array([gs[ns.ehyb], gs[ns.eproj]])
        else:
            eshift = numpy.array([0, 0])
        if not everywhere:
            eshift = comm.bcast(eshift, root=0)
        self.eshift = eshift
        if self.calc_one_rdm and comm.size > 1 and not getattr(comm, 'already_reduced', False):
            # the RDM sums are part of the reference's one estimates vector (mixed.py:261); here they are a second buffer
            red = numpy.zeros_like(self.rdm_acc)
            comm.Reduce(self.rdm_acc, red, op=None, root=0)
            self.rdm_acc[...] = red
        if comm.rank == 0:
            row = [step] + list(gs[:ns.time + 1])
            self.blocks.append(numpy.array(row))
            if self.calc_one_rdm:                               # mixed.py:279-283
                rdm = self.rdm_acc / nsteps / gs[ns.weight].real
                self.one_rdm.append(rdm)
            if self.output is not None:
                self.output.push(row, 'energies')               # mixed.py:278
                if self.calc_one_rdm:
                    self.output.push(self.one_rdm[-1], 'one_rdm')
                self.output.increment()
            if self.verbose:
                print(" ".join("{: .10e}".format(x) for x in numpy.array(row).real))
        self.zero()

    def get_shift(self, hybrid=True):
        """mixed.py:345-360."""
        return self.eshift[0].real if hybrid else self.eshift[1].real

    def projected_energy(self):
        return (self.estimates[self.names.enumer] / self.estimates[self.names.edenom]).real

    def zero(self):
        self.rdm_acc[:] = 0
        self.estimates[:] = 0
        self.global_estimates[:] = 0
        self.estimates[self.names.time] = time.time()

    def print_key(self, eol='', encode=False):
        """mixed.py:290-310: what the output columns are (same keys and texts, so logs read alike)."""
        header = eol + '# Explanation of output column headers:\n' + '# -------------------------------------' + eol
        print(header.encode('utf-8') if encode else header)
        for k, v in self.key.items():
            line = '# %s : %s' % (k, v) + eol
            print(line.encode('utf-8') if encode else line)

    def print_header(self, eol='', encode=False):
        print(" ".join("{:>17s}".format(x) for x in self.header) + eol)

    def setup_output(self, filename):
        """mixed.py:368-371."""
        from pauxy_amd.estimators.utils import H5EstimatorHelper
        from pauxy_amd.utils import io as _io
        with _io.h5.File(filename, 'a') as fh5:
            fh5['basic/headers'] = numpy.array(self.header).astype('S')
        self.output = H5EstimatorHelper(filename, 'basic', flush_every=self.flush_every)


def local_energy(system, G, Ghalf=None, two_rdm=None, rchol=None, eri=None, C0=None, ecoul0=None,
                 exxa0=None, exxb0=None, UVT=None, device=None):
    """pauxy.estimators.mixed.local_energy (mixed.py:383-437) for ONE Green's
    function, evaluated by the device energy kernels: ``Ghalf`` (half-rotated form) or, without it, the
    full ``G`` is used for Generic; ``G`` for Hubbard and the UEG.  Call-compatible with the reference:
    the device is the handle ``pauxy_amd.context`` already holds for ``system`` (the one the propagator,
    walkers and estimators of that system share); ``device=`` names another one.  The evaluation uses a
    scratch handle when the shared one carries a population (its walkers are not disturbed)."""
    from pauxy_amd import _lib as L
    from pauxy_amd import context
    if device is None:
        device = context.scratch_device(system)
    if device.nw < 1:
        device.walkers_alloc(1)
    if device.kind == 'ueg':
        device.set(L.F_G, numpy.asarray(G, dtype=numpy.complex128), 0)
    else:
        if Ghalf is None:
            # estimators/generic.py:398-434 (local_energy_generic_cholesky): full-G form
            E = device.local_energy_full_g(numpy.asarray(G, dtype=numpy.complex128)[None])[0]
            return (complex(E[0]), complex(E[1]), complex(E[2]))
        device.set(L.F_GHALF, numpy.concatenate([Ghalf[0], Ghalf[1]]).astype(numpy.complex128), 0)
    E = device.local_energy()[0]
    return (complex(E[0]), complex(E[1]), complex(E[2]))
