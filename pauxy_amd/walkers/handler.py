"""Walker population living in HBM, behind PAUXY's ``Walkers`` surface.

Mirrors pauxy/walkers/handler.py:19-164 (constructor, attributes),
:166-181 (``orthogonalise``), :225-338 (``pop_control`` + ``comb``) and the
per-walker objects of pauxy/walkers/single_det.py: ``psi.walkers[i]`` is a light
proxy whose attributes (``weight``, ``ot``, ``phi`` ...) read and write the
batched device arrays, and whose methods (``greens_function``, ``local_energy``,
``reortho`` ...) trigger ONE batched launch for the whole population and hand
back this walker's row.

Scalars (weight, ot, hybrid energy ...) are mirrored on the host so the
unchanged PAUXY driver can read ``w.weight`` inside its per-walker loop without
one device round trip per walker: the mirror is refreshed once after each
batched launch and flushed back before the next one if the driver wrote to it.
"""
import os
import sys
import weakref

import time

import numpy

from pauxy_amd import _lib as L
from pauxy_amd.utils import io as _io
from pauxy_amd.context import get_context, trial_psi, hidden

_SCALARS = {
    'weight': (L.F_WEIGHT, numpy.float64),
    'unscaled_weight': (L.F_UNSCALED_WEIGHT, numpy.float64),
    'ot': (L.F_OT, numpy.complex128),
    'hybrid_energy': (L.F_HYBRID_ENERGY, numpy.complex128),
    'phase': (L.F_PHASE, numpy.complex128),
    'detR': (L.F_DETR, numpy.float64),
    'eloc': (L.F_ELOC, numpy.complex128),
    'log_detR': (L.F_LOG_DETR, numpy.float64),
}


def comb_parent_ix(weights, target, r):
    """Comb teeth against cumulative weights (walkers/handler.py:269-286).  Tooth ic goes to the first
    walker iw with (ic + r) * step < cumsum(weights)[iw]; the reference finds it with a sequential scan,
    here every tooth is located by bisection of the SAME sequential numpy.cumsum (identical comparisons,
    identical decisions); teeth beyond the last cumulative weight are dropped, as in the scan."""
    weights = numpy.asarray(weights, dtype=numpy.float64)
    n = len(weights)
    cprobs = numpy.cumsum(weights)
    total_weight = cprobs[-1]                  # == sum(weights): same left-to-right additions
    step = total_weight / target
    teeth = (numpy.arange(int(target)) + r) * step
    iw = numpy.searchsorted(cprobs, teeth, side='right')
    return numpy.bincount(iw[iw < n], minlength=n).astype('i')


def comb_pairs(parent_ix):
    """zip(clone, kill) of walkers/handler.py:295-301 (truncating)."""
    kill = numpy.where(parent_ix == 0)[0]
    clone = numpy.where(parent_ix > 1)[0]
    return list(zip(clone.tolist(), kill.tolist()))


class WalkerView(object):
    """Proxy for walker ``i`` of a device-resident population (the attribute
    and method surface of pauxy/walkers/single_det.py:11-364 that the hot path,
    the driver and the estimators touch)."""

    def __init__(self, handler, i):
        # the population this walker is a row of, as a weak reference: the population lists its walkers, and the
        # reference's serialise (utils/misc.py:72-135) follows every __dict__ without a cycle guard
        object.__setattr__(self, '_href', weakref.ref(handler))
        object.__setattr__(self, '_i', i)
        object.__setattr__(self, '_pending', False)
        object.__setattr__(self, 'total_weight', 0.0)
        object.__setattr__(self, 'old_total_weight', 0.0)
        object.__setattr__(self, 'le_oratio', 1.0)
        object.__setattr__(self, 'field_configs', None)
        object.__setattr__(self, 'stack', None)
        object.__setattr__(self, 'alive', 1)

    @property
    def _h(self):
        return self._href()

    # -- scalar attributes through the host mirror
    def __getattr__(self, name):
        if name in _SCALARS:
            v = self._h._mirror(name)[self._i]
            return complex(v) if numpy.iscomplexobj(v) else float(v)
        if name == 'ovlp':
            return complex(self._h._mirror('ot')[self._i])
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in _SCALARS:
            self._h._mirror(name)[self._i] = value
            self._h._dirty.add(name)
        elif name == 'ovlp':
            self._h._mirror('ot')[self._i] = value
            self._h._dirty.add('ot')
        else:
            object.__setattr__(self, name, value)

    # use_log_shift: the same values for every walker (walkers/walker.py:49-52, handler.py:471-474)
    @property
    def log_shift(self):
        return self._h.log_shift

    @property
    def detR_shift(self):
        return self._h.detR_shift

    @property
    def log_detR_shift(self):
        return self._h.log_detR_shift

    @property
    def nup(self):
        return self._h.dev.na

    @property
    def ndown(self):
        return self._h.dev.nb

    @property
    def phi(self):
        return self._h.dev.get(L.F_PHI, self._i, 1)[0]

    @phi.setter
    def phi(self, value):
        self._h.dev.set(L.F_PHI, numpy.asarray(value, dtype=numpy.complex128), self._i)
        self._h.phi_version += 1

    @property
    def weights(self):
        """MultiDetWalker.weights (walkers/multi_det.py:226): conj(c_d) <D_d|phi> per determinant."""
        self._h._ensure_greens()
        return self._h.dev.det_weights()[self._i]

    @property
    def Gmod(self):
        self._h._ensure_greens()
        gh = self._h.dev.get(L.F_GHALF, self._i, 1)[0]
        na = self._h.dev.na
        return [gh[:na], gh[na:]]

    @property
    def G(self):
        self._h._ensure_greens(want_G=True)
        return self._h.dev.get(L.F_G, self._i, 1)[0]

    # -- methods (walkers/single_det.py)
    def greens_function(self, trial):
        """single_det.py:295-321: batched for the whole population, returns this walker's det."""
        return complex(self._h._ensure_greens()[self._i] * numpy.exp(-self._h.log_shift))     # :320

    def calc_overlap(self, trial):
        """single_det.py:170-199."""
        self._h._flush()
        return complex(self._h.dev.calc_overlap()[self._i] * numpy.exp(-self._h.log_shift))   # :192

    def local_energy(self, system, two_rdm=None, rchol=None, eri=None, UVT=None):
        """single_det.py:340-364 -> estimators/mixed.py:383-437."""
        E = self._h._ensure_energy()[self._i]
        return (complex(E[0]), complex(E[1]), complex(E[2]))

    def reortho(self, trial):
        """single_det.py:215-255."""
        return float(self._h._ensure_reortho()[self._i])

    def get_buffer(self):
        """Minimal state for transport (the reference ships every numeric
        attribute, walkers/walker.py:63-100; everything else here is derived)."""
        return numpy.concatenate([self.phi.ravel(), numpy.array(
            [self.weight, self.unscaled_weight, self.ot, self.hybrid_energy, self.phase, self.detR, self.eloc],
            dtype=numpy.complex128)])

    def set_buffer(self, buff):
        n = self._h.dev.M * (self._h.dev.na + self._h.dev.nb)
        self.phi = buff[:n].reshape(self._h.dev.M, -1)
        (self.weight, self.unscaled_weight) = (buff[n].real, buff[n + 1].real)
        (self.ot, self.hybrid_energy, self.phase) = (buff[n + 2], buff[n + 3], buff[n + 4])
        (self.detR, self.eloc) = (buff[n + 5].real, buff[n + 6])


class Walkers(object):
    """Drop-in for pauxy.walkers.handler.Walkers (single-determinant walkers)."""
    ctx = hidden()
    dev = hidden()
    comm = hidden()
    system = hidden()
    trial = hidden()
    _device_comm_fault = hidden()       # (test hook of the communicator bring-up, see _init_device_comm)

    def __init__(self, system, trial, qmc, walker_opts={}, verbose=False, comm=None, nprop_tot=None,
                 nbp=None, device_id=None):
        self.nwalkers = qmc.nwalkers
        self.ntot_walkers = qmc.ntot_walkers
        self.write_freq = walker_opts.get('write_freq', 0)
        self.write_file = walker_opts.get('write_file', 'restart.h5')
        self.use_log_shift = walker_opts.get('use_log_shift', False)
        self.read_file = walker_opts.get('read_file', None)
        self.write_restart = self.write_freq > 0
        self.comm = comm
        self.shift_counter = 1
        self.log_shift = self.detR_shift = self.log_detR_shift = 0.0
        if nbp is not None and nprop_tot is not None and nprop_tot != nbp:
            raise NotImplementedError("ITCF field history (nprop_tot != nbp) is not on the device path")
        self.walker_type = 'SD' if getattr(trial, 'ndets', 1) == 1 else 'MSD'     # walkers/handler.py:53-68
        if (self.walker_type == 'SD' and getattr(trial, 'name', '') == 'MultiSlater'
                and numpy.asarray(trial.psi).ndim == 3):
            trial.psi = trial.psi[0]                       # walkers/handler.py:61
        self.pcont_method = walker_opts.get('population_control', 'comb')
        if self.pcont_method != 'comb':
            raise NotImplementedError("only the comb population control is implemented")
        self.min_weight = walker_opts.get('min_weight', 0.1)
        self.max_weight = walker_opts.get('max_weight', 4.0)
        self.ctx = get_context(system, trial, device_id)
        self.dev = self.ctx.dev
        self.system, self.trial = system, trial
        self.dev.walkers_alloc(self.nwalkers)
        self.nw = self.nwalkers
        if self.use_log_shift and self.walker_type != 'SD':
            raise NotImplementedError("use_log_shift: single-determinant walkers only (as in the reference, "
                                      "walkers/multi_det.py has no shifts)")
        self.dev.set_log_shift(self.use_log_shift)
        # Population control on the device over the library-owned communicator whenever the ranks sit on GPUs
        # (_init_device_comm: RCCL collectives + peer windows for the walkers, with agreed fall-backs).  The
        # host-mediated path (pop_control_distributed) stays for CPU process groups (gloo) and as the last resort.
        # walkers: {device_comm: True / 'rccl' / 'ipc' / 'sendrecv' / False}; AFQ_DEVICE_COMM=0 forces the host path.
        self.device_comm, self.device_comm_error, self.device_comm_kind = False, '', ''
        self._device_comm_fault = walker_opts.get('device_comm_fault', None)
        opt = walker_opts.get('device_comm', None)
        want = opt if opt is not None else os.environ.get('AFQ_DEVICE_COMM', '1') != '0'
        forced = opt is not None and opt is not False        # tried whatever the driver's communicator sits on
        if (comm is not None and comm.size > 1 and want and
                (forced or (getattr(comm, 'device', None) is not None and comm.device.type == 'cuda'))):
            self.device_comm, self.device_comm_error = self._init_device_comm(comm, want)
        self.target_weight = qmc.ntot_walkers
        # host mirrors of the per-walker scalars
        self._host = {}
        self._valid = set()
        self._dirty = set()
        self.phi_version = 0
        self._greens_version = -1
        self._greens_has_G = False
        self._energy_version = -1
        self._reortho_version = -1
        self._det = None
        self._energy = None
        # initial population: trial.init, weight walker_opts['weight'] (walkers/walker.py:24-29)
        init = getattr(trial, 'init', None)
        if init is None:
            init = trial_psi(trial)
            init = init[0] if init.ndim == 3 else init
        init = numpy.asarray(init, dtype=numpy.complex128)
        phi0 = numpy.broadcast_to(init, (self.nw,) + init.shape).copy()
        self.dev.set(L.F_PHI, phi0)
        w0 = walker_opts.get('weight', 1.0)
        self.dev.set(L.F_WEIGHT, numpy.full(self.nw, w0))
        self.dev.set(L.F_UNSCALED_WEIGHT, numpy.full(self.nw, w0))
        ot = self.dev.calc_overlap()                       # single_det.py:65-67
        self.dev.set(L.F_OT, ot)
        self.walkers = [WalkerView(self, i) for i in range(self.nw)]
        self.nbp = nbp
        self._bp_pending = nbp is not None        # afq_bp_configure needs the propagator (BH1^H): deferred
        self.set_total_weight(qmc.ntot_walkers)
        # E_L of the initial walkers (single_det.py:86-92) is evaluated lazily on request
        self.buff_size = init.size + 7
        self.walker_buffer = numpy.zeros(self.buff_size, dtype=numpy.complex128)
        # restart files (walkers/handler.py:146-161): one dataset walker_<global index> = [weight, phase, ot, phi]
        rank = 0 if comm is None else comm.rank
        if self.write_restart and rank == 0:
            with _io.h5.File(self.write_file, 'w') as fh5:
                for i in range(self.ntot_walkers):
                    fh5.create_dataset('walker_%d' % i, (3 + init.size,), dtype=numpy.complex128)
        if self.read_file is not None:
            self.read_walkers(comm)

    # ------------------------------------------------------------ mirrors
    def _mirror(self, name):
        if name not in self._valid:
            field, _ = _SCALARS[name]
            self._host[name] = self.dev.get(field)
            self._valid.add(name)
        return self._host[name]

    def _flush(self):
        """Upload host-side writes before any device launch."""
        if self._bp_pending and self.ctx.propagator_set:
            # walkers/walker.py:43,55-58: field history + phi_old (the initial walker) per walker
            self.dev.bp_configure(self.nbp)
            self._bp_pending = False
        for name in list(self._dirty):
            self.dev.set(_SCALARS[name][0], self._host[name])
        self._dirty.clear()

    def _invalidate(self, *names):
        for n in (names or _SCALARS.keys()):
            if n in self._dirty:
                continue
            self._valid.discard(n)

    # ------------------------------------------------- batched lazy kernels
    def _ensure_greens(self, want_G=False):
        if self._greens_version != self.phi_version or (want_G and not self._greens_has_G):
            self._flush()
            self._det = self.dev.greens(want_G=want_G or self.dev.kind == 'ueg')
            self._greens_version = self.phi_version
            self._greens_has_G = want_G or self.dev.kind == 'ueg'
        return self._det

    def _ensure_energy(self):
        if self._energy_version != self.phi_version:
            self._ensure_greens()
            self._energy = self.dev.local_energy()
            self._energy_version = self.phi_version
        return self._energy

    def _ensure_reortho(self):
        if self._reortho_version != self.phi_version:
            self._flush()
            self._detR = self.dev.reortho()
            self.phi_version += 1
            self._reortho_version = self.phi_version
            self._invalidate('ot', 'detR', 'weight')
        return self._detR

    def _end_sweep(self):
        """Called by every once-per-step entry point of the driver (pop_control,
        orthogonalise, Mixed.update): whatever per-walker 'already propagated'
        marks are left from the sweep that just ended are dropped."""
        for w in self.walkers:
            w._pending = False

    # ---------------------------------------------------------- reference API
    def orthogonalise(self, trial, free_projection):
        """walkers/handler.py:166-181 (one batched Gram-Schmidt launch; the
        free-projection weight/phase update is applied on the device)."""
        self._end_sweep()
        self._reortho_version = -1
        self._ensure_reortho()

    def set_total_weight(self, total_weight):
        for w in self.walkers:
            w.total_weight = total_weight
            w.old_total_weight = total_weight
        self.total_weight = total_weight

    def copy_historic_wfn(self):
        pass

    def add_field_config(self, *a, **k):
        """walkers/handler.py:183-199: the reference appends the step's fields to every walker's FieldConfig from the
        host.  Here the history is written on the device by the weight-update kernel of the step itself
        (afq_bp_configure), so there is nothing for the driver to add."""
        return None

    def get_write_buffers(self):
        """[nw, 3 + M*(Na+Nb)] rows of (weight, phase, ot, phi) -- walkers/handler.py:432-435, batched."""
        self._end_sweep()
        self._flush()
        phi = self.dev.get(L.F_PHI).reshape(self.nw, -1)
        buf = numpy.empty((self.nw, 3 + phi.shape[1]), dtype=numpy.complex128)
        buf[:, 0] = self.dev.get(L.F_WEIGHT)
        buf[:, 1] = self.dev.get(L.F_PHASE)
        buf[:, 2] = self.dev.get(L.F_OT)
        buf[:, 3:] = phi
        return buf

    def get_write_buffer(self, i):
        return self.get_write_buffers()[i]

    def write_walkers(self, comm):
        """walkers/handler.py:444-455.  The reference writes through parallel HDF5 (driver='mpio'); here
        the ranks' rows are all-gathered (a few MB, once per ``write_freq`` steps) and rank 0 writes."""
        start = time.time()
        mine = self.get_write_buffers()
        size = 1 if comm is None else comm.size
        rank = 0 if comm is None else comm.rank
        if size > 1:
            everything = numpy.zeros((size,) + mine.shape, dtype=numpy.complex128)
            comm.Allgather(mine, everything)
            everything = everything.reshape((-1, mine.shape[1]))
        else:
            everything = mine
        if rank == 0:
            with _io.h5.File(self.write_file, 'r+') as fh5:
                for ix in range(everything.shape[0]):
                    fh5['walker_%d' % ix][:] = everything[ix]
            print(" # Writing walkers to file.")
            print(" # Time to write restart: {:13.8e} s".format(time.time() - start))

    def set_walkers_from_buffers(self, first, rows):
        """Rows of (weight, phase, ot, phi) into walkers first.. -- walkers/handler.py:437-442, batched."""
        rows = numpy.asarray(rows, dtype=numpy.complex128)
        n = rows.shape[0]
        if n == 0:
            return
        self._flush()
        M, nt = self.dev.M, self.dev.na + self.dev.nb
        self.dev.set(L.F_PHI, numpy.ascontiguousarray(rows[:, 3:]).reshape(n, M, nt), first)
        self.dev.set(L.F_WEIGHT, numpy.ascontiguousarray(rows[:, 0].real), first)
        self.dev.set(L.F_PHASE, numpy.ascontiguousarray(rows[:, 1]), first)
        self.dev.set(L.F_OT, numpy.ascontiguousarray(rows[:, 2]), first)
        self.phi_version += 1
        self._invalidate()

    def read_walkers(self, comm):
        """walkers/handler.py:477-485: walker i of this rank is dataset walker_<i + nw*rank>; walkers the
        file does not hold keep their initial state (the reference prints the same warning)."""
        rank = 0 if comm is None else comm.rank
        with _io.h5.File(self.read_file, 'r') as fh5:
            run_first, run = 0, []
            for i in range(self.nw):
                name = 'walker_%d' % (i + self.nw * rank)
                if name in fh5:
                    if not run:
                        run_first = i
                    run.append(fh5[name][:])
                else:
                    print(" # Could not read walker data from: %s" % self.read_file)
                    self.set_walkers_from_buffers(run_first, run)
                    run = []
            self.set_walkers_from_buffers(run_first, run)

    def recompute_greens_function(self, trial, time_slice=None):
        self._greens_version = -1
        self._ensure_greens()

    def pop_control(self, comm, fetch=True):
        """walkers/handler.py:225-338 (comb).  Size-1 communicator: one device
        launch.  Several ranks: all-gather of |weights|, identical comb on every
        rank from rank 0's uniform, point-to-point copies of the cloned walkers.
        ``fetch=False`` (single rank, batched loop): nothing is read back, the total
        weight stays on the device for the next weight cap."""
        self._end_sweep()
        if self.ntot_walkers == 1:
            return
        self._flush()
        if self.use_log_shift:
            self.update_log_ovlp(comm)
        size = 1 if comm is None else comm.size
        if size == 1 or self.device_comm:
            # single rank, or the collective of afq_comm_init: only rank 0 draws the comb uniform (handler.py:276)
            r = numpy.random.random() if (size == 1 or comm.rank == 0) else 0.0
            if not fetch:
                self.dev.popcontrol_comb(r, self.target_weight, fetch=False)
                self.phi_version += 1
                self._invalidate()
                return
            try:
                parent_ix, total = self.dev.popcontrol_comb(r, self.target_weight)
            except L.AfqError as e:
                if e.code == -6:
                    print("# Warning: total weight is below 1e-8.  Something is seriously wrong.")
                    sys.exit()
                raise
            self.last_parent_ix = parent_ix                # the global comb when several ranks took part
        else:
            total = self._pop_control_distributed(comm)
        self.set_total_weight(total)
        self.phi_version += 1
        self._invalidate()

    def _init_device_comm(self, comm, want=True):
        """Brings up the library-owned communicator on every rank, or on none.  Candidates in order (``want`` picks one:
        'rccl', 'sendrecv', 'ipc'; True tries all):
          rccl      ncclCommInitRank; RCCL all-gather / all-reduce, walkers through mapped peer windows (live slots only)
          sendrecv  the same communicator with the fixed-capacity ncclSend / ncclRecv transport
          ipc       no RCCL at all: every exchange through the mapped windows, this communicator as the bootstrap
        Nothing that can block on a peer (ncclCommInitRank, the probe) is entered before ALL ranks have agreed -- one
        Allreduce of a flag over the driver's communicator -- that they can enter it: a rank that cannot load librccl
        must not leave the others waiting inside the collective.  Every candidate ends with afq_comm_probe (a
        known-answer round through the all-gather, one full slot to and from every peer on the chosen transport, and
        the all-reduce) and a last agreement; any rank's failure sends every rank on to the next candidate, and finally
        to the host-mediated path with the reasons kept in ``device_comm_error``."""
        dev = self.dev

        def agree(ok):
            flags = numpy.zeros(1)
            comm.Allreduce(numpy.array([1.0 if ok else 0.0]), flags)
            return int(round(flags[0])) == comm.size

        def allgather_bytes(mine):
            send = numpy.frombuffer(mine, dtype=numpy.uint8).astype(numpy.float64)
            recv = numpy.zeros(comm.size * send.size)
            comm.Allgather(send, recv)
            return recv.astype(numpy.uint8).tobytes()

        def teardown():
            try:
                dev.comm_destroy()
            except L.AfqError:
                pass

        # Fault injection for the bring-up tests (tests/test_gpu_bench.py): AFQ_COMM_FAULT="rccl:avail:3,ipc:probe:3" makes
        # rank 3 report that it cannot load librccl, and fail the probe of the IPC candidate.  Stages: avail (rccl /
        # sendrecv only), init, probe.  The injected failure is raised where a real one would be -- AFTER the collective
        # calls of the stage, so the other ranks are never left inside one.
        # The hook is test infrastructure: it is honoured only on request -- walkers: {device_comm_fault: "..."} or the
        # environment variable TOGETHER with AFQ_ALLOW_FAULT_INJECTION=1 (a stray AFQ_COMM_FAULT alone changes nothing) -- and
        # rank 0 says on stderr that failures are being injected.
        faults = set()
        spec = getattr(self, '_device_comm_fault', None)
        if spec is None and os.environ.get('AFQ_ALLOW_FAULT_INJECTION', '0') == '1':
            spec = os.environ.get('AFQ_COMM_FAULT', '')
        if spec and comm.rank == 0:
            sys.stderr.write("pauxy_amd: communicator bring-up with INJECTED failures (%s): test mode\n" % spec)
        for item in (spec or '').split(','):
            part = item.strip().split(':')
            if len(part) == 3 and part[2].lstrip('-').isdigit():
                faults.add((part[0], part[1], int(part[2])))

        def fault(kind, stage):
            return 'injected fault (AFQ_COMM_FAULT %s:%s:%d)' % (kind, stage, comm.rank) if (kind, stage, comm.rank) in faults else ''

        candidates = [want] if want in ('rccl', 'sendrecv', 'ipc') else ['rccl', 'sendrecv', 'ipc']
        errors = []
        rccl_up = rccl_failed = False
        for kind in candidates:
            err = ''
            if kind in ('rccl', 'sendrecv'):
                if rccl_failed:
                    continue                      # the communicator itself did not come up: no point in its other transport
                if not rccl_up:
                    rccl_failed = True            # until ncclCommInitRank has succeeded everywhere
                    # 1. can every rank load librccl?  (local question, then agreement)
                    if not agree(dev.comm_available() and not fault(kind, 'avail')):
                        errors.append(kind + ': librccl is not loadable on every rank')
                        continue
                    # 2. rank 0's id to everybody; ncclCommInitRank is a blocking collective, entered by all or none
                    try:
                        uid = dev.comm_unique_id() if comm.rank == 0 else bytes(128)
                    except L.AfqError as e:
                        uid, err = bytes(128), str(e)
                    uid = comm.bcast(uid, root=0)
                    if uid == bytes(128):
                        errors.append(kind + ': ' + (err or 'rank 0 could not create the communicator id'))
                        continue
                    try:
                        dev.comm_init(uid, comm.rank, comm.size)
                    except L.AfqError as e:
                        err = str(e)
                    err = err or fault(kind, 'init')
                    if not agree(not err):
                        teardown()
                        errors.append(kind + ': ' + (err or 'ncclCommInitRank failed on another rank'))
                        continue
                    rccl_up, rccl_failed = True, False
                try:
                    dev.comm_set_transport(kind == 'rccl')
                except L.AfqError as e:
                    err = str(e)
            else:
                if rccl_up:
                    teardown()
                    rccl_up = False
                try:
                    dev.comm_init_ipc(comm.rank, comm.size, allgather_bytes)
                except L.AfqError as e:
                    err = str(e)
                err = err or fault(kind, 'init')
            if not agree(not err):
                errors.append(kind + ': ' + (err or 'set-up failed on another rank'))
                if kind == 'ipc':
                    teardown()
                continue
            # 3. known-answer probe through everything the first population control will use (collective)
            try:
                dev.comm_probe()
                if kind == 'rccl' and not dev.comm_stats()['window']:
                    err = 'peer windows could not be exported / mapped on every rank'     # (agreed inside the library)
            except L.AfqError as e:
                err = str(e)
            err = err or fault(kind, 'probe')
            if agree(not err):
                self.device_comm_kind = kind
                return True, '; '.join(errors)
            errors.append(kind + ': ' + (err or 'probe failed on another rank'))
            if kind == 'ipc':
                teardown()
        if rccl_up:
            teardown()
        reason = '; '.join(errors)
        if comm.rank == 0:
            print("# Warning: device communicator not used (%s); population control goes through the host." % reason)
        return False, reason

    def update_log_ovlp(self, comm):
        """walkers/handler.py:456-475: running averages of log <|ot|>, log <|detR|>, <|log_detR|> over the global
        population.  Three sums come back from the device (one small synchronising copy per population control,
        which this option therefore costs), the averages go back as handle state."""
        send = self.dev.log_ovlp_sums()
        if comm is not None and comm.size > 1:
            global_av = numpy.zeros(3, dtype=numpy.float64)
            comm.Allreduce(send, global_av)
        else:
            global_av = send
        log_shift = numpy.log(global_av[0] / self.ntot_walkers)
        detR_shift = numpy.log(global_av[1] / self.ntot_walkers)
        log_detR_shift = global_av[2] / self.ntot_walkers
        n, nm1 = self.shift_counter, self.shift_counter - 1
        self.log_shift = float((self.log_shift * nm1 + log_shift) / n)
        self.log_detR_shift = float((self.log_detR_shift * nm1 + log_detR_shift) / n)
        self.detR_shift = float((self.detR_shift * nm1 + detR_shift) / n)
        self.shift_counter += 1
        self.dev.set_log_shift(True, self.log_shift, self.detR_shift)

    def tune_exchange_capacity(self):
        """Device communicator.  Window transport (live slots only): grow the windows when the largest transfer seen
        reaches half their capacity (they start at nw slots per peer up to 512 walkers per rank -- no overflow possible --
        and at max(512, nw / 4) above).  Fixed-size ncclSend / ncclRecv transport: size the per-peer exchange slots from
        the largest transfer seen so far -- four times that + 8, at least 32, at most nw.  The capacity STARTS at nw (what a rank owns: no
        overflow before there is history) and shrinks only after 20 events (every slot of the capacity crosses the link at
        every event on this transport: nw slots of 160 KB to each of 7 peers are 290 MB per event at the bench sizes); an
        overflow aborts the run, spare slots only cost link time.  Called at block boundaries, right behind the block's host sync; every rank computes the same
        global comb, hence the same statistic and the same new capacity (send and receive sizes must agree)."""
        if not self.device_comm:
            return
        st = self.dev.comm_stats()
        if st['window']:
            # live slots only cross the link, so spare capacity costs memory, not time: the windows never shrink, and
            # they grow (to at most nw slots per peer, which cannot overflow) as soon as the largest transfer seen
            # reaches half the capacity -- above 512 walkers per rank they start at max(512, nw / 4) slots
            if st['capacity'] < self.nw and 2 * st['max_transfer'] >= st['capacity']:
                self.dev.comm_set_capacity(min(self.nw, max(4 * st['max_transfer'] + 8, 2 * st['capacity'])))
            return
        want = min(self.nw, max(32, 4 * st['max_transfer'] + 8))
        if want > st['capacity'] or (st['events'] >= 20 and want < 0.6 * st['capacity']):
            self.dev.comm_set_capacity(want)

    def _pop_control_distributed(self, comm):
        total, self.last_parent_ix = pop_control_distributed(self.dev, comm, self.nw, self.target_weight)
        return total


def pop_control_distributed(dev, comm, nw, target_weight):
    """walkers/handler.py:225-338 across ranks.  ``dev`` is the rank's AfqDevice
    (anything with get/scale_weights/copy_walker/pack/unpack/reset_weights).

    Two collectives per call: one all-gather carries every rank's |weights| plus rank 0's comb uniform
    (the reference's Allgather :232 and bcast :291), then every rank decides the identical global comb and
    the cloned walkers move in ONE batched exchange (all sends and receives of a rank posted together;
    the reference's per-pair Isend/Recv loop :303-331)."""
    weights = numpy.abs(dev.get(L.F_WEIGHT))
    r = numpy.random.random() if comm.rank == 0 else 0.0   # handler.py:276 (only rank 0 draws)
    mine = numpy.concatenate([weights, [r]])
    gathered = numpy.empty((nw + 1) * comm.size)
    comm.Allgather(mine, gathered)                         # handler.py:232
    gathered = gathered.reshape(comm.size, nw + 1)
    global_weights = numpy.ascontiguousarray(gathered[:, :nw]).reshape(-1)
    r = float(gathered[0, nw])
    total_weight = float(numpy.cumsum(global_weights)[-1])  # sum(global_weights), sequential
    if total_weight < 1e-8:
        if comm.rank == 0:
            print("# Warning: total weight is {:13.8e}.  Something is seriously wrong.".format(total_weight))
        sys.exit()
    scale = total_weight / target_weight
    dev.scale_weights(scale)                               # handler.py:244-246
    parent_ix = comb_parent_ix(global_weights / scale, target_weight, r)
    outgoing, incoming = {}, {}                            # peer rank -> local walker indices, in pair order
    for c, k in comb_pairs(parent_ix):
        src_rank, dst_rank = c // nw, k // nw
        if src_rank == dst_rank:
            if src_rank == comm.rank:
                dev.copy_walker(c % nw, k % nw)
        elif src_rank == comm.rank:
            outgoing.setdefault(dst_rank, []).append(c % nw)
        elif dst_rank == comm.rank:
            incoming.setdefault(src_rank, []).append(k % nw)
    WalkerTransport(dev, comm).exchange(outgoing, incoming)
    dev.reset_weights()                                    # handler.py:337-338
    return total_weight, parent_ix


class WalkerTransport(object):
    """Moves packed walkers (phi + scalars [+ back-propagation state], ``afq_walker_pack``) between ranks:
    per peer one contiguous buffer of packed walkers, all transfers of a pop-control posted as one batch of
    isend / irecv (RCCL send/recv over xGMI on the GPU node; host-staged for CPU backends)."""

    def __init__(self, dev, comm):
        import torch
        self.torch = torch
        self.dev = dev
        self.comm = comm
        self.nbytes = dev.pack_bytes()
        # pack / unpack work on the memory the walkers live in: the GPU of an AfqDevice (host memory for
        # the numpy stand-in of the CPU tests, which says so through ``buffer_device``)
        self.gpu = torch.device(getattr(dev, 'buffer_device', None) or ('cuda:%d' % dev.device_id))
        self.direct = self.gpu.type == 'cpu' or (getattr(comm, 'device', None) is not None and
                                                comm.device.type == 'cuda')

    def exchange(self, outgoing, incoming):
        """outgoing / incoming: {peer rank: [local walker index, ...]} in the global pair order (both sides
        enumerate the same comb, so the k-th packed walker sent to a peer is the k-th it expects)."""
        torch = self.torch
        per = self.nbytes // 8
        sbuf, rbuf = {}, {}
        for peer in sorted(outgoing):
            buf = torch.empty(len(outgoing[peer]) * per, dtype=torch.float64, device=self.gpu)
            for j, iw in enumerate(outgoing[peer]):
                self.dev.pack(iw, buf.data_ptr() + j * self.nbytes)
            sbuf[peer] = buf
        if sbuf:
            self.dev.sync()
        for peer in sorted(incoming):
            n = len(incoming[peer]) * per
            rbuf[peer] = torch.empty(n, dtype=torch.float64, device=self.gpu if self.direct else 'cpu')
        if not self.direct:
            sbuf = {p: b.cpu() for p, b in sbuf.items()}
        self.comm.exchange_tensors(sbuf, rbuf)
        for peer in sorted(incoming):
            buf = rbuf[peer] if self.direct else rbuf[peer].to(self.gpu)
            if buf.is_cuda:
                torch.cuda.current_stream(buf.device).synchronize()
            for j, iw in enumerate(incoming[peer]):
                self.dev.unpack(iw, buf.data_ptr() + j * self.nbytes)
        if incoming:
            self.dev.sync()


