"""Multi-rank population control on CPU: 2 processes, gloo backend.

Parity definition for N ranks: the N-rank run must equal the
1-rank oracle with N*nw walkers given the same comb uniform.  The device is
replaced by a numpy stand-in with the same method surface (there is no GPU
here); the code under test is the rank logic: all-gather of weights, the comb
decided identically on every rank, packed walker send/recv, weight reset."""
import ctypes
import os
import socket

import numpy
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import afqmc_ref as ref
from pauxy_amd import _lib as L
from pauxy_amd.comm import TorchComm, FakeComm
from pauxy_amd.walkers.handler import pop_control_distributed, comb_parent_ix, comb_pairs, Walkers

NW, M, NE = 4, 3, 2


class NumpyDevice(object):
    """Stand-in for AfqDevice (tests only): same packed layout as afq_walker_pack."""
    buffer_device = 'cpu'           # walkers live in host memory: WalkerTransport needs no staging

    def __init__(self, phi, weight):
        self.phi = phi.copy()
        self.weight = weight.copy()
        self.unscaled = weight.copy()
        self.ot = numpy.ones(len(weight), dtype=complex)

    def get(self, field):
        assert field == L.F_WEIGHT
        return self.weight.copy()

    def scale_weights(self, scale):
        self.unscaled = self.weight.copy()
        self.weight = self.weight / scale

    def reset_weights(self):
        self.weight[:] = 1.0

    def copy_walker(self, src, dst):
        self.phi[dst] = self.phi[src]
        self.unscaled[dst] = self.unscaled[src]
        self.ot[dst] = self.ot[src]

    def pack_bytes(self):
        return 16 * (M * NE + 4) + 8 * 4

    def _flat(self, iw):
        return numpy.concatenate([self.phi[iw].ravel().view(numpy.float64),
                                  numpy.array([self.ot[iw]]).view(numpy.float64), numpy.zeros(6),
                                  [self.unscaled[iw], 0.0, self.weight[iw], 0.0]])

    def pack(self, iw, ptr):
        buf = numpy.ascontiguousarray(self._flat(iw))
        ctypes.memmove(ptr, buf.ctypes.data, buf.nbytes)

    def unpack(self, iw, ptr):
        buf = numpy.empty(self.pack_bytes() // 8)
        ctypes.memmove(buf.ctypes.data, ptr, buf.nbytes)
        n = 2 * M * NE
        self.phi[iw] = buf[:n].view(numpy.complex128).reshape(M, NE)
        self.ot[iw] = buf[n:n + 2].view(numpy.complex128)[0]
        self.unscaled[iw] = buf[n + 8]
        self.weight[iw] = buf[n + 10]

    def sync(self):
        pass

    # use_log_shift surface (afq_log_ovlp_sums / afq_set_log_shift)
    def log_ovlp_sums(self):
        return numpy.array([numpy.abs(self.ot).sum(), numpy.abs(self.detR).sum(), numpy.abs(self.log_detR).sum()])

    def set_log_shift(self, on, log_shift=0.0, detR_shift=0.0):
        self.shifts = (bool(on), log_shift, detR_shift)


class ShiftHost(object):
    """The attributes Walkers.update_log_ovlp touches."""

    def __init__(self, dev, ntot):
        self.dev, self.ntot_walkers, self.shift_counter = dev, ntot, 1
        self.log_shift = self.detR_shift = self.log_detR_shift = 0.0


def shift_population(k):
    rng = numpy.random.RandomState(40 + k)
    return (rng.rand(2 * NW) * numpy.exp(1j * rng.rand(2 * NW)) * 1e-3, rng.rand(2 * NW) * 5.0, rng.randn(2 * NW))


def population(seed=5):
    rng = numpy.random.RandomState(seed)
    phi = rng.rand(2 * NW, M, NE) + 1j * rng.rand(2 * NW, M, NE)
    w = numpy.array([3.1, 2.5, 0.9, 1.4, 0.01, 0.03, 2.2, 0.04])
    return phi, w


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _work(rank, out)
    except Exception as e:          # surface failures instead of hanging the parent
        out.put((rank, repr(e)))
        raise
    dist.barrier()
    dist.destroy_process_group()


def _work(rank, out):
    comm = TorchComm()
    phi, w = population()
    dev = NumpyDevice(phi[rank * NW:(rank + 1) * NW], w[rank * NW:(rank + 1) * NW])
    numpy.random.seed(123)            # only rank 0's draw is used
    total, pix = pop_control_distributed(dev, comm, NW, 2 * NW)
    est = numpy.array([dev.unscaled.sum(), 0.0], dtype=numpy.complex128)
    red = numpy.zeros_like(est)
    comm.Reduce(est, red, root=0)
    # walkers/handler.py:456-475 over two ranks: two population controls' worth of running averages
    host = ShiftHost(dev, 2 * NW)
    for k in range(2):
        ot, detR, log_detR = shift_population(k)
        sl = slice(rank * NW, (rank + 1) * NW)
        dev.ot, dev.detR, dev.log_detR = ot[sl], detR[sl], log_detR[sl]
        Walkers.update_log_ovlp(host, comm)
    assert dev.shifts == (True, host.log_shift, host.detR_shift)
    out.put((rank, dev.phi, dev.weight, dev.unscaled, total, pix, red,
             (host.log_shift, host.detR_shift, host.log_detR_shift, host.shift_counter)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_comb_matches_single_rank_oracle():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for rr in res:
        assert len(rr) == 8, rr
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-rank oracle with the same uniform
    numpy.random.seed(123)
    r = numpy.random.random()
    phi, w = population()
    walkers = [dict(phi=phi[i].copy(), weight=w[i], unscaled_weight=w[i], ot=1.0, ovlp=1.0, hybrid_energy=0.0,
                    total_weight=0.0) for i in range(2 * NW)]
    pix = ref.pop_control(None, walkers, 2 * NW, r)
    assert pix.max() > 1 and (pix == 0).any()          # the case really clones and kills
    got_phi = numpy.concatenate([res[0][1], res[1][1]])
    assert numpy.array_equal(got_phi, numpy.array([x['phi'] for x in walkers]))
    assert numpy.array_equal(numpy.concatenate([res[0][2], res[1][2]]), numpy.ones(2 * NW))
    assert numpy.allclose(numpy.concatenate([res[0][3], res[1][3]]),
                          numpy.array([x['unscaled_weight'] for x in walkers]), rtol=0, atol=0)
    for rr in res:
        assert rr[4] == pytest.approx(w.sum(), rel=1e-15)
        assert numpy.array_equal(rr[5], pix)
        assert rr[6][0].real == pytest.approx(sum(x['unscaled_weight'] for x in walkers), rel=1e-14)
    # use_log_shift: both ranks hold the single-rank oracle's running averages of the combined population
    class Model(object):
        pass
    m = Model()
    for k in range(2):
        ot, detR, log_detR = shift_population(k)
        ref.update_log_ovlp(m, [dict(ot=ot[i], detR=detR[i], log_detR=log_detR[i]) for i in range(2 * NW)])
    for rr in res:
        assert rr[7][0] == pytest.approx(m.log_shift, rel=1e-13)
        assert rr[7][1] == pytest.approx(m.detR_shift, rel=1e-13)
        assert rr[7][2] == pytest.approx(m.log_detR_shift, rel=1e-13)
        assert rr[7][3] == 3
    # cross-rank clone really happened in this case
    assert any(c // NW != k // NW for c, k in comb_pairs(pix))


def test_comb_host_matches_oracle_and_quirks():
    w = numpy.array([3.0, 1e-9, 1e-9, 1.0 - 2e-9])
    a = comb_parent_ix(w / (w.sum() / 4), 4, 0.5)
    b = ref.comb_parent_ix(w / (w.sum() / 4), 4, 0.5)
    assert numpy.array_equal(a, b) and list(a) == [3, 0, 0, 1]
    assert comb_pairs(a) == [(0, 1)]                   # zip truncation, walkers/handler.py:301
    rng = numpy.random.RandomState(0)
    for _ in range(50):
        w = rng.rand(16) * rng.choice([0.01, 1.0, 5.0], 16)
        r = rng.rand()
        sc = w.sum() / 16
        assert numpy.array_equal(comb_parent_ix(w / sc, 16, r), ref.comb_parent_ix(w / sc, 16, r))


def test_fake_comm_surface():
    c = FakeComm()
    buf = numpy.zeros(3)
    c.Allgather(numpy.arange(3.0), buf)
    assert numpy.array_equal(buf, numpy.arange(3.0))
    assert c.bcast({'a': 1})['a'] == 1 and c.rank == 0 and c.size == 1


# ---- bring-up of the library-owned communicator: the agreement protocol of Walkers._init_device_comm on CPU ranks -------
class BringUpDevice(object):
    """Records what Walkers._init_device_comm asks of the device (no GPU: every call succeeds)."""

    def __init__(self):
        self.calls, self.window = [], True

    def comm_available(self):
        self.calls.append('available')
        return True

    def comm_unique_id(self):
        self.calls.append('unique_id')
        return bytes(range(1, 129))

    def comm_init(self, uid, rank, size):
        assert uid == bytes(range(1, 129))
        self.calls.append('init')

    def comm_set_transport(self, window):
        self.calls.append('transport:%d' % int(window))
        self.window = bool(window)

    def comm_init_ipc(self, rank, size, allgather):
        got = allgather(bytes([rank]) * 4)                      # the bootstrap all-gather the library would ask for
        assert got == b''.join(bytes([r]) * 4 for r in range(size))
        self.calls.append('init_ipc')

    def comm_probe(self):
        self.calls.append('probe')

    def comm_stats(self):
        return {'window': self.window}

    def comm_destroy(self):
        self.calls.append('destroy')


class BringUpHost(object):
    def __init__(self):
        self.dev, self.device_comm_kind = BringUpDevice(), ''


BRINGUP_CASES = [
    # AFQ_COMM_FAULT, expected (communicator up, kind), calls every rank must have made in this order
    ('', (True, 'rccl'), ['available', 'unique_id?', 'init', 'transport:1', 'probe']),
    ('rccl:avail:2', (True, 'ipc'), ['available', 'init_ipc', 'probe']),
    ('rccl:probe:1', (True, 'sendrecv'), ['available', 'unique_id?', 'init', 'transport:1', 'probe', 'transport:0', 'probe']),
    ('rccl:init:0,ipc:probe:2', (False, ''), ['available', 'unique_id?', 'init', 'destroy', 'init_ipc', 'probe', 'destroy']),
    ('rccl:avail:1,ipc:init:0', (False, ''), ['available', 'init_ipc', 'destroy']),
]


def _bringup_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = TorchComm()
        res = []
        for fault, _, _ in BRINGUP_CASES:
            os.environ.pop('AFQ_COMM_FAULT', None)
            host = BringUpHost()
            host._device_comm_fault = fault              # (what walkers: {device_comm_fault: ...} sets)
            ok, reason = Walkers._init_device_comm(host, comm, True)
            res.append((ok, host.device_comm_kind, host.dev.calls, reason))
        out.put((rank, res))
    except Exception as e:
        out.put((rank, repr(e)))
        raise
    dist.barrier()
    dist.destroy_process_group()


def test_device_comm_bring_up_is_agreed_by_all_ranks_whichever_rank_fails():
    """Walkers._init_device_comm over three gloo ranks with faults injected on single ranks (AFQ_COMM_FAULT): every rank
    ends on the same communicator (or on none), no rank enters ncclCommInitRank unless all can load the library, a
    failed candidate is torn down everywhere before the next one is tried, and the reasons are kept."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_bringup_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, per_case in res:
        assert isinstance(per_case, list), per_case
        for (fault, (ok_want, kind_want), calls_want), (ok, kind, calls, reason) in zip(BRINGUP_CASES, per_case):
            assert (ok, kind) == (ok_want, kind_want), (rank, fault, ok, kind, reason)
            want = [c for c in calls_want if not (c == 'unique_id?' and rank != 0)]
            want = [c.rstrip('?') for c in want]
            assert calls == want, (rank, fault, calls)
            if fault:
                assert fault.split(':')[0] in reason, (fault, reason)
