"""Two ranks of the real driver on ONE GPU (two processes, gloo process group, both on cuda:0): the batched loop over
several ranks must reproduce the single-rank run with the same total population when both consume the same
auxiliary fields (N ranks == 1 rank with N * nw walkers).

What runs here is everything of the multi-rank path that a 1-GPU box allows: AFQMC.run_batched per rank on real device
handles, the global comb with walkers cloned across ranks (packed on the device, moved through the process group), the
block reductions, the energy-shift broadcast.  The second case asks for the library-owned RCCL communicator, which
RCCL refuses for two ranks on one GPU: every rank must then agree on the host-mediated path and give the same numbers."""
import os
import socket

import numpy
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
NW, NSTEPS, NBLOCKS = 6, 10, 2          # per rank; two ranks


def build():
    from pauxy_amd import systems, trial as trial_mod
    s = systems.synthetic_generic(12, 10, (3, 3), seed=3)
    t = trial_mod.rhf_trial_generic(s)
    return s, t




def options(nw_total, walkers=None):
    one_rdm = os.environ.get('AFQ_TEST_ONE_RDM', '1') == '1'
    o = {'qmc': {'timestep': 0.01, 'num_steps': NSTEPS, 'blocks': NBLOCKS, 'stabilise_freq': 5, 'pop_control_freq': 5,
                 'num_walkers': nw_total},
         'propagator': {'device_rng': False},
         'estimators': {'mixed': {'energy_eval_freq': 2, 'verbose': False, 'one_rdm': one_rdm}}}
    if walkers:
        o['walkers'] = walkers
    return o


def tables(nr=2):
    rng = numpy.random.RandomState(77)
    return rng.normal(size=(NSTEPS * NBLOCKS, nr * NW, 10)), rng.rand(NSTEPS * NBLOCKS)


class Feed(object):
    """numpy.random.{normal, random} stand-ins handing out the table rows of this rank's walkers in order."""

    def __init__(self, first, count, nr=2):
        xi, r = tables(nr)
        self.rows = iter(xi[:, first:first + count].reshape(-1, 10))
        self.r = iter(r[4::5])                 # one comb uniform per population control (every 5th step)

    def normal(self, loc, scale, size):
        return next(self.rows)

    def random(self):
        return next(self.r)


def drive(comm, nw_total, first, count, walkers=None, nr=2):
    from pauxy_amd.qmc.afqmc import AFQMC
    s, t = build()
    feed = Feed(first, count, nr)
    numpy.random.normal, numpy.random.random = feed.normal, feed.random
    afqmc = AFQMC(comm=comm, options=options(nw_total, walkers), system=s, trial=t)
    # weights spread over a decade so that the comb clones and kills, also across the two ranks
    w0 = numpy.exp(0.9 * numpy.random.RandomState(5).normal(size=nr * NW))[first:first + count]
    for i, w in enumerate(afqmc.psi.walkers):
        w.weight = w0[i]
    rec = dict(weight=[], ot=[], pix=[])

    def on_step(step, psi):
        rec['weight'].append(psi._mirror('weight').copy())
        rec['ot'].append(psi._mirror('ot').copy())
        if step % 5 == 0:
            rec['pix'].append(numpy.array(psi.last_parent_ix).copy())

    if os.environ.get('AFQ_TEST_NO_CALLBACK', '0') == '1':
        afqmc.run_batched()             # bench.py's mode: nothing read back at the comb, overlapped block boundaries
    else:
        afqmc.run_batched(on_step=on_step, fetch_popcontrol=True)
    mixed = afqmc.estimators.estimators['mixed']
    blocks = numpy.array(mixed.blocks) if comm is None or comm.rank == 0 else None
    rdm = numpy.array(mixed.one_rdm) if (comm is None or comm.rank == 0) and mixed.calc_one_rdm else None
    phi = numpy.array([w.phi for w in afqmc.psi.walkers])
    return dict(weight=numpy.array(rec['weight']), ot=numpy.array(rec['ot']), pix=numpy.array(rec['pix']),
                blocks=blocks, rdm=rdm, phi=phi, device_comm=bool(getattr(afqmc.psi, 'device_comm', False)),
                device_comm_error=getattr(afqmc.psi, 'device_comm_error', ''),
                device_comm_kind=getattr(afqmc.psi, 'device_comm_kind', ''),
                comm_stats=afqmc.psi.dev.comm_stats() if getattr(afqmc.psi, 'device_comm', False) else None)


def _worker(rank, port, walkers, q, nr=2):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(nr),
                          LOCAL_RANK='0')
        import torch
        import torch.distributed as dist
        from pauxy_amd.comm import TorchComm
        dist.init_process_group('gloo', rank=rank, world_size=nr)
        comm = TorchComm(device=torch.device('cpu'))
        out = drive(comm, nr * NW, rank * NW, NW, walkers, nr)
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:          # surface the failure instead of leaving the parent waiting
        q.put((rank, repr(e)))
        raise


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def many_ranks(nr, walkers=None):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, walkers, q, nr)) for r in range(nr)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
    for rank, out in res:
        assert isinstance(out, dict), (rank, out)
    return [r[1] for r in res]


def two_ranks(walkers=None):
    return tuple(many_ranks(2, walkers))


def single_rank(nr=2):
    import numpy.random as npr
    keep = npr.normal, npr.random
    try:
        return drive(None, nr * NW, 0, nr * NW, None, nr)
    finally:
        npr.normal, npr.random = keep


def compare(one, a, b):
    assert one['pix'].shape[0] == 4 and (one['pix'] > 1).any() and (one['pix'] == 0).any()
    assert numpy.array_equal(a['pix'], one['pix']) and numpy.array_equal(b['pix'], one['pix'])
    # walkers really crossed the rank boundary
    from pauxy_amd.walkers.handler import comb_pairs
    assert any(c // NW != k // NW for pix in one['pix'] for c, k in comb_pairs(pix))
    for key in ('weight', 'ot'):
        got = numpy.concatenate([a[key], b[key]], axis=1)
        assert got.shape == one[key].shape
        assert numpy.max(numpy.abs(got - one[key])) <= 1e-9 * max(1.0, numpy.max(numpy.abs(one[key]))), key
    got_phi = numpy.concatenate([a['phi'], b['phi']])
    assert numpy.max(numpy.abs(got_phi - one['phi'])) <= 1e-9
    # mixed one-body RDM (estimators/mixed.py:226-229,279-283): walker.G travelled with the clones, the sums were reduced
    if one['rdm'] is not None:
        assert a['rdm'].shape == one['rdm'].shape and a['rdm'].shape[0] == NBLOCKS
        assert numpy.max(numpy.abs(a['rdm'] - one['rdm'])) <= 1e-9 * numpy.max(numpy.abs(one['rdm']))
    assert a['blocks'].shape == one['blocks'].shape
    assert numpy.max(numpy.abs(a['blocks'][:, 1:10] - one['blocks'][:, 1:10])) <= 1e-9 * numpy.max(numpy.abs(one['blocks'][:, 1:10]))


def test_two_ranks_equal_one_rank_with_twice_the_walkers():
    one = single_rank()
    a, b = two_ranks()
    assert not a['device_comm'] and not b['device_comm']
    compare(one, a, b)


def test_rccl_refused_then_host_path():
    """walkers: {device_comm: 'rccl'} on a gloo group: afq_comm_init is attempted (after the ranks agreed that librccl
    loads), RCCL refuses two ranks on one GPU; both ranks must fall back together and still match the single-rank run."""
    one = single_rank()
    a, b = two_ranks({'device_comm': 'rccl'})
    assert not a['device_comm'] and not b['device_comm']
    assert a['device_comm_error'] and b['device_comm_error']
    compare(one, a, b)


def check_ipc(one, a, b):
    assert a['device_comm'] and b['device_comm']
    assert a['device_comm_kind'] == 'ipc' and b['device_comm_kind'] == 'ipc'
    compare(one, a, b)
    from pauxy_amd.walkers.handler import comb_pairs
    # traffic: every rank wrote exactly the walkers the global comb sent from it to the other rank, nothing else
    total = 0
    for rank, out in enumerate((a, b)):
        st = out['comm_stats']
        assert st['kind'] == 'ipc' and st['window'] == 1 and st['error'] == 0 and st['overflow'] == 0
        sent = sum(1 for pix in one['pix'] for c, k in comb_pairs(pix) if c // NW == rank and k // NW != rank)
        assert st['walkers_sent'] == sent
        per = 12 * 6                                        # M x (na + nb) of build()
        # a slot is phi + 6 scalars, plus Ghalf + its overlap when the Green's function is cached, plus walker.G with one_rdm
        assert 16 * (per + 6) * sent <= st['bytes_sent'] <= 16 * (2 * per + 7 + 2 * 12 * 12) * sent
        total += sent
    assert total > 0                                        # walkers did cross the process boundary


def test_device_comb_over_ipc_windows_two_processes():
    """The device path across REAL process boundaries on the one GPU of the box: walkers: {device_comm: 'ipc'} -- every
    rank maps the other's window (hipIpcGetMemHandle / hipIpcOpenMemHandle, bootstrap over the gloo group), weights
    all-gather, walker slots and the block reduction are kernels of one process writing into the other's memory, ordered
    by system-scope flags; nothing of the comb is read back by the host.  Must equal the single-rank run."""
    one = single_rank()
    a, b = two_ranks({'device_comm': 'ipc'})
    check_ipc(one, a, b)


def check_many(nr, bench_mode=False):
    one = single_rank(nr)
    outs = many_ranks(nr, {'device_comm': 'ipc'})
    from pauxy_amd.walkers.handler import comb_pairs
    assert all(o['device_comm'] and o['device_comm_kind'] == 'ipc' for o in outs)
    got_phi = numpy.concatenate([o['phi'] for o in outs])
    assert numpy.max(numpy.abs(got_phi - one['phi'])) <= 1e-9
    assert outs[0]['blocks'].shape == one['blocks'].shape
    assert numpy.max(numpy.abs(outs[0]['blocks'][:, 1:10] - one['blocks'][:, 1:10])) <= 1e-9 * numpy.max(numpy.abs(one['blocks'][:, 1:10]))
    if one['rdm'] is not None:
        assert numpy.max(numpy.abs(outs[0]['rdm'] - one['rdm'])) <= 1e-9 * numpy.max(numpy.abs(one['rdm']))
    for o in outs:
        st = o['comm_stats']
        assert st['kind'] == 'ipc' and st['window'] == 1 and st['error'] == 0 and st['overflow'] == 0 and st['size'] == nr
    if bench_mode:          # nothing was read back at the combs: only the totals can be checked
        assert sum(o['comm_stats']['walkers_sent'] for o in outs) > 0
        return
    assert (one['pix'] > 1).any() and (one['pix'] == 0).any()
    for o in outs:
        assert numpy.array_equal(o['pix'], one['pix'])
    for key in ('weight', 'ot'):
        got = numpy.concatenate([o[key] for o in outs], axis=1)
        assert got.shape == one[key].shape
        assert numpy.max(numpy.abs(got - one[key])) <= 1e-9 * max(1.0, numpy.max(numpy.abs(one[key]))), key
    total = 0
    for rank, o in enumerate(outs):
        sent = sum(1 for pix in one['pix'] for c, k in comb_pairs(pix) if c // NW == rank and k // NW != rank)
        assert o['comm_stats']['walkers_sent'] == sent
        total += sent
    assert total > 0


def test_device_comb_over_ipc_windows_four_processes():
    """More than one peer per rank: four processes on the one GPU, every rank maps three windows, the comb's walkers go to
    whichever rank the global plan names (one flag and one ticket counter per peer), the weights all-gather and the block
    reduction run over four windows.  Must equal the single-rank run with four times the walkers."""
    check_many(4)


def test_device_comb_over_ipc_windows_eight_processes():
    """The rank count of the node the scaling bench runs on: eight processes (here all on the one GPU), seven mapped
    windows per rank, 48 walkers in all.  Must equal the single-rank run with eight times the walkers, comb decisions
    and per-rank traffic included."""
    check_many(8)


def test_eight_processes_bench_mode_overlapped_block_boundary(monkeypatch):
    """Eight ranks driven exactly as bench.py --gpus 8 drives them: no per-step callback, nothing read back at the comb,
    estimator terms riding on the weight updates, the head of the next block queued before the window-reduced sums of the
    block are waited for."""
    monkeypatch.setenv('AFQ_TEST_ONE_RDM', '0')
    monkeypatch.setenv('AFQ_TEST_NO_CALLBACK', '1')
    check_many(8, bench_mode=True)


def _timeout_worker(rank, port, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK='0')
        import time
        import torch
        import torch.distributed as dist
        from pauxy_amd import _lib as L
        from pauxy_amd.comm import TorchComm
        from pauxy_amd.context import get_context
        dist.init_process_group('gloo', rank=rank, world_size=2)
        comm = TorchComm(device=torch.device('cpu'))
        s, t = build()
        dev = get_context(s, t).dev
        dev.walkers_alloc(NW)
        dev.set(L.F_PHI, numpy.broadcast_to(t.psi, (NW,) + t.psi.shape).copy())
        dev.set(L.F_OT, dev.calc_overlap())

        def allgather_bytes(mine):
            send = numpy.frombuffer(mine, dtype=numpy.uint8).astype(numpy.float64)
            recv = numpy.zeros(comm.size * send.size)
            comm.Allgather(send, recv)
            return recv.astype(numpy.uint8).tobytes()

        dev.comm_init_ipc(rank, 2, allgather_bytes)
        dev.comm_probe()
        dev.comm_set_timeout(0.5)
        out = {}
        if rank == 0:
            # the peer never enters this population control: the plan kernel must give up after the budget, not hang
            t0 = time.time()
            dev.popcontrol_comb(0.3, 2 * NW, fetch=False)
            dev.estimates_update(False)
            try:
                dev.estimates_get(zero=True)
                out['first'] = 'no error'
            except L.AfqError as e:
                out['first'] = e.code
            out['waited'] = time.time() - t0
        comm.barrier()
        # tear down on both ranks: the sticky flag belongs to the communicator that raised it
        dev.comm_destroy()
        dev.reset_weights()
        dev.estimates_update(False)
        dev.estimates_get(zero=True)                   # must not report the old communicator's error
        out['after_destroy'] = 'ok'
        # a fresh communicator on the same handles works, with clean statistics
        dev.comm_init_ipc(rank, 2, allgather_bytes)
        dev.comm_probe()
        dev.comm_set_timeout(30.0)
        pix, total = dev.popcontrol_comb(0.3, 2 * NW)
        dev.estimates_update(False)
        dev.estimates_get(zero=True)
        st = dev.comm_stats()
        out.update(events=st['events'], error=st['error'], total=total)
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, repr(e)))
        raise


def test_wait_budget_is_configurable_and_sticky_flags_die_with_their_communicator():
    """a rank whose peer never shows up gives up after the configured budget (afq_comm_set_timeout: 0.5 s here,
    300 s by default) with AFQ_ECOMM at the next host synchronisation; destroying the communicator clears the sticky
    flag, and a new communicator on the same handles starts clean."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_timeout_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = dict(q.get(timeout=300) for _ in procs)
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
    for rank in (0, 1):
        assert isinstance(res[rank], dict), res[rank]
        assert res[rank]['after_destroy'] == 'ok' and res[rank]['error'] == 0 and res[rank]['events'] == 1
        assert abs(res[rank]['total'] - 2 * NW) < 1e-9
    assert res[0]['first'] == -8                        # AFQ_ECOMM
    assert 0.4 < res[0]['waited'] < 20.0


def test_device_communicator_falls_through_to_ipc_windows():
    """walkers: {device_comm: True} tries RCCL first (refused here: two ranks on one GPU), then -- every rank agreeing at
    each step -- the IPC-window communicator, which works on one GPU as well."""
    one = single_rank()
    a, b = two_ranks({'device_comm': True})
    assert 'rccl' in a['device_comm_error']               # the reason the first candidate was dropped is kept
    check_ipc(one, a, b)


def test_ipc_windows_with_riding_estimator_terms(monkeypatch):
    """Without the one-body RDM the estimator terms of the plain steps ride on the weight update on every rank, and the
    block reduction runs through the windows."""
    monkeypatch.setenv('AFQ_TEST_ONE_RDM', '0')
    one = single_rank()
    a, b = two_ranks({'device_comm': 'ipc'})
    check_ipc(one, a, b)


def test_ipc_windows_bench_mode_overlapped_block_boundary(monkeypatch):
    """run_batched exactly as bench.py drives it on several GPUs -- no per-step callback, so nothing is read back at the
    comb and the head of the next block's first step is queued BEFORE the host waits for the block's (window-reduced)
    sums -- must leave the same walkers and block rows as the single-rank run."""
    monkeypatch.setenv('AFQ_TEST_ONE_RDM', '0')
    monkeypatch.setenv('AFQ_TEST_NO_CALLBACK', '1')
    one = single_rank()
    a, b = two_ranks({'device_comm': 'ipc'})
    assert a['device_comm_kind'] == 'ipc' and a['comm_stats']['walkers_sent'] + b['comm_stats']['walkers_sent'] > 0
    got_phi = numpy.concatenate([a['phi'], b['phi']])
    assert numpy.max(numpy.abs(got_phi - one['phi'])) <= 1e-9
    assert a['blocks'].shape == one['blocks'].shape
    assert numpy.max(numpy.abs(a['blocks'][:, 1:10] - one['blocks'][:, 1:10])) <= 1e-9 * numpy.max(numpy.abs(one['blocks'][:, 1:10]))


def test_two_ranks_with_estimator_terms_riding_on_the_weight_update(monkeypatch):
    """Without the one-body RDM the plain steps of run_batched take their estimator terms along with the weight update
    (afq_estimates_fuse_next) on every rank; the host-mediated comb and the block reduction must not notice."""
    monkeypatch.setenv('AFQ_TEST_ONE_RDM', '0')
    one = single_rank()
    a, b = two_ranks()
    compare(one, a, b)
