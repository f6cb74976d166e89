"""SURVEY 8e parity definition on the device path: N ranks with the same host-drawn streams equal the ONE-rank
run with N x nw walkers (walkers/handler.py:225-338; the reference's own multi-rank run differs only in its
per-rank seeds, qmc/utils.py:14).

The box has one GPU and RCCL refuses two ranks on one device, so the ranks here are the handles of the
in-process communicator (afq_comm_init_local): the same prep / global-plan / pack / unpack kernels and slot
buffers the RCCL transport uses, with device-to-device copies in place of ncclAllGather / ncclSend / ncclRecv.
The RCCL calls themselves are exercised on a communicator of size 1."""
import numpy
import pytest

from oracle import afqmc_ref as ref
from pauxy_amd import _lib as L
from pauxy_amd import device as devmod
from tests.helpers import generic_model, make_device

pytestmark = pytest.mark.gpu
FIELDS = (L.F_PHI, L.F_WEIGHT, L.F_UNSCALED_WEIGHT, L.F_OT, L.F_HYBRID_ENERGY, L.F_PHASE, L.F_DETR, L.F_ELOC)


def start(golden, nranks, nw, seed=5, spread=1.0):
    d = golden('generic_ops.npz')
    model = generic_model(d, 'A_')
    rng = numpy.random.RandomState(seed)
    M, nt = model.M, model.na + model.nb
    ntot = nranks * nw
    phis = numpy.array([model.psi + 0.05 * (rng.rand(M, nt) + 1j * rng.rand(M, nt)) for _ in range(ntot)])
    ots = numpy.array([ref.calc_overlap(p, model.psi, model.na, model.nb) for p in phis])
    weights = numpy.exp(spread * rng.normal(size=ntot))
    one = make_device(model, ntot)
    ranks = [make_device(model, nw) for _ in range(nranks)]
    for dev, sl in [(one, slice(0, ntot))] + [(ranks[i], slice(i * nw, (i + 1) * nw)) for i in range(nranks)]:
        dev.set(L.F_PHI, phis[sl]); dev.set(L.F_OT, ots[sl]); dev.set(L.F_WEIGHT, weights[sl])
    devmod.comm_init_local(ranks)
    return model, one, ranks, rng


def gather(ranks, field):
    return numpy.concatenate([r.get(field) for r in ranks])


def same_population(one, ranks, exact=True):
    for f in FIELDS:
        a, b = one.get(f), gather(ranks, f)
        if exact:
            assert numpy.array_equal(a, b), f
        else:
            assert numpy.max(numpy.abs(a - b)) <= 1e-12 * max(1.0, numpy.max(numpy.abs(a))), f


def close_all(devs):
    for dv in devs:
        dv.close()


@pytest.mark.parametrize("nranks,nw", [(2, 6), (3, 4), (4, 16)])
def test_comb_across_ranks_equals_one_rank(golden, nranks, nw):
    model, one, ranks, rng = start(golden, nranks, nw)
    ntot = nranks * nw
    xi = rng.normal(size=(ntot, one.K))
    sent = [0] * nranks
    for step in range(3):
        # one step with a cached Green's function at its end (the cache has to travel with the clones),
        # then the comb; the next step consumes the cache
        one.propagate(xi, 0.1)
        for i, rk in enumerate(ranks):
            rk.propagate(xi[i * nw:(i + 1) * nw], 0.1)
        r = rng.rand()
        pix_one, tw_one = one.popcontrol_comb(r, ntot)
        pix, tw = devmod.popcontrol_comb_local(ranks, r, ntot)
        assert numpy.array_equal(pix, pix_one)
        assert tw == pytest.approx(tw_one, rel=1e-14)
        same_population(one, ranks, exact=(step == 0))     # later steps: the total weight differs in the last bit
        st = ranks[0].comm_stats()
        assert st['overflow'] == 0 and st['size'] == nranks and st['events'] == step + 1
        # count-aware exchange: every rank has written exactly the walkers the comb sent from it to other ranks -- bytes
        # moved == live slots x slot size (phi + Ghalf + cached overlap + 6 scalars, 16 bytes each), no empty slots
        for c_, k_ in zip(numpy.where(pix > 1)[0], numpy.where(pix == 0)[0]):
            if c_ // nw != k_ // nw:
                sent[c_ // nw] += 1
        for i, rk in enumerate(ranks):
            sti = rk.comm_stats()
            assert sti['window'] == 1 and sti['error'] == 0 and sti['kind'] == 'in-process'
            assert sti['walkers_sent'] == sent[i]
            assert sti['bytes_sent'] == sent[i] * 16 * (2 * one.M * (one.na + one.nb) + 7)
    pairs_cross = sum(1 for c, k in zip(numpy.where(pix > 1)[0], numpy.where(pix == 0)[0]) if c // nw != k // nw)
    assert ranks[0].comm_stats()['max_transfer'] >= (1 if pairs_cross else 0)
    close_all([one] + ranks)


@pytest.mark.parametrize("nw", [8, 48])
def test_walkers_did_change_rank(golden, nw):
    """The case the transport exists for: every heavy walker on rank 0, every dead one on rank 1 (48: more slots per
    peer than the pack / unpack grid has block rows, every row loops)."""
    nranks = 2
    model, one, ranks, rng = start(golden, nranks, nw)
    w = numpy.concatenate([numpy.full(nw, 3.0), numpy.full(nw, 1e-3)])
    one.set(L.F_WEIGHT, w); ranks[0].set(L.F_WEIGHT, w[:nw]); ranks[1].set(L.F_WEIGHT, w[nw:])
    for rk in ranks:
        rk.comm_set_capacity(nw)
    pix_one, _ = one.popcontrol_comb(0.37, nranks * nw)
    pix, _ = devmod.popcontrol_comb_local(ranks, 0.37, nranks * nw)
    assert numpy.array_equal(pix, pix_one)
    assert numpy.all(pix[nw:] == 0) and numpy.all(pix[:nw] == 2)
    same_population(one, ranks)
    assert ranks[0].comm_stats()['max_transfer'] == nw
    assert ranks[0].comm_stats()['walkers_sent'] == nw and ranks[1].comm_stats()['walkers_sent'] == 0
    # the walkers of rank 1 now ARE those of rank 0
    assert numpy.array_equal(ranks[1].get(L.F_PHI), ranks[0].get(L.F_PHI))
    close_all([one] + ranks)


def test_exchange_overflow_is_an_error(golden):
    nranks, nw = 2, 8
    model, one, ranks, rng = start(golden, nranks, nw)
    w = numpy.concatenate([numpy.full(nw, 3.0), numpy.full(nw, 1e-3)])
    ranks[0].set(L.F_WEIGHT, w[:nw]); ranks[1].set(L.F_WEIGHT, w[nw:])
    for rk in ranks:
        rk.comm_set_capacity(3)
    with pytest.raises(L.AfqError) as e:
        devmod.popcontrol_comb_local(ranks, 0.37, nranks * nw)
    assert e.value.code == L.AFQ_EOVERFLOW
    # asynchronous comb: the flag is sticky and surfaces at the next estimator fetch of every rank
    for rk in ranks:
        with pytest.raises(L.AfqError) as e:
            rk.estimates_get()
        assert e.value.code == L.AFQ_EOVERFLOW
    close_all([one] + ranks)


def test_multi_rank_loop_equals_one_rank(golden):
    """30 steps of the batched driver sequence (qmc/afqmc.py:223-255: reortho / 5, propagate with the in-kernel
    weight cap on the GLOBAL total weight, comb / 2 without read-back, estimators with energy / 5, all-reduce per
    block) on 3 ranks x 5 walkers against one rank x 15 walkers, same fields and comb uniforms."""
    nranks, nw = 3, 5
    model, one, ranks, rng = start(golden, nranks, nw, spread=0.3)
    ntot = nranks * nw
    eshift = 0.0
    for step in range(1, 31):
        xi = rng.normal(size=(ntot, one.K))
        r = rng.rand()
        groups = [(one, xi)] + [(ranks[i], xi[i * nw:(i + 1) * nw]) for i in range(nranks)]
        for dev, x in groups:
            if step % 5 == 0:
                dev.reortho(fetch=False)
            dev.set_weight_cap(0.10 if step > 1 else 0.0, -1.0)
            dev.propagate(x, eshift)
        if step % 2 == 0:
            one.popcontrol_comb(r, ntot, fetch=False)
            devmod.popcontrol_comb_local(ranks, r, ntot, fetch=False)
        for dev, _ in groups:
            dev.estimates_update(step % 5 == 0)
        if step % 10 == 0:
            devmod.estimates_allreduce_local(ranks)
            e_one = one.estimates_get(zero=True)
            for rk in ranks:
                e = rk.estimates_get(zero=True)
                assert numpy.max(numpy.abs(e - e_one)) <= 1e-10 * numpy.max(numpy.abs(e_one))
            eshift = (e_one[7] / e_one[1]).real                     # ehyb / weight, estimators/mixed.py:268-271
        same_population(one, ranks, exact=False)
    assert ranks[0].comm_stats()['events'] == 15
    close_all([one] + ranks)


def test_mixed_one_rdm_travels_and_reduces(golden):
    """estimators/mixed.py:226-229 across ranks: walker.G is walker state (it is accumulated stale between energy
    evaluations), so it travels with a cloned walker, and the RDM sums are part of the block reduction (:261).
    3 ranks x 5 walkers against one rank x 15 walkers."""
    nranks, nw = 3, 5
    model, one, ranks, rng = start(golden, nranks, nw, spread=0.8)
    ntot = nranks * nw
    for dev in [one] + ranks:
        dev.estimates_rdm(True)
    crossed = False
    for step in range(1, 21):
        xi = rng.normal(size=(ntot, one.K))
        r = rng.rand()
        one.propagate(xi, 0.0)
        for i, rk in enumerate(ranks):
            rk.propagate(xi[i * nw:(i + 1) * nw], 0.0)
        if step % 2 == 0:
            pix_one, _ = one.popcontrol_comb(r, ntot)
            pix, _ = devmod.popcontrol_comb_local(ranks, r, ntot)
            assert numpy.array_equal(pix, pix_one)
            kill, clone = numpy.where(pix == 0)[0], numpy.where(pix > 1)[0]
            crossed = crossed or any(c // nw != k // nw for c, k in zip(clone, kill))
        for dev in [one] + ranks:
            dev.estimates_update(step % 5 == 0)         # energy (and a fresh walker.G) every 5th step only
        a, b = one.get(L.F_G), gather(ranks, L.F_G)
        assert numpy.max(numpy.abs(a - b)) <= 1e-12 * numpy.max(numpy.abs(a))
        if step % 10 == 0:
            devmod.estimates_allreduce_local(ranks)
            want = one.estimates_rdm_get(zero=True)
            for rk in ranks:
                got = rk.estimates_rdm_get(zero=True)
                assert numpy.max(numpy.abs(got - want)) <= 1e-10 * numpy.max(numpy.abs(want))
            for dev in [one] + ranks:
                dev.estimates_get(zero=True)
    assert crossed
    close_all([one] + ranks)


def test_back_propagation_state_travels(golden):
    nranks, nw, nbp = 2, 6, 4
    model, one, ranks, rng = start(golden, nranks, nw, spread=0.8)
    ntot = nranks * nw
    for dev in [one] + ranks:
        dev.bp_configure(nbp)
    for step in range(nbp):
        xi = rng.normal(size=(ntot, one.K))
        one.propagate(xi, 0.0)
        for i, rk in enumerate(ranks):
            rk.propagate(xi[i * nw:(i + 1) * nw], 0.0)
        r = rng.rand()
        pix_one, _ = one.popcontrol_comb(r, ntot)
        pix, _ = devmod.popcontrol_comb_local(ranks, r, ntot)
        assert numpy.array_equal(pix, pix_one)
    assert numpy.array_equal(one.bp_steps(), numpy.concatenate([rk.bp_steps() for rk in ranks]))
    _, den_one, G_one = one.bp_update(model.psi, 10, 'full')
    den, G = 0.0, 0.0
    for rk in ranks:
        _, dn, g = rk.bp_update(model.psi, 10, 'full')
        den, G = den + dn, G + g
    assert abs(den - den_one) <= 1e-11 * abs(den_one)
    assert numpy.max(numpy.abs(G - G_one)) <= 1e-10 * numpy.max(numpy.abs(G_one))
    close_all([one] + ranks)


def test_rccl_communicator_of_one_rank(golden):
    """afq_comm_init / the RCCL calls (ncclCommInitRank, ncclAllGather of the weights, ncclAllReduce of the
    estimators) on the one GPU of the box: must reproduce the plain single-rank path."""
    d = golden('generic_ops.npz')
    model = generic_model(d, 'A_')
    nw = 8
    rng = numpy.random.RandomState(11)
    M, nt = model.M, model.na + model.nb
    phis = numpy.array([model.psi + 0.05 * (rng.rand(M, nt) + 1j * rng.rand(M, nt)) for _ in range(nw)])
    w = numpy.exp(rng.normal(size=nw))
    plain, withcomm = make_device(model, nw), make_device(model, nw)
    withcomm.comm_init(withcomm.comm_unique_id(), 0, 1)
    xi = rng.normal(size=(nw, plain.K))
    for dev in (plain, withcomm):
        dev.set(L.F_PHI, phis); dev.set(L.F_WEIGHT, w)
        dev.set(L.F_OT, numpy.array([ref.calc_overlap(p, model.psi, model.na, model.nb) for p in phis]))
        dev.propagate(xi, 0.0)
    pa, ta = plain.popcontrol_comb(0.61, nw)
    pb, tb = withcomm.popcontrol_comb(0.61, nw)
    assert numpy.array_equal(pa, pb) and ta == tb
    for f in FIELDS:
        assert numpy.array_equal(plain.get(f), withcomm.get(f)), f
    for dev in (plain, withcomm):
        dev.estimates_update(True)
    withcomm.estimates_allreduce()
    assert numpy.array_equal(plain.estimates_get(), withcomm.estimates_get())
    buf = rng.rand(7) + 1j * rng.rand(7)
    assert numpy.array_equal(withcomm.estimates_allreduce(buf.copy()), buf)
    st = withcomm.comm_stats()
    assert (st['rank'], st['size'], st['events']) == (0, 1, 1)
    withcomm.comm_destroy()
    close_all([plain, withcomm])


def test_ipc_communicator_of_one_rank(golden):
    """afq_comm_init_ipc with one rank: the window all-gather, the window all-reduce and afq_comm_probe run (a rank also
    writes its own row through the same kernels), the bootstrap callback is called; equal to the plain single-rank path."""
    d = golden('generic_ops.npz')
    model = generic_model(d, 'A_')
    nw = 8
    rng = numpy.random.RandomState(11)
    M, nt = model.M, model.na + model.nb
    phis = numpy.array([model.psi + 0.05 * (rng.rand(M, nt) + 1j * rng.rand(M, nt)) for _ in range(nw)])
    w = numpy.exp(rng.normal(size=nw))
    plain, withcomm = make_device(model, nw), make_device(model, nw)
    calls = []
    withcomm.comm_init_ipc(0, 1, lambda mine: (calls.append(len(mine)), mine)[1])
    assert list(withcomm.comm_probe()) == [0, 0, 0]
    xi = rng.normal(size=(nw, plain.K))
    for dev in (plain, withcomm):
        dev.set(L.F_PHI, phis); dev.set(L.F_WEIGHT, w)
        dev.set(L.F_OT, numpy.array([ref.calc_overlap(p, model.psi, model.na, model.nb) for p in phis]))
        dev.propagate(xi, 0.0)
    pa, ta = plain.popcontrol_comb(0.61, nw)
    pb, tb = withcomm.popcontrol_comb(0.61, nw)
    assert numpy.array_equal(pa, pb) and ta == tb
    for f in FIELDS:
        assert numpy.array_equal(plain.get(f), withcomm.get(f)), f
    for dev in (plain, withcomm):
        dev.estimates_update(True)
    withcomm.estimates_allreduce()
    assert numpy.array_equal(plain.estimates_get(), withcomm.estimates_get())
    buf = rng.rand(7) + 1j * rng.rand(7)
    assert numpy.array_equal(withcomm.estimates_allreduce(buf.copy()), buf)
    big = rng.rand(3 * M * M) + 1j * rng.rand(3 * M * M)         # longer than one window row: reduced in chunks
    assert numpy.array_equal(withcomm.estimates_allreduce(big.copy()), big)
    st = withcomm.comm_stats()
    assert (st['rank'], st['size'], st['events'], st['kind'], st['error']) == (0, 1, 1, 'ipc', 0)
    withcomm.comm_destroy()
    close_all([plain, withcomm])


def test_spin_summed_force_bias_after_an_exchange():
    """More than 32 walkers per rank and an RHF-type trial: the force bias contracts the spin-summed Ghalf, which the
    Green's-function kernel writes -- walkers received from another rank bring their Ghalf, so the sum has to be
    redone.  2 ranks x 36 walkers against one rank x 72 (whose path tests/test_gpu_sizes.py checks against the oracle)."""
    from tests.test_gpu_sizes import build
    nranks, nw = 2, 36
    model, rng = build(20, 30, 4, 4, False, seed=23)
    ntot = nranks * nw
    phis = numpy.array([model.psi + 0.1 * (rng.rand(20, 8) + 1j * rng.rand(20, 8)) for _ in range(ntot)])
    ots = numpy.array([ref.calc_overlap(p, model.psi, 4, 4) for p in phis])
    weights = numpy.exp(0.8 * rng.normal(size=ntot))
    one = make_device(model, ntot)
    ranks = [make_device(model, nw) for _ in range(nranks)]
    for dev, sl in [(one, slice(0, ntot))] + [(ranks[i], slice(i * nw, (i + 1) * nw)) for i in range(nranks)]:
        dev.set(L.F_PHI, phis[sl]); dev.set(L.F_OT, ots[sl]); dev.set(L.F_WEIGHT, weights[sl])
    devmod.comm_init_local(ranks)
    crossed = False
    for step in range(1, 5):
        xi = rng.normal(size=(ntot, one.K))
        one.propagate(xi, 0.1)
        for i, rk in enumerate(ranks):
            rk.propagate(xi[i * nw:(i + 1) * nw], 0.1)
        if step < 4:
            r = rng.rand()
            pix_one, _ = one.popcontrol_comb(r, ntot)
            pix, _ = devmod.popcontrol_comb_local(ranks, r, ntot)
            assert numpy.array_equal(pix, pix_one)
            kill, clone = numpy.where(pix == 0)[0], numpy.where(pix > 1)[0]
            crossed = crossed or any(c // nw != k // nw for c, k in zip(clone, kill))
        same_population(one, ranks, exact=False)
    assert crossed
    close_all([one] + ranks)


def test_windows_above_512_walkers_per_rank_start_smaller_than_the_population_and_grow(golden):
    """More than 512 walkers per rank: the peer windows start at max(512, nw / 4) slots per peer (k_comm.hip: default_cap),
    with the overflow flag as the guard; the driver grows them at a block boundary as soon as the largest transfer seen
    reaches half the capacity (Walkers.tune_exchange_capacity).  600 walkers on each of two ranks: a skewed population
    that moves 300 walkers stays within the initial 512 slots and equals the one-rank comb; the statistics make the driver
    raise the capacity to the population size; then the extreme case -- every walker of rank 1 replaced by a clone from
    rank 0, 600 transfers, which 512 slots could not hold -- goes through."""
    from pauxy_amd.walkers.handler import Walkers
    nranks, nw = 2, 600
    model, one, ranks, rng = start(golden, nranks, nw)
    assert ranks[0].comm_stats()['capacity'] == 512

    class Host(object):                                    # what Walkers.tune_exchange_capacity touches
        def __init__(self, dev):
            self.dev, self.nw, self.device_comm = dev, nw, True

    def comb_with(w, r):
        one.set(L.F_WEIGHT, w)
        for i, rk in enumerate(ranks):
            rk.set(L.F_WEIGHT, w[i * nw:(i + 1) * nw])
        pix_one, _ = one.popcontrol_comb(r, nranks * nw)
        pix, _ = devmod.popcontrol_comb_local(ranks, r, nranks * nw)
        assert numpy.array_equal(pix, pix_one)
        same_population(one, ranks)
        return pix

    # half of rank 1's walkers die, rank 0's are heavy enough to fill the gaps: 300 walkers cross
    w = numpy.concatenate([numpy.full(nw, 1.5), numpy.full(nw // 2, 1.0), numpy.full(nw // 2, 1e-6)])
    comb_with(w, 0.41)
    st = ranks[0].comm_stats()
    assert st['overflow'] == 0 and 256 <= st['max_transfer'] <= 512, st
    for rk in ranks:
        Walkers.tune_exchange_capacity(Host(rk))
    assert [rk.comm_stats()['capacity'] for rk in ranks] == [nw, nw]
    # every walker of rank 1 dies, every walker of rank 0 is cloned once: nw transfers in one event
    w = numpy.concatenate([numpy.full(nw, 3.0), numpy.full(nw, 1e-6)])
    pix = comb_with(w, 0.37)
    assert numpy.all(pix[nw:] == 0) and numpy.all(pix[:nw] == 2)
    st = ranks[0].comm_stats()
    assert st['overflow'] == 0 and st['max_transfer'] == nw and st['error'] == 0
    close_all([one] + ranks)


@pytest.mark.parametrize("open_walker", [False, True])
def test_closed_shell_verdict_covers_walkers_received_from_another_rank(open_walker):
    """The one-spin exchange energy (k_energy.hip: launch_exx_quadratic) rests on a device flag that says every walker's
    cached Ghalf has equal spin blocks.  Walkers that arrive from another rank bring their Ghalf along, so the unpack kernel
    checks them: rank 0 holds closed walkers only, rank 1 a heavy walker -- closed, or open (beta block different) -- whose
    clones land on rank 0.  Energies after the exchange against the oracle on every walker of both ranks
    (estimators/generic.py:49-83 through the walkers' own Green's functions)."""
    from pauxy_amd import systems, trial as trial_mod
    from pauxy_amd.propagation import setup
    M, N, K, nw, nranks = 40, 13, 30, 36, 2
    s = systems.synthetic_generic(M, K, (N, N), seed=7)
    t = trial_mod.rhf_trial_generic(s)
    dt = 0.005
    BH1, mf = setup.generic_propagator_arrays(s, t, dt)
    model = ref.RefModel('generic', M, N, N, t.psi, BH1, mf, dt, hs_pot=s.hs_pot, rchol=t._rchol,
                         H1=s.H1.astype(complex), ecore=s.ecore)
    rng = numpy.random.RandomState(91)
    ntot = nranks * nw
    half = t.psi[None, :, :N] + 0.05 * (rng.rand(ntot, M, N) + 1j * rng.rand(ntot, M, N))
    phis = numpy.concatenate([half, half], axis=2)
    heavy = nw + 5
    if open_walker:
        phis[heavy, :, N:] += 0.03 * (rng.rand(M, N) + 1j * rng.rand(M, N))
    weights = numpy.ones(ntot)
    weights[heavy] = 9.0
    weights[:6] = 0.05                                                  # rank 0 loses walkers, rank 1's heavy one fills the slots
    ots = numpy.array([ref.calc_overlap(p, model.psi, N, N) for p in phis])
    ranks = [make_device(model, nw) for _ in range(nranks)]
    for i, rk in enumerate(ranks):
        sl = slice(i * nw, (i + 1) * nw)
        rk.set(L.F_PHI, phis[sl]); rk.set(L.F_OT, ots[sl]); rk.set(L.F_WEIGHT, weights[sl])
    devmod.comm_init_local(ranks)
    xi = rng.normal(size=(ntot, K))
    for i, rk in enumerate(ranks):
        rk.propagate(xi[i * nw:(i + 1) * nw], -0.1)                     # leaves the end-of-step Green's functions behind
        rk.set(L.F_WEIGHT, weights[i * nw:(i + 1) * nw])
    pix, _ = devmod.popcontrol_comb_local(ranks, 0.37, ntot)
    assert pix[heavy] > 1 and any(pix[w] == 0 for w in range(nw)), pix
    for i, rk in enumerate(ranks):
        E = rk.local_energy()
        out = rk.get(L.F_PHI)
        n_open = 0
        for w in range(nw):
            is_open = not numpy.array_equal(out[w, :, :N], out[w, :, N:])
            n_open += is_open
            _, gh_ref, G_ref = ref.greens_function(out[w], model.psi, N, N)
            e_ref = numpy.array(model.local_energy(G_ref, gh_ref))
            assert numpy.max(numpy.abs(E[w] - e_ref)) <= 1e-10 * max(1.0, numpy.max(numpy.abs(e_ref))), (i, w, is_open)
        if i == 0:
            assert (n_open > 0) == open_walker, n_open                  # the heavy walker's clones did land on rank 0
    close_all(ranks)
