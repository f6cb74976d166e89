"""SURVEY 8a row 3 in performance mode: the device stream of auxiliary fields that replaces
numpy.random.normal(0, 1, nfields) (propagation/continuous.py:133; seeding rule qmc/utils.py:3-16).
Known-answer test of the Philox4x32-10 core against the Random123 vectors, bit-for-bit reproduction of the
stream by the Python restatement (tests/philox_ref.py), the stream afq_propagate actually consumes, and the
statistics of 256 walkers x 500 fields x 100 steps (the bench.py workload): moments, serial and cross-walker
correlation, independence of the per-rank streams."""
import numpy
import pytest

from pauxy_amd import _lib as L
from pauxy_amd.device import AfqDevice
from tests.philox_ref import KAT, device_normals, philox4x32_10

pytestmark = pytest.mark.gpu
NW, K, STEPS = 256, 500, 100


def test_philox_known_answers_on_device():
    dev = AfqDevice(0)
    ck = numpy.array([list(c) + list(k) for c, k, _ in KAT], dtype=numpy.uint32)
    out = dev.philox4x32(ck)
    for row, (_, _, want) in zip(out, KAT):
        assert tuple(int(x) for x in row) == want
    # and a few hundred random counters / keys against the restatement
    rng = numpy.random.RandomState(3)
    ck = rng.randint(0, 2 ** 32, size=(300, 6), dtype=numpy.uint64).astype(numpy.uint32)
    out = dev.philox4x32(ck)
    for row, q in zip(out, ck):
        assert tuple(int(x) for x in row) == philox4x32_10(tuple(int(x) for x in q[:4]), (int(q[4]), int(q[5])))
    dev.close()


def test_device_normals_match_restatement():
    dev = AfqDevice(0)
    seed, stream = 0x1234567890ABCDEF, 5
    dev.rng_seed(seed, stream)
    n = 4001                                             # odd: the last pair is half used
    for counter in range(3):                             # every launch advances the counter word
        x = dev.rng_normal(n)
        want = device_normals(n, seed, stream, counter)
        assert numpy.max(numpy.abs(x - want)) < 1e-13
    dev.close()


def test_propagate_consumes_the_stream(golden):
    """afq_propagate(xi=NULL): with the force bias off the shifted fields ARE the drawn fields."""
    from tests.helpers import hubbard_model, make_device
    d = golden('hubbard_ops.npz')
    model = hubbard_model(d, 'C_', 'hubbard')
    nw = 6
    dev = make_device(model, nw, force_bias=False)
    rng = numpy.random.RandomState(2)
    M, nt = model.M, model.na + model.nb
    dev.set(L.F_PHI, numpy.array([model.psi + 0.05 * rng.rand(M, nt) for _ in range(nw)], dtype=complex))
    dev.rng_seed(99, 3)
    for counter in range(2):
        dev.propagate(None, 0.0)
        xs = dev.get(L.F_XSHIFTED)
        want = device_normals(nw * dev.K, 99, 3, counter).reshape(nw, dev.K)
        assert numpy.max(numpy.abs(xs.imag)) == 0.0
        assert numpy.max(numpy.abs(xs.real - want)) < 1e-13
    dev.close()


def draws(dev, seed, stream):
    dev.rng_seed(seed, stream)
    return numpy.array([dev.rng_normal(NW * K).reshape(NW, K) for _ in range(STEPS)])   # [step, walker, field]


def test_stream_statistics():
    dev = AfqDevice(0)
    x = draws(dev, 7, 0)
    n = x.size
    sig = 1.0 / numpy.sqrt(n)
    assert abs(x.mean()) < 4 * sig
    assert abs(x.var() - 1.0) < 0.01                                      # 1 %; 4 sigma is 4 sqrt(2 / n) = 1.6e-3
    assert abs(x.var() - 1.0) < 4 * numpy.sqrt(2.0 / n)
    assert abs((x ** 3).mean()) < 4 * numpy.sqrt(15.0 / n)                # skewness
    assert abs((x ** 4).mean() - 3.0) < 4 * numpy.sqrt(96.0 / n)          # kurtosis
    # lag-1 along the fields of one walker (adjacent elements share a Philox block: cos / sin of one pair)
    assert abs((x[..., 1:] * x[..., :-1]).mean()) < 4 / numpy.sqrt(x[..., 1:].size)
    # the two members of a Box-Muller pair must not be correlated in their squares either
    assert abs(((x[..., 0::2] ** 2 - 1) * (x[..., 1::2] ** 2 - 1)).mean()) < 4 * 2.0 / numpy.sqrt(x[..., 0::2].size)
    # neighbouring walkers, the same field; consecutive steps, the same (walker, field)
    assert abs((x[:, 1:, :] * x[:, :-1, :]).mean()) < 4 / numpy.sqrt(x[:, 1:, :].size)
    assert abs((x[1:] * x[:-1]).mean()) < 4 / numpy.sqrt(x[1:].size)
    # full cross-walker correlation matrix over (step, field) samples: off-diagonal entries ~ N(0, 1/50000)
    flat = x.transpose(1, 0, 2).reshape(NW, -1)
    corr = flat @ flat.T / flat.shape[1]
    off = corr[~numpy.eye(NW, dtype=bool)]
    assert numpy.max(numpy.abs(off)) < 5.5 / numpy.sqrt(flat.shape[1])    # max of 65 280 normals: 5.5 sigma
    # tails: P(|x| > 4) = 6.33e-5
    tail = numpy.count_nonzero(numpy.abs(x) > 4.0)
    assert abs(tail - 6.334e-5 * n) < 5 * numpy.sqrt(6.334e-5 * n)
    # different ranks (stream = global rank) and different seeds: element-wise uncorrelated, never equal
    for seed, stream in ((7, 1), (7, 7), (8, 0)):
        y = draws(dev, seed, stream)
        assert abs((x * y).mean()) < 4 * sig
        assert numpy.count_nonzero(x == y) == 0
    # and the same (seed, stream) is reproducible
    assert numpy.array_equal(draws(dev, 7, 0), x)
    dev.close()


def test_afqmc_uses_the_global_rank_as_stream():
    """Two ranks with the same LOCAL rank must not draw the same fields (the host path seeds seed + rank)."""
    from pauxy_amd.comm import FakeComm
    from pauxy_amd.qmc.afqmc import AFQMC
    from pauxy_amd import systems, trial as trial_mod
    from pauxy_amd.context import release_context

    class Rank(FakeComm):
        def __init__(self, rank):
            self.rank = rank

    xs = []
    for rank in (0, 3):
        s = systems.synthetic_generic(12, 20, (3, 3), seed=7)
        t = trial_mod.rhf_trial_generic(s)
        opts = {'qmc': {'num_walkers': 4, 'rng_seed': 5}, 'propagator': {'device_rng': True, 'rng_seed': 5},
                'estimators': {'mixed': {'verbose': False}}}
        afqmc = AFQMC(comm=Rank(rank), options=opts, system=s, trial=t)
        xs.append(afqmc.psi.dev.rng_normal(64))
        release_context(s, t)
    assert numpy.count_nonzero(xs[0] == xs[1]) == 0
    assert numpy.max(numpy.abs(xs[0] - device_normals(64, 5, 0, 0))) < 1e-13
    assert numpy.max(numpy.abs(xs[1] - device_normals(64, 5, 3, 0))) < 1e-13
