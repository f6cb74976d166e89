"""The C3-size golden trajectories (tests/golden/traj_generic_c3*.npz: BASELINE configs[2] -- M = 100, K = 500, 25 + 25
electrons, RHF trial -- through the GENUINE driver at the cadence bench.py times): what the CPU (oracle) and GPU tests share.
The fixtures hold seeds, checksums and the reference's outputs; the inputs regenerate here with the package's own set-up code
(held to the checksums), the fields from the recorded MT19937 state (held to per-row sums and first elements)."""
import numpy

from pauxy_amd import systems, trial as trial_mod
from pauxy_amd.propagation.setup import generic_propagator_arrays


def close(a, b, tol):
    a, b = numpy.asarray(a), numpy.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(numpy.max(numpy.abs(b))))
    err = float(numpy.max(numpy.abs(a - b))) / scale
    assert err <= tol, err


def inputs(d):
    """(system, trial, BH1, mf_shift) regenerated from the seed, checked against what the genuine reference built."""
    M, K = int(d['M']), int(d['K'])
    na, nb = [int(x) for x in d['nelec']]
    system = systems.synthetic_generic(M, K, (na, nb), seed=int(d['seed']))
    assert abs(system.H1[0].sum() - float(d['h1_sum'])) < 1e-9 * abs(float(d['h1_sum']))
    assert abs(numpy.abs(system.hs_pot).sum() - float(d['chol_abs_sum'])) < 1e-9 * float(d['chol_abs_sum'])
    trial = trial_mod.rhf_trial_generic(system)
    # (eigenvector signs are LAPACK's business: every observable of the run is invariant under psi, phi -> psi D, phi D)
    assert abs(numpy.abs(trial.psi).sum() - float(d['psi_abs_sum'])) < 1e-9 * float(d['psi_abs_sum'])
    rchol = numpy.asarray(trial._rchol)
    assert tuple(rchol.shape) == tuple(int(x) for x in d['rchol_shape'])
    assert abs(numpy.abs(rchol).sum() - float(d['rchol_abs_sum'])) < 1e-9 * float(d['rchol_abs_sum'])
    BH1, mf_shift = generic_propagator_arrays(system, trial, float(d['dt']))
    assert abs(BH1.sum() - complex(d['BH1_sum'])) < 1e-9 * abs(complex(d['BH1_sum']))
    assert abs(numpy.abs(BH1).sum() - float(d['BH1_abs_sum'])) < 1e-9 * float(d['BH1_abs_sum'])
    assert abs(mf_shift.sum() - complex(d['mf_shift_sum'])) < 1e-9 * max(1.0, abs(complex(d['mf_shift_sum'])))
    assert abs(numpy.abs(mf_shift).sum() - float(d['mf_shift_abs_sum'])) < 1e-9 * float(d['mf_shift_abs_sum'])
    return system, trial, BH1, mf_shift


def initial_walkers(d, trial):
    """phi0 [nw, M, na + nb]: the trial, with the beta blocks of d['open_ix'] perturbed exactly as make_golden.c3_open_walkers
    perturbs them."""
    nw = int(d['nwalkers'])
    na = int(d['nelec'][0])
    phi = numpy.array([numpy.asarray(trial.psi, dtype=numpy.complex128).copy() for _ in range(nw)])
    if 'open_ix' in d:
        rng = numpy.random.RandomState(int(d['open_seed']))
        for i in d['open_ix']:
            shape = phi[i][:, na:].shape
            phi[i][:, na:] += 1e-3 * (rng.rand(*shape) + 1j * rng.rand(*shape))
    return phi


class StateReplay(object):
    """Stands in for numpy.random.{normal, random}: the MT19937 stream the reference drew from, restarted from the recorded
    state and held to the recorded row sums; the comb is handed the recorded (where forced: the forced) uniform."""

    def __init__(self, d):
        self.d = d
        self.rs = numpy.random.RandomState()
        rest = d['rng_state_rest']
        self.rs.set_state(('MT19937', d['rng_state_keys'], int(rest[0]), int(rest[1]), float(d['rng_state_gauss'])))
        self.nw = int(d['nwalkers'])
        self.nrow = 0
        self.ncomb = 0
        self.forced = set(int(s) for s in d['r_override_steps'])

    def normal(self, loc, scale, size):
        x = self.rs.normal(loc, scale, size)
        s, w = divmod(self.nrow, self.nw)              # every walker is alive on every step of these runs
        self.nrow += 1
        assert x[0] == self.d['xi_first'][s, w], (s, w)
        assert abs(x.sum() - self.d['xi_sum'][s, w]) < 1e-10
        return x

    def random(self):
        x = self.rs.random_sample()
        self.ncomb += 1
        step = self.ncomb * int(self.d['npop_control'])
        r = float(self.d['r'][step - 1])
        assert step in self.forced or x == r
        return r


def open_history(d):
    """is_open[step] (bool [nw]) BEFORE the propagation of step 1 .. nsteps_total, following the recorded comb decisions
    through zip(clone, kill) (walkers/handler.py:295-301)."""
    nw = int(d['nwalkers'])
    is_open = numpy.zeros(nw, dtype=bool)
    if 'open_ix' in d:
        is_open[d['open_ix']] = True
    hist = []
    npop = int(d['npop_control'])
    total = int(d['nsteps']) * int(d['nblocks'])
    ev = 0
    for step in range(1, total + 1):
        hist.append(is_open.copy())
        if step % npop == 0:
            pix = d['parent_ix'][ev]
            ev += 1
            kill = numpy.where(pix == 0)[0]
            clone = numpy.where(pix > 1)[0]
            for c, k in zip(clone, kill):
                is_open[k] = is_open[c]
    return numpy.array(hist)
