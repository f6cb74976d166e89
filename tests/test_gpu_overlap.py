"""run_batched with the block boundary overlapped (the head of the first step of a block is enqueued before the host
waits for the sums of the block just ended: afq_propagate_begin / _finish, afq_estimates_get_begin / _end) against the
plain order: the energy shift enters a step only in its weight update, so every number must be IDENTICAL -- walkers,
weights, block estimates -- including across re-orthogonalisations and combs that fall on a block boundary."""
import numpy
import pytest

from pauxy_amd import systems, trial as trial_mod
from pauxy_amd.context import release_context
from pauxy_amd.qmc.afqmc import AFQMC

pytestmark = pytest.mark.gpu


def drive(overlap, nsteps_block, total):
    s = systems.synthetic_generic(20, 30, (5, 4), seed=9)
    t = trial_mod.rhf_trial_generic(s)
    opts = {'qmc': {'timestep': 0.01, 'num_steps': nsteps_block, 'blocks': 10 ** 6, 'stabilise_freq': 4,
                    'pop_control_freq': 3, 'num_walkers': 24, 'rng_seed': 5},
            'propagator': {'device_rng': True, 'rng_seed': 11},
            'estimators': {'mixed': {'verbose': False, 'energy_eval_freq': 2}}}
    afqmc = AFQMC(options=opts, system=s, trial=t)
    eshift = afqmc.run_batched(total // 2, first_step=1, eshift=0.0, overlap_blocks=overlap)
    # a second call continues the run: the boundary logic restarts cleanly with nothing half done
    eshift = afqmc.run_batched(total - total // 2, first_step=total // 2 + 1, eshift=eshift, overlap_blocks=overlap)
    mixed = afqmc.estimators.estimators['mixed']
    out = dict(blocks=numpy.array(mixed.blocks), eshift=eshift, weight=afqmc.psi._mirror('weight').copy(),
               phi=numpy.array([w.phi for w in afqmc.psi.walkers]), ot=afqmc.psi._mirror('ot').copy())
    release_context(s, t)
    return out


@pytest.mark.parametrize("nsteps_block,total", [(4, 24), (6, 36), (1, 8)])
def test_overlapped_block_boundary_is_bit_identical(nsteps_block, total):
    a = drive(False, nsteps_block, total)
    b = drive(True, nsteps_block, total)
    assert a['blocks'].shape == b['blocks'].shape and a['blocks'].shape[0] >= total // nsteps_block
    # column 0 = step, the last column = wall time of the block
    assert numpy.array_equal(a["blocks"][:, :-1], b["blocks"][:, :-1], equal_nan=True)   # blocks without an energy evaluation: 0 / 0
    assert a['eshift'] == b['eshift'] and a['eshift'] != 0.0
    for k in ('weight', 'phi', 'ot'):
        assert numpy.array_equal(a[k], b[k]), k
