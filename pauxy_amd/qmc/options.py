"""QMC options container (pauxy/qmc/options.py:84-121: names, aliases, defaults)."""


def _get(inputs, key, default, alias=()):
    if key in inputs:
        return inputs[key]
    for a in alias:
        if a in inputs:
            return inputs[a]
    return default


class QMCOpts(object):
    def __init__(self, inputs, system=None, verbose=False):
        self.nwalkers = _get(inputs, 'num_walkers', 10, ['nwalkers'])
        self.dt = _get(inputs, 'timestep', 0.005, ['dt'])
        self.nsteps = _get(inputs, 'num_steps', 10, ['nsteps', 'steps'])
        self.nblocks = _get(inputs, 'blocks', 1000, ['num_blocks', 'nblocks'])
        self.total_steps = self.nsteps * self.nblocks
        self.nstblz = _get(inputs, 'stabilise_freq', 10, ['nstabilise', 'reortho'])
        self.npop_control = _get(inputs, 'pop_control_freq', 1, ['npop_control', 'pop_control'])
        self.eqlb_time = _get(inputs, 'equilibration_time', 2.0, ['tau_eqlb'])
        self.neqlb = int(self.eqlb_time / self.dt)
        self.beta = None
        self.rng_seed = _get(inputs, 'rng_seed', None, ['random_seed', 'seed'])
        self.ntot_walkers = self.nwalkers
