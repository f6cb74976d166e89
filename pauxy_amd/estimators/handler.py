"""Container the driver calls (pauxy/estimators/handler.py:56-162): owns the
``mixed`` estimator and forwards ``update`` / ``print_step``.  Back-propagation
and ITCF estimators are not on the device path yet."""
from pauxy_amd.estimators.mixed import Mixed


class Estimators(object):
    def __init__(self, estimates, root, qmc, system, trial, BT2, verbose=False):
        self.filename = None
        self.basename = estimates.get('basename', 'estimates')
        self.index = estimates.get('index', 0)
        mixed = estimates.get('mixed', {})
        self.estimators = {}
        self.estimators['mixed'] = Mixed(mixed, system, root, self.filename, qmc, trial, complex)
        for key in ('back_propagation', 'back_propagated', 'itcf'):
            if estimates.get(key) is not None:
                raise NotImplementedError("%s estimator is not on the device path yet" % key)
        self.back_propagation = False
        self.calc_itcf = False
        self.nprop_tot = None
        self.nbp = None
        self.json_string = ''

    def dump_metadata(self):
        pass

    def print_step(self, comm, nprocs, step, nsteps=None, free_projection=False):
        for k, e in self.estimators.items():
            e.print_step(comm, nprocs, step, nsteps=nsteps, free_projection=free_projection)

    def update(self, system, qmc, trial, psi, step, free_projection=False):
        for k, e in self.estimators.items():
            e.update(system, qmc, trial, psi, step, free_projection)
