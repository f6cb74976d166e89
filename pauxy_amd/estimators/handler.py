"""Container the driver calls (pauxy/estimators/handler.py:56-162): owns the
``mixed`` estimator (and, when requested, ``back_prop``) and forwards ``update`` /
``print_step``.  The ITCF estimator is not on the device path."""
from pauxy_amd.estimators.back_propagation import BackPropagation
from pauxy_amd.estimators.mixed import Mixed


class Estimators(object):
    def __init__(self, estimates, root, qmc, system, trial, BT2, verbose=False):
        self.filename = None
        self.basename = estimates.get('basename', 'estimates')
        self.index = estimates.get('index', 0)
        mixed = estimates.get('mixed', {})
        self.estimators = {}
        self.estimators['mixed'] = Mixed(mixed, system, root, self.filename, qmc, trial, complex)
        if estimates.get('itcf') is not None:
            raise NotImplementedError("itcf estimator is not on the device path yet")
        bp = estimates.get('back_propagation', estimates.get('back_propagated'))     # handler.py:83-85
        self.back_propagation = bp is not None
        if self.back_propagation:
            self.estimators['back_prop'] = BackPropagation(bp, root, self.filename, qmc, system, trial, complex, BT2)
            self.nprop_tot = self.estimators['back_prop'].nmax                           # handler.py:91-92
            self.nbp = self.estimators['back_prop'].nmax
        else:
            self.nprop_tot = None
            self.nbp = None
        self.calc_itcf = False
        self.json_string = ''

    def dump_metadata(self):
        pass

    def print_step(self, comm, nprocs, step, nsteps=None, free_projection=False):
        for k, e in self.estimators.items():
            e.print_step(comm, nprocs, step, nsteps=nsteps, free_projection=free_projection)

    def update(self, system, qmc, trial, psi, step, free_projection=False):
        for k, e in self.estimators.items():
            e.update(system, qmc, trial, psi, step, free_projection)
