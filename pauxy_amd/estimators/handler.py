"""Container the driver calls (pauxy/estimators/handler.py:56-162): owns the
``mixed`` estimator (and, when requested, ``back_prop``) and forwards ``update`` /
``print_step``.  The ITCF estimator is not on the device path.

Output file (handler.py:60-71,117-124): ``<basename>.<index>.h5`` (or ``filename``),
created empty by the root rank, ``metadata`` = JSON description of the run; as in
the reference the default is ``estimates.<index>.h5`` in the working directory.
Opting out: ``estimators: {write_file: False}``, or ``AFQ_ESTIMATES_FILE=0`` in the
environment for harnesses that only want the in-memory rows ``Mixed.blocks`` (the
variable only changes the default: a run that names its file still gets it)."""
import os

from pauxy_amd.estimators.back_propagation import BackPropagation
from pauxy_amd.estimators.mixed import Mixed
from pauxy_amd.utils import io as _io


def _series_file(basename, index):
    """Name of file number ``index`` of an output series: <basename>.<index>.h5 (handler.py:64,122-124)."""
    return "%s.%s.h5" % (basename, index)


def _touch(filename):
    with _io.h5.File(filename, 'w'):
        pass


class Estimators(object):
    def __init__(self, estimates, root, qmc, system, trial, BT2, verbose=False):
        opts = estimates
        self.index = opts.get('index', 0)
        self.basename = opts.get('basename', 'estimates')                    # handler.py:62
        self.flush_every = opts.get('flush_every')
        named = opts.get('filename') is not None or 'basename' in opts
        write = opts.get('write_file', named or os.environ.get('AFQ_ESTIMATES_FILE', '1') != '0')
        name = opts.get('filename') if (root and write) else None
        if root and write and name is None:
            # first free slot of the series unless overwriting is allowed (handler.py:63-69)
            keep_existing = not opts.get('overwrite', True)
            name = _series_file(self.basename, self.index)
            while keep_existing and os.path.isfile(name):
                self.index += 1
                name = _series_file(self.basename, self.index)
        self.filename = name
        if name is not None:
            _touch(name)
        self.estimators = {}
        mixed_opts = dict(opts.get('mixed', {}), flush_every=opts.get('mixed', {}).get('flush_every', self.flush_every))
        self.estimators['mixed'] = Mixed(mixed_opts, system, root, name, qmc, trial, complex)
        if opts.get('itcf') is not None:
            raise NotImplementedError("itcf estimator is not on the device path yet")
        bp_opts = opts.get('back_propagation', opts.get('back_propagated'))              # handler.py:83-85
        self.back_propagation = bp_opts is not None
        self.nprop_tot = self.nbp = None
        if self.back_propagation:
            bp_opts = dict(bp_opts, flush_every=bp_opts.get('flush_every', self.flush_every))
            est = BackPropagation(bp_opts, root, name, qmc, system, trial, complex, BT2)
            self.estimators['back_prop'] = est
            self.nprop_tot = self.nbp = est.nmax                                         # handler.py:91-92
        self.calc_itcf = False
        self.json_string = ''

    def dump_metadata(self):
        """handler.py:117-120: the run description as the ``metadata`` dataset (replaced if present)."""
        if self.filename is None:
            return
        with _io.h5.File(self.filename, 'a') as fh5:
            if 'metadata' in fh5:
                del fh5['metadata']
            fh5['metadata'] = self.json_string

    def increment_file_number(self):
        self.index += 1
        self.filename = _series_file(self.basename, self.index)

    def reset(self, root):
        """handler.py:110-115: start the next file of the series."""
        if not root or self.filename is None:
            return
        self.flush()
        self.increment_file_number()
        _touch(self.filename)
        self.dump_metadata()
        for est in self.estimators.values():
            est.setup_output(self.filename)

    def flush(self):
        """Write the estimator blocks still queued (pauxy_amd/estimators/utils.py)."""
        for est in self.estimators.values():
            out = getattr(est, 'output', None)
            if out is not None:
                out.flush()

    def print_step(self, comm, nprocs, step, nsteps=None, free_projection=False):
        for est in self.estimators.values():
            est.print_step(comm, nprocs, step, nsteps=nsteps, free_projection=free_projection)

    def update(self, system, qmc, trial, psi, step, free_projection=False):
        for est in self.estimators.values():
            est.update(system, qmc, trial, psi, step, free_projection)
