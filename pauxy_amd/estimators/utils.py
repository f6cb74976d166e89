"""Estimator output file: one dataset per block under ``<base>/<name>/<index>``.

Layout of pauxy/estimators/utils.py:279-327 (``H5EstimatorHelper``): the block
counter is zero-padded to nine digits so that the dataset names sort, e.g.
``basic/energies/000000012``; headers live in ``<base>/headers`` as fixed-length
byte strings; ``metadata`` is a JSON string (estimators/handler.py:119-120).

The reference re-opens the file for every push.  Here pushes are queued and
written ``flush_every`` blocks at a time (1 = the reference's behaviour); the
queue is also written by ``flush()``, which the driver's ``finalise`` calls.
"""
import atexit
import weakref

import numpy

from pauxy_amd.utils import io as _io


def _flush_ref(ref):
    helper = ref()
    if helper is not None:
        helper.flush()


class H5EstimatorHelper(object):
    def __init__(self, filename, base, nav=1, flush_every=None):
        self.filename = filename
        self.base = base
        self.index = 0
        self.nzero = 9
        self.nav = nav
        if flush_every is None:
            # the built-in container rewrites the file on every append: batch by default
            flush_every = 1 if _io.HAVE_H5PY else 32
        self.flush_every = max(1, int(flush_every))
        self._pending = []
        atexit.register(_flush_ref, weakref.ref(self))

    def push(self, data, name):
        ix = str(self.index)
        dset = self.base + '/' + name + '/' + '0' * (self.nzero - len(ix)) + ix
        self._pending.append((dset, numpy.array(data)))
        if len(self._pending) >= self.flush_every:
            self.flush()

    def flush(self):
        if not self._pending:
            return
        with _io.h5.File(self.filename, 'a') as fh5:
            for dset, data in self._pending:
                fh5[dset] = data
        self._pending = []

    def increment(self):
        self.index = (self.index + 1) // self.nav

    def reset(self):
        self.index = 0
