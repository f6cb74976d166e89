"""The object walk the reference's driver applies to ITSELF, restated for the tests.

``pauxy.qmc.afqmc.AFQMC.__init__`` ends with ``to_json(self)`` (qmc/afqmc.py:193 -> utils/io.py:44-48), i.e.
``json.dumps(serialise(afqmc))`` with ``serialise`` of utils/misc.py:72-135: every attribute value that has a
``__dict__`` and is no function / HDF5 file is recursed into (:64-70, :82-84) -- no cycle guard, no depth limit --
dicts likewise (:85-86), functions and bound methods are skipped unless verbose (:87-94), the keys ``estimates`` /
``global_estimates`` are skipped (:95-96), ``walkers`` becomes the ``str`` of its first element (:97-98), 1-D arrays
become nested lists when their norm is non-zero (:99-119), scalars and ``None`` stay (:123-128), everything else is
dropped.  The plug-in objects must survive that walk and come out JSON-serialisable; this module applies the same
rules (``verbose=0``) so that the property can be checked wherever the suite runs, with the reference absent.
"""
import json
import types

import numpy
import scipy.sparse


def is_object(v):
    """utils/misc.py:64-70."""
    return (hasattr(v, '__class__') and '__dict__' in dir(v) and not isinstance(v, types.FunctionType)
            and 'h5py' not in str(type(v)))


def walk(obj, _depth=0):
    if _depth > 50:
        raise RecursionError("object graph does not terminate under the reference's serialise rules")
    out = {}
    items = obj.items() if isinstance(obj, dict) else obj.__dict__.items()
    for k, v in items:
        if isinstance(v, (scipy.sparse.csr_matrix, scipy.sparse.csc_matrix)):
            continue
        if is_object(v):
            out[k] = walk(v, _depth + 1)
        elif isinstance(v, dict):
            out[k] = walk(v, _depth + 1)
        elif isinstance(v, types.FunctionType) or hasattr(v, '__self__'):
            continue
        elif k in ('estimates', 'global_estimates'):
            continue
        elif k == 'walkers':
            out[k] = [str(x) for x in v][0]
        elif isinstance(v, numpy.ndarray):
            if v.ndim == 1 and v[0] is not None and numpy.linalg.norm(v) > 1e-8:
                out[k] = [[v.real.tolist(), v.imag.tolist()]] if v.dtype == complex else (v.tolist(),)
        elif k == 'store':
            continue
        elif isinstance(v, (int, float, bool, str)):
            out[k] = v
        elif isinstance(v, complex):
            out[k] = v.real
        elif v is None:
            out[k] = v
    return out


def kinds(tree):
    """The walk's result with every leaf replaced by the name of its JSON kind (what dropin_trace.json stores)."""
    if isinstance(tree, dict):
        return dict((k, kinds(v)) for k, v in tree.items())
    if isinstance(tree, (list, tuple)):
        return 'array'
    if isinstance(tree, bool):
        return 'bool'
    if isinstance(tree, (int, float)):
        return 'number'
    if tree is None:
        return 'null'
    return 'string'


def to_json(obj):
    """utils/io.py:44-48."""
    return json.dumps(walk(obj), sort_keys=False, indent=4)
