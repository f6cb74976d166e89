"""Shared builders: golden fixture -> oracle RefModel / plain arrays."""
import numpy
import scipy.sparse

from oracle import afqmc_ref as ref


def ragged(d, name, tag=''):
    flat = d[tag + name + '_flat']
    off = d[tag + name + '_off']
    return [flat[off[i]:off[i + 1]].astype(numpy.int64) for i in range(len(off) - 1)]


def ueg_sparse(d, M, tag=''):
    nq = len(d[tag + 'vqvec'])
    mats = []
    for name in ('iA', 'iB'):
        mats.append(scipy.sparse.csc_matrix(
            (d[tag + name + '_val'], (d[tag + name + '_row'], d[tag + name + '_col'])),
            shape=(M * M, nq), dtype=numpy.complex128))
    return mats


def generic_model(d, tag, rchol=None):
    na, nb = [int(x) for x in d[tag + 'nelec']]
    h1e = d[tag + 'h1e']
    M = h1e.shape[0]
    return ref.RefModel('generic', M, na, nb, d[tag + 'psi'], d[tag + 'BH1'], d[tag + 'mf_shift'],
                        float(d[tag + 'dt']), hs_pot=d[tag + 'chol'],
                        rchol=d[tag + 'rchol'] if rchol is None else rchol,
                        H1=numpy.array([h1e, h1e]), ecore=float(d[tag + 'ecore']))


def hubbard_model(d, tag, kind):
    na, nb = [int(x) for x in d['nelec']]
    M = d['T'].shape[-1]
    return ref.RefModel(kind, M, na, nb, d[tag + 'psi'], d[tag + 'BH1'], d[tag + 'mf_shift'],
                        float(d[tag + 'dt']), U=float(d['U']), H1=d['T'])


def ueg_model(d, tag, systag=''):
    na, nb = [int(x) for x in d[systag + 'nelec']]
    M = len(d[systag + 'sp_eigv'])
    iA, iB = ueg_sparse(d, M, systag)
    H1diag = numpy.array([d[systag + 'sp_eigv'], d[systag + 'sp_eigv']])
    return ref.RefModel('ueg', M, na, nb, d[tag + 'psi'], d[tag + 'BH1'], d[tag + 'mf_shift'],
                        float(d[tag + 'dt']), iA=iA, iB=iB, H1diag=H1diag,
                        vqvec=d[systag + 'vqvec'], vol=float(d[systag + 'vol']),
                        ikpq_i=ragged(d, 'ikpq_i', systag), ikpq_kpq=ragged(d, 'ikpq_kpq', systag),
                        ipmq_i=ragged(d, 'ipmq_i', systag), ipmq_pmq=ragged(d, 'ipmq_pmq', systag),
                        ecore=float(d[systag + 'ecore']))
