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


def msd_model(d, tag, systag=''):
    """Multi-determinant generic model from msd_ops.npz / traj_msd.npz."""
    na, nb = [int(x) for x in d[systag + 'nelec']]
    h1e = d[systag + 'h1e']
    M = h1e.shape[0]
    return ref.RefModel('generic_msd', M, na, nb, d[tag + 'psi'], d[tag + 'BH1'], d[tag + 'mf_shift'], 0.005,
                        coeffs=d[tag + 'coeffs'], hs_pot=d[systag + 'chol'], H1=numpy.array([h1e, h1e]),
                        ecore=float(d[systag + 'ecore']))


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


def make_device(model, nw, device_id=0, **prop_kw):
    """RefModel (plain arrays) -> AfqDevice with nw walkers allocated."""
    from pauxy_amd.device import AfqDevice
    dev = AfqDevice(device_id)
    if model.kind == 'generic':
        dev.set_system_generic(model.hs_pot, model.rchol, model.H1, model.ecore, model.na, model.nb)
    elif model.kind in ('hubbard', 'hubbard_spin'):
        dev.set_system_hubbard(model.H1, model.U, model.na, model.nb)
        prop_kw.setdefault('hubbard_spin', model.kind == 'hubbard_spin')
    elif model.kind == 'ueg':
        dev.set_system_ueg(model.iA, model.iB, model.ikpq_i, model.ikpq_kpq, model.ipmq_i, model.ipmq_pmq,
                           model.vqvec, model.vol, model.H1diag, model.ecore, model.na, model.nb)
    if model.kind == 'generic_msd':
        from pauxy_amd import systems, trial as trial_mod
        s = systems.Generic((model.na, model.nb), model.H1.real, model.hs_pot, ecore=model.ecore)
        t = trial_mod.MultiDetTrial(s, (model.coeffs, model.psi))
        per = model.M * (model.na + model.nb)
        dev.set_system_generic(model.hs_pot, t._rchol[:per], model.H1, model.ecore, model.na, model.nb)
        dev.set_trial_multi(model.psi, model.coeffs, t._rchol)
    else:
        dev.set_trial(model.psi)
    dev.set_propagator(model.BH1, model.mf_shift, model.dt, exp_order=model.exp_order, **prop_kw)
    dev.walkers_alloc(nw)
    return dev
