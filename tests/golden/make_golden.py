#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the GENUINE reference.

Runs only in the build container (needs /root/reference).  Nothing of the
reference travels: this script copies /root/reference/pauxy to a scratch
directory under /tmp, imports it from there, and stores *data only* (inputs and
the reference's outputs) as small .npz files next to this script.

Import recipe: ``import h5py`` / ``from mpi4py import
MPI`` are unconditional in the reference's hot-path modules and neither package
is installed, so two no-arithmetic stub packages (an in-memory dict "file" and
a size-1 communicator) are put first on sys.path; three numpy-2 compatibility
edits (``ndarray.all() is None`` tests on object arrays) are applied to the
scratch copy; the one Cython module is built in the scratch copy.

Usage:  python tests/golden/make_golden.py [all | <group> | <fixture file> ...]   (writes tests/golden/*)
        python tests/golden/make_golden.py --check [...]    regenerate into a scratch directory and compare with the
                                                            committed fixtures (exit status 1 on any difference)
Groups: ops traj msd bp hirsch io dropin (see FIXTURES at the end of the file: one call per committed fixture).
"""
import os
import shutil
import subprocess
import sys

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
SCRATCH = os.environ.get("AFQ_ORACLE_SCRATCH", "/tmp/oracle")
REF = "/root/reference/pauxy"

H5PY_STUB = '''
import numpy
_STORE = {}
def _norm(p):
    return '/'.join(x for x in p.split('/') if x)
class Group(object):
    # path-prefix view of one in-memory "file" (a flat dict: dataset path -> array, group path -> None)
    def __init__(self, d, prefix):
        self.d = d
        self.prefix = prefix
    def _key(self, k):
        return _norm(self.prefix + '/' + k)
    def __setitem__(self, k, v):
        self.d[self._key(k)] = numpy.array(v)
    def __getitem__(self, k):
        key = self._key(k)
        if self.d.get(key) is not None:
            return self.d[key]
        if key in self.d or any(x.startswith(key + '/') for x in self.d):
            return Group(self.d, key)
        raise KeyError(key)
    def __contains__(self, k):
        try:
            self[k]
            return True
        except KeyError:
            return False
    def __delitem__(self, k):
        key = self._key(k)
        for x in [x for x in self.d if x == key or x.startswith(key + '/')]:
            del self.d[x]
    def create_group(self, k):
        key = self._key(k)
        if key in self.d or any(x.startswith(key + '/') for x in self.d):
            raise ValueError('name already exists')
        self.d[key] = None
        return Group(self.d, key)
    def create_dataset(self, name, shape=None, dtype=None, data=None):
        if data is None:
            data = numpy.zeros(shape, dtype=dtype)
        self[name] = data
        return self[name]
class File(Group):
    def __init__(self, name, mode='r', **kw):
        if mode == 'w':
            _STORE[name] = {}
        Group.__init__(self, _STORE.setdefault(name, {}), '')
        self.name = name
    def __enter__(self): return self
    def __exit__(self, *a): return False
    def close(self): pass
'''
MPI_STUB = '''
import numpy
SUM='sum'; COMM_TYPE_SHARED=0
class _Req:
    def wait(self): pass
class _Comm:
    rank=0; size=1
    def __init__(self): self._box={}; self.bcast_log=[]
    def Get_rank(self): return 0
    def Get_size(self): return 1
    def barrier(self): pass
    def Barrier(self): pass
    def bcast(self, x, root=0):
        self.bcast_log.append(x); return x
    def Bcast(self, x, root=0): return x
    def Reduce(self, s, r, op=None, root=0): r[...] = s
    def Allreduce(self, s, r, op=None): r[...] = s
    def Allgather(self, s, r): r[...] = numpy.asarray(s).reshape(r.shape)
    def gather(self, x, root=0): return [x]
    def scatter(self, x, root=0): return x[0]
    def Split_type(self, *a, **k): return self
    def Isend(self, buf, dest=0, tag=0):
        self._box[tag] = numpy.array(buf, copy=True); return _Req()
    def Recv(self, buf, source=0, tag=0):
        buf[...] = self._box.pop(tag)
COMM_WORLD=_Comm()
'''
SETUP_EXT = '''
from setuptools import setup, Extension
from Cython.Build import cythonize
import numpy
setup(ext_modules=cythonize([Extension("pauxy.estimators.ueg_kernels",
      ["pauxy/estimators/ueg_kernels.pyx"], include_dirs=[numpy.get_include()])]))
'''


def prepare_scratch():
    if not os.path.isdir(os.path.join(SCRATCH, "pauxy")):
        os.makedirs(SCRATCH, exist_ok=True)
        shutil.copytree(REF, os.path.join(SCRATCH, "pauxy"))
        subprocess.check_call(["chmod", "-R", "u+w", SCRATCH])
        for f in ("pauxy/systems/hubbard_holstein.py",):
            subprocess.check_call(["sed", "-i",
                                   r"s/ks\.all() is None/(ks.dtype == object)/; "
                                   r"s/ks\.all() is not None/(ks.dtype != object)/",
                                   os.path.join(SCRATCH, f)])
        for f in ("pauxy/qmc/afqmc.py", "pauxy/qmc/thermal_afqmc.py"):
            subprocess.check_call(["sed", "-i",
                                   r"s/system\.ktwist\.all() is not None/(system.ktwist.dtype != object)/",
                                   os.path.join(SCRATCH, f)])
    for pkg, body in (("h5py", H5PY_STUB), ("mpi4py", "class _rc: recv_mprobe=False\nrc=_rc()\n")):
        d = os.path.join(SCRATCH, "stubs", pkg)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "__init__.py"), "w") as f:
            f.write(body)
    with open(os.path.join(SCRATCH, "stubs", "mpi4py", "MPI.py"), "w") as f:
        f.write(MPI_STUB)
    import glob
    if not glob.glob(os.path.join(SCRATCH, "pauxy/estimators/ueg_kernels*.so")):
        with open(os.path.join(SCRATCH, "setup_ext.py"), "w") as f:
            f.write(SETUP_EXT)
        subprocess.check_call([sys.executable, "setup_ext.py", "build_ext", "--inplace"],
                              cwd=SCRATCH, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, SCRATCH)
    sys.path.insert(0, os.path.join(SCRATCH, "stubs"))


prepare_scratch()

import h5py                                                      # noqa: E402  (stub)
from mpi4py import MPI                                           # noqa: E402  (stub)
from pauxy.systems.generic import Generic                        # noqa: E402
from pauxy.systems.hubbard import Hubbard                        # noqa: E402
from pauxy.systems.ueg import UEG                                # noqa: E402
from pauxy.trial_wavefunction.multi_slater import MultiSlater    # noqa: E402
from pauxy.trial_wavefunction.uhf import UHF                     # noqa: E402
from pauxy.trial_wavefunction.hartree_fock import HartreeFock    # noqa: E402
from pauxy.propagation.continuous import Continuous              # noqa: E402
from pauxy.walkers.single_det import SingleDetWalker             # noqa: E402
from pauxy.estimators.mixed import local_energy                  # noqa: E402
from pauxy.estimators.generic import (                           # noqa: E402
    local_energy_generic_cholesky, local_energy_generic_cholesky_opt)
from pauxy.utils.misc import dotdict                             # noqa: E402
from pauxy.utils.testing import generate_hamiltonian, get_random_nomsd, get_random_phmsd  # noqa: E402
from pauxy.walkers.multi_det import MultiDetWalker               # noqa: E402
from pauxy.qmc.afqmc import AFQMC                                # noqa: E402


OUT = HERE          # --check writes to a scratch directory instead


def save(name, out):
    """One fixture.  Wall-clock entries (the Time column of the block rows, estimates[time]) are zeroed: the fixtures hold
    nothing that differs between two runs of this script."""
    if 'blocks' in out:
        out['blocks'] = numpy.array(out['blocks'])
        out['blocks'][:, -1] = 0
    if 'final_estimates' in out:
        out['final_estimates'] = numpy.array(out['final_estimates'])
        out['final_estimates'][9] = 0
    numpy.savez_compressed(os.path.join(OUT, name), **out)


def rand_phi(M, ne):
    a = numpy.random.rand(M * ne)
    b = numpy.random.rand(M * ne)
    return (a + 1j * b).reshape((M, ne))


def single_walker_ops(system, trial, prop_opts, dt, out, tag):
    """Run every hot-path op of one walker through the reference and record
    inputs + outputs under keys prefixed by ``tag``."""
    qmc = dotdict({'dt': dt, 'nstblz': 5})
    prop = Continuous(system, trial, qmc, options=prop_opts)
    walker = SingleDetWalker(system, trial)
    M, na, nb = system.nbasis, system.nup, system.ndown
    phi = rand_phi(M, na + nb)
    # make it a mild perturbation of the trial so overlaps are well conditioned
    phi = trial.psi + 0.1 * phi
    walker.phi = phi.copy()
    out[tag + 'phi'] = phi
    out[tag + 'psi'] = trial.psi
    out[tag + 'dt'] = dt
    out[tag + 'BH1'] = prop.propagator.BH1
    out[tag + 'mf_shift'] = prop.propagator.mf_shift
    det = walker.greens_function(trial)
    out[tag + 'det'] = det
    out[tag + 'Ghalf_a'] = walker.Gmod[0]
    out[tag + 'Ghalf_b'] = walker.Gmod[1]
    out[tag + 'G'] = walker.G
    if na == nb:
        out[tag + 'ovlp'] = walker.calc_overlap(trial)
    xbar = prop.propagator.construct_force_bias(system, walker, trial)
    out[tag + 'xbar'] = numpy.array(xbar)
    xi = numpy.random.normal(0.0, 1.0, system.nfields)
    out[tag + 'xi'] = xi
    # exercise the clipping branch: blow two force-bias entries up
    xbar_big = numpy.array(xbar, dtype=numpy.complex128)
    xbar_big[0] *= 1e3
    xbar_big[-1] = 3.0 - 4.0j
    out[tag + 'xbar_big'] = xbar_big.copy()
    xb = xbar_big.copy()
    for i in range(system.nfields):
        if numpy.absolute(xb[i]) > 1.0:
            xb[i] /= numpy.absolute(xb[i])
    xs = xi - xb
    out[tag + 'xs_clip'] = xs
    out[tag + 'cmf_clip'] = -prop.sqrt_dt * xs.dot(prop.propagator.mf_shift)
    out[tag + 'cfb_clip'] = xi.dot(xb) - 0.5 * xb.dot(xb)
    xs = xi - xbar
    VHS = prop.propagator.construct_VHS(system, xs)
    out[tag + 'VHS'] = numpy.array(VHS)
    p2 = phi.copy()
    if len(VHS.shape) == 3:
        prop.apply_exponential(p2[:, :na], VHS[0])
        prop.apply_exponential(p2[:, na:], VHS[1])
    else:
        prop.apply_exponential(p2[:, :na], VHS)
        prop.apply_exponential(p2[:, na:], VHS)
    out[tag + 'phi_exp'] = p2
    from pauxy.propagation.operations import kinetic_real
    p3 = phi.copy()
    kinetic_real(p3, system, prop.propagator.BH1)
    out[tag + 'phi_kin'] = p3
    E = walker.local_energy(system, rchol=trial._rchol)
    out[tag + 'energy'] = numpy.array(E)
    # full propagate step + weight update, with the recorded field
    w2 = SingleDetWalker(system, trial)
    w2.phi = phi.copy()
    w2.ot = w2.calc_overlap(trial) if na == nb else w2.greens_function(trial)
    w2.ovlp = w2.ot
    w2.hybrid_energy = 0.25 + 0.1j
    state = numpy.random.get_state()
    _orig = numpy.random.normal
    numpy.random.normal = lambda *a, **k: xi.copy()
    try:
        if na == nb:
            prop.propagate_walker(w2, system, trial, 0.3)
            out[tag + 'step_phi'] = w2.phi.copy()
            out[tag + 'step_weight'] = w2.weight
            out[tag + 'step_ot'] = w2.ot
            out[tag + 'step_ehyb'] = w2.hybrid_energy
    finally:
        numpy.random.normal = _orig
        numpy.random.set_state(state)
    # reortho
    w3 = SingleDetWalker(system, trial)
    w3.phi = phi.copy()
    detR = w3.reortho(trial)
    out[tag + 'phi_qr'] = w3.phi.copy()
    out[tag + 'detR'] = detR
    return prop


def make_generic_ops():
    out = {}
    # case A: the reference's own estimator test system (estimators/tests/test_generic.py:50-64)
    numpy.random.seed(7)
    nmo, nelec = 24, (4, 2)
    h1e, chol, enuc, eri = generate_hamiltonian(nmo, nelec, cplx=False)
    system = Generic(nelec=nelec, h1e=numpy.array([h1e, h1e]),
                     chol=chol.reshape((-1, nmo * nmo)).T.copy(), ecore=enuc)
    wfn = get_random_nomsd(system, ndet=1, cplx=False)
    trial = MultiSlater(system, wfn)
    trial.half_rotate(system)
    e_opt = local_energy_generic_cholesky_opt(system, trial.G, trial.GH, trial.rot_chol())
    e_full = local_energy_generic_cholesky(system, trial.G, Ghalf=trial.GH)
    out['A_h1e'] = h1e
    out['A_chol'] = system.chol_vecs
    out['A_ecore'] = enuc
    out['A_nelec'] = numpy.array(nelec)
    out['A_h1e_mod'] = system.h1e_mod
    out['A_rchol'] = trial._rchol
    out['A_trialG'] = trial.G
    out['A_trialGH_a'] = trial.GH[0]
    out['A_trialGH_b'] = trial.GH[1]
    out['A_e_opt'] = numpy.array(e_opt)
    out['A_e_full'] = numpy.array(e_full)
    # walkers take the trial's dtype (walkers/single_det.py:69-76): run the
    # per-walker ops with the complex128 dtype AFQMC always uses
    trial.psi = trial.psi[0].astype(numpy.complex128)
    trial._rchol = trial._rchol.astype(numpy.complex128)
    single_walker_ops(system, trial, {}, 0.005, out, 'A_')
    # case B: complex trial, equal spins (propagation/tests/test_generic.py system)
    numpy.random.seed(7)
    nmo, nelec = 10, (5, 5)
    h1e, chol, enuc, eri = generate_hamiltonian(nmo, nelec, cplx=False)
    system = Generic(nelec=nelec, h1e=numpy.array([h1e, h1e]),
                     chol=chol.reshape((-1, nmo * nmo)).T.copy(), ecore=enuc)
    wfn = get_random_nomsd(system, ndet=1, cplx=True)
    trial = MultiSlater(system, wfn)
    trial.half_rotate(system)
    out['B_h1e'] = h1e
    out['B_chol'] = system.chol_vecs
    out['B_ecore'] = enuc
    out['B_nelec'] = numpy.array(nelec)
    out['B_h1e_mod'] = system.h1e_mod
    out['B_rchol'] = trial._rchol
    trial.psi = trial.psi[0]
    single_walker_ops(system, trial, {}, 0.005, out, 'B_')
    save('generic_ops.npz', out)


def make_hubbard_ops():
    out = {}
    options = {'nx': 4, 'ny': 4, 'nup': 8, 'ndown': 8, 'U': 4}
    system = Hubbard(inputs=options)
    numpy.random.seed(7)
    uhf = UHF(system, {'ueff': 4.0})
    assert abs(uhf.emin - (-12.56655451978628)) < 1e-8     # trial_wavefunction/tests/test_uhf.py
    wfn = numpy.zeros((1, system.nbasis, system.ne), dtype=numpy.complex128)
    wfn[0] = uhf.psi.copy()
    trial = MultiSlater(system, (numpy.array([1.0 + 0j]), wfn))
    trial.psi = trial.psi[0]
    out['T'] = system.T
    out['U'] = system.U
    out['nx'] = 4
    out['ny'] = 4
    out['nelec'] = numpy.array([8, 8])
    out['h1e_mod'] = system.h1e_mod
    out['uhf_emin'] = uhf.emin
    # propagation/tests/test_hubbard.py:118-136 pinned overlap
    walker = SingleDetWalker(system, trial, nbp=1, nprop_tot=1)
    prop = Continuous(system, trial, dotdict({'dt': 0.01, 'nstblz': 5}),
                      options={'charge_decomposition': False})
    rec = []
    _orig = numpy.random.normal

    def _rec(*a, **k):
        x = _orig(*a, **k)
        rec.append(x.copy())
        return x
    numpy.random.normal = _rec
    try:
        prop.propagate_walker(walker, system, trial, 0.0)
    finally:
        numpy.random.normal = _orig
    assert abs(walker.ovlp.real - 0.765551499039435) < 1e-10
    out['pin_xi'] = rec[0]
    out['pin_ovlp'] = walker.ovlp
    single_walker_ops(system, trial, {'charge_decomposition': True}, 0.01, out, 'C_')
    single_walker_ops(system, trial, {'charge_decomposition': False}, 0.01, out, 'S_')
    save('hubbard_ops.npz', out)


def ueg_arrays(system, out, tag=''):
    out[tag + 'rs'] = system.rs
    out[tag + 'ecut'] = system.ecut
    out[tag + 'nelec'] = numpy.array(system.nelec)
    out[tag + 'sp_eigv'] = system.sp_eigv
    out[tag + 'basis'] = system.basis
    out[tag + 'qvecs'] = system.qvecs
    out[tag + 'vqvec'] = system.vqvec
    out[tag + 'vol'] = system.vol
    out[tag + 'ecore'] = system.ecore
    out[tag + 'h1e_mod_diag'] = numpy.diag(system.h1e_mod[0])
    for name in ('iA', 'iB'):
        m = getattr(system, name).tocoo()
        out[tag + name + '_row'] = m.row
        out[tag + name + '_col'] = m.col
        out[tag + name + '_val'] = m.data
    for name in ('ikpq_i', 'ikpq_kpq', 'ipmq_i', 'ipmq_pmq'):
        lst = getattr(system, name)
        out[tag + name + '_flat'] = numpy.concatenate(lst) if len(lst) else numpy.zeros(0, dtype=numpy.int64)
        out[tag + name + '_off'] = numpy.cumsum([0] + [len(x) for x in lst])


def make_ueg_ops():
    out = {}
    system = UEG(inputs={'rs': 2, 'nup': 7, 'ndown': 7, 'ecut': 2})
    ueg_arrays(system, out)
    occ = numpy.eye(system.nbasis)[:, :system.nup]
    wfn = numpy.zeros((1, system.nbasis, system.nup + system.ndown), dtype=numpy.complex128)
    wfn[0, :, :system.nup] = occ
    wfn[0, :, system.nup:] = occ
    trial = MultiSlater(system, (numpy.array([1 + 0j]), wfn))
    trial.psi = trial.psi[0]
    numpy.random.seed(7)
    prop = single_walker_ops(system, trial, {}, 0.005, out, 'U_')
    # propagation/tests/test_planewave.py:12-38 known answers
    walker = SingleDetWalker(system, trial)
    numpy.random.seed(7)
    walker.phi = rand_phi(system.nbasis, system.nup + system.ndown)
    walker.greens_function(trial)
    fb = prop.propagator.construct_force_bias(system, walker, trial)
    assert abs(numpy.linalg.norm(fb) - 0.16660828645573392) < 1e-12
    xi = numpy.random.rand(system.nfields)
    vhs = prop.propagator.construct_VHS(system, xi - fb)
    assert abs(numpy.linalg.norm(vhs) - 0.1467322554815581) < 1e-12
    out['pw_phi'] = walker.phi
    out['pw_fb'] = numpy.array(fb)
    out['pw_xi'] = xi
    out['pw_vhs'] = numpy.array(vhs)
    save('ueg_ops.npz', out)


def record_trajectory(afqmc, comm, out, r_override=None):
    """Run AFQMC.run while recording the random numbers it draws and the
    walker scalars after every step.  ``r_override`` {step: value}: the comb of that step is handed this uniform instead of
    the one it drew (the draw still happens: the generator's state advances as in any run) -- the comb's r is an input, and
    at the benchmark's time step an unforced r hardly ever lands where a walker is cloned."""
    psi = afqmc.psi
    nw = len(psi.walkers)
    K = afqmc.system.nfields
    nsteps = afqmc.qmc.total_steps
    xi = numpy.full((nsteps + 1, nw, K), numpy.nan)
    rr = numpy.full(nsteps + 1, numpy.nan)
    traj = dict(weight=[], unscaled_weight=[], ot=[], ehyb=[], parent_ix=[], phase=[], eloc=[])
    state = dict(step=0, iw=0)
    out['phi0'] = numpy.array([w.phi for w in psi.walkers])
    _normal, _random = numpy.random.normal, numpy.random.random
    prop = afqmc.propagators
    _pw = prop.propagate_walker

    def normal(*a, **k):
        x = _normal(*a, **k)
        xi[state['step'], state['iw']] = x
        return x

    def random(*a, **k):
        x = _random(*a, **k)
        if r_override and state['step'] in r_override:
            x = r_override[state['step']]
        rr[state['step']] = x
        return x

    def propagate_walker(w, system, trial, eshift):
        state['iw'] = psi.walkers.index(w)
        return _pw(w, system, trial, eshift)

    est_update = afqmc.estimators.update

    def update(system, qmc, trial, psi_, step, fp):
        # called once per step after pop control (qmc/afqmc.py:244)
        traj['weight'].append([w.weight for w in psi_.walkers])
        traj['unscaled_weight'].append([w.unscaled_weight for w in psi_.walkers])
        traj['ot'].append([w.ot for w in psi_.walkers])
        traj['ehyb'].append([w.hybrid_energy for w in psi_.walkers])
        traj['phase'].append([w.phase for w in psi_.walkers])
        traj['eloc'].append([w.eloc for w in psi_.walkers])
        state['step'] = step + 1
        return est_update(system, qmc, trial, psi_, step, fp)

    comm.bcast_log = []
    numpy.random.normal, numpy.random.random = normal, random
    prop.propagate_walker = propagate_walker
    afqmc.estimators.update = update
    state['step'] = 1
    try:
        afqmc.run(comm=comm, verbose=0)
    finally:
        numpy.random.normal, numpy.random.random = _normal, _random
        prop.propagate_walker = _pw
        afqmc.estimators.update = est_update
    pix = [d['ix'] for d in comm.bcast_log if isinstance(d, dict) and 'ix' in d]
    out['xi'] = xi[1:]
    out['r'] = rr[1:]
    out['weight'] = numpy.array(traj['weight'], dtype=numpy.float64)
    out['unscaled_weight'] = numpy.array(traj['unscaled_weight'], dtype=numpy.float64)
    out['ot'] = numpy.array(traj['ot'], dtype=numpy.complex128)
    out['ehyb'] = numpy.array(traj['ehyb'], dtype=numpy.complex128)
    out['phase'] = numpy.array(traj['phase'], dtype=numpy.complex128)
    out['eloc'] = numpy.array(traj['eloc'], dtype=numpy.complex128)
    out['parent_ix'] = numpy.array(pix, dtype=numpy.int32).reshape(len(pix), -1)
    store = h5py._STORE[afqmc.estimators.filename]
    keys = sorted(k for k in store if k.startswith('basic/energies/'))
    out['blocks'] = numpy.array([store[k] for k in keys])
    out['dt'] = afqmc.qmc.dt
    out['nsteps'] = afqmc.qmc.nsteps
    out['nblocks'] = afqmc.qmc.nblocks
    out['nstblz'] = afqmc.qmc.nstblz
    out['npop_control'] = afqmc.qmc.npop_control
    out['energy_eval_freq'] = afqmc.estimators.estimators['mixed'].energy_eval_freq
    out['psi'] = afqmc.trial.psi
    out['BH1'] = prop.propagator.BH1
    out['mf_shift'] = prop.propagator.mf_shift
    out['nelec'] = numpy.array([afqmc.system.nup, afqmc.system.ndown])
    out['nfb_trig'] = prop.nfb_trig
    out['nhe_trig'] = prop.nhe_trig
    # final-state estimator pass the reference's driver tests pin
    mixed = afqmc.estimators.estimators['mixed']
    mixed.update(afqmc.system, afqmc.qmc, afqmc.trial, afqmc.psi, 0, prop.free_projection)
    out['final_estimates'] = mixed.estimates.copy()
    out['final_phi'] = numpy.array([w.phi for w in psi.walkers])


def make_traj_generic():
    # qmc/tests/test_afqmc.py:190-229 (single-determinant RHF trial)
    out = {}
    nmo, nelec = 11, (3, 3)
    options = {'verbosity': 0, 'get_sha1': False,
               'qmc': {'timestep': 0.005, 'steps': 10, 'blocks': 10, 'rng_seed': 8},
               'estimates': {'mixed': {'energy_eval_freq': 1}},
               'trial': {'name': 'MultiSlater'}}
    numpy.random.seed(7)
    h1e, chol, enuc, eri = generate_hamiltonian(nmo, nelec, cplx=False)
    system = Generic(nelec=nelec, h1e=numpy.array([h1e, h1e]),
                     chol=chol.reshape((-1, nmo * nmo)).T.copy(), ecore=enuc)
    comm = MPI.COMM_WORLD
    afqmc = AFQMC(comm=comm, system=system, options=options)
    out['h1e'] = h1e
    out['chol'] = system.chol_vecs
    out['ecore'] = enuc
    out['rchol'] = afqmc.trial._rchol
    record_trajectory(afqmc, comm, out)
    numer = out['final_estimates'][2]
    assert abs(numer.real - 3.8763193646854273) < 1e-9, numer
    save('traj_generic.npz', out)


def c3_inputs(M=100, K=500, seed=7):
    """The synthetic generic Hamiltonian of SURVEY 8(d) / BASELINE configs[2], restated here with numpy only (the package's
    own pauxy_amd.systems.synthetic_generic is held to the checksums this generator stores, not used to make them):
    h1 = (R + R^T) / 2, R ~ U(0, 1); L_n = (A_n + A_n^T) / 2, A_n ~ N(0, (0.1 / M)^2); RandomState(seed)."""
    rng = numpy.random.RandomState(seed)
    R = rng.random_sample((M, M))
    h1 = 0.5 * (R + R.T)
    A = rng.normal(scale=0.1 / M, size=(K, M, M))
    L = 0.5 * (A + A.transpose(0, 2, 1))
    chol = numpy.ascontiguousarray(L.reshape(K, M * M).T)
    return h1, chol


def c3_open_walkers(phi, nopen, seed, na, eps=1e-3):
    """Walkers 0, 3, 6, ... (``nopen`` of them) get their beta block perturbed: open-shell walkers beside closed ones."""
    rng = numpy.random.RandomState(seed)
    ix = [3 * i for i in range(nopen)]
    for i in ix:
        phi[i][:, na:] += eps * (rng.rand(*phi[i][:, na:].shape) + 1j * rng.rand(*phi[i][:, na:].shape))
    return ix


def make_traj_generic_c3(name='traj_generic_c3.npz', nwalkers=36, nopen=0, blocks=3, r_override=None):
    """BASELINE configs[2] at its own size through the GENUINE driver: M = 100, K = 500, 25 + 25 electrons, RHF trial,
    walkers starting at the trial (closed-shell: spin blocks bitwise equal), 10 steps per block, re-orthogonalisation every
    10, comb every 5, energy every 10 -- the cadence bench.py times.  The inputs regenerate from seed 7 and the fields from
    the recorded MT19937 state, so the fixture holds seeds, checksums and the reference's outputs only.
    ``nopen`` > 0: every third walker starts with a perturbed beta block (c3_open_walkers)."""
    out = {}
    M, K, nelec = 100, 500, (25, 25)
    h1, chol = c3_inputs(M, K, 7)
    system = Generic(nelec=nelec, h1e=numpy.array([h1, h1]), chol=chol, ecore=0.0)
    e, v = numpy.linalg.eigh(h1)
    psi = numpy.zeros((M, 50), dtype=numpy.complex128)
    psi[:, :25] = v[:, :25]
    psi[:, 25:] = v[:, :25]
    trial = MultiSlater(system, (numpy.array([1.0 + 0j]), psi[None].copy()))
    trial.half_rotate(system, None)                    # trial_wavefunction/utils.py:76-77
    options = {'verbosity': 0, 'get_sha1': False,
               'qmc': {'timestep': 0.005, 'steps': 10, 'blocks': blocks, 'rng_seed': 8, 'nwalkers': nwalkers,
                       'pop_control_freq': 5, 'stabilise_freq': 10},
               'estimates': {'mixed': {}},
               }
    comm = MPI.COMM_WORLD
    afqmc = AFQMC(comm=comm, system=system, trial=trial, options=options)
    assert len(afqmc.psi.walkers) == nwalkers
    if nopen:
        phis = [w.phi for w in afqmc.psi.walkers]
        out['open_ix'] = numpy.array(c3_open_walkers(phis, nopen, 21, 25), dtype=numpy.int32)
        out['open_seed'] = 21
        for i in out['open_ix']:
            w = afqmc.psi.walkers[i]
            w.inverse_overlap(afqmc.trial)
            w.ot = w.calc_otrial(afqmc.trial)
            w.ovlp = w.ot
            w.greens_function(afqmc.trial)
            w.le_oratio = 1.0
    out['rng_state_keys'] = numpy.array(numpy.random.get_state()[1], dtype=numpy.uint32)
    st = numpy.random.get_state()
    out['rng_state_rest'] = numpy.array([st[2], st[3]], dtype=numpy.int64)
    out['rng_state_gauss'] = float(st[4])
    record_trajectory(afqmc, comm, out, r_override)
    out['r_override_steps'] = numpy.array(sorted(r_override or {}), dtype=numpy.int64)
    xi = out.pop('xi')
    out['xi_sum'] = xi.sum(axis=2)                     # [step, walker]: holds the regenerated stream to the recorded one
    out['xi_first'] = xi[:, :, 0].copy()
    out['seed'] = 7
    out['M'], out['K'] = M, K
    phi0 = out.pop('phi0')
    out['phi0_sum'] = phi0.sum(axis=(1, 2))
    out['nwalkers'] = nwalkers
    fp = out.pop('final_phi')
    out['final_phi_colnorm'] = numpy.linalg.norm(fp, axis=1)      # [walker, column]
    out['final_phi_sum'] = fp.sum(axis=1)
    # checksums of what the test regenerates with the package's own set-up code
    for key, arr in (('psi', out.pop('psi')), ('BH1', out.pop('BH1')), ('mf_shift', out.pop('mf_shift'))):
        arr = numpy.asarray(arr)
        out[key + '_abs_sum'] = numpy.abs(arr).sum()
        out[key + '_sum'] = arr.sum()
    rchol = numpy.asarray(afqmc.trial._rchol)
    out['rchol_abs_sum'] = numpy.abs(rchol).sum()
    out['rchol_shape'] = numpy.array(rchol.shape)
    out['h1_sum'], out['chol_abs_sum'] = h1.sum(), numpy.abs(chol).sum()
    save(name, out)


def make_traj_hubbard(name, nup, pin=None, nwalkers=10, npop=1, blocks=10, prop_extra=None):
    # qmc/tests/test_afqmc.py:145-188 (continuous HS, UHF trial)
    out = {}
    options = {'verbosity': 0, 'get_sha1': False,
               'qmc': {'timestep': 0.01, 'num_steps': 10, 'blocks': blocks, 'rng_seed': 8,
                       'num_walkers': nwalkers, 'pop_control_freq': npop},
               'model': {'name': "Hubbard", 'nx': 4, 'ny': 4, 'nup': nup, "U": 4, 'ndown': nup},
               'trial': {'name': 'UHF'},
               'estimates': {'mixed': {'energy_eval_freq': 1}},
               'propagator': {'hubbard_stratonovich': 'continuous'}}
    if prop_extra:
        options['propagator'].update(prop_extra)
    comm = MPI.COMM_WORLD
    afqmc = AFQMC(comm=comm, options=options)
    out['free_projection'] = bool(afqmc.propagators.free_projection)
    out['hybrid'] = bool(afqmc.propagators.hybrid)
    out['T'] = afqmc.system.T
    out['U'] = afqmc.system.U
    record_trajectory(afqmc, comm, out)
    if pin is not None:
        numer = out['final_estimates'][2]
        assert abs(numer.real - pin) < 1e-8, numer
    save(name, out)


def make_traj_ueg():
    # qmc/tests/test_afqmc.py:49-97
    out = {}
    options = {'verbosity': 0, 'get_sha1': False,
               'qmc': {'timestep': 0.01, 'num_steps': 10, 'blocks': 5, 'rng_seed': 8},
               'model': {'name': "UEG", 'rs': 2.44, 'ecut': 2, 'nup': 7, 'ndown': 7},
               'estimates': {'mixed': {'energy_eval_freq': 1}},
               'trial': {'name': 'hartree_fock'}}
    comm = MPI.COMM_WORLD
    afqmc = AFQMC(comm=comm, options=options)
    ueg_arrays(afqmc.system, out, 'sys_')
    record_trajectory(afqmc, comm, out)
    numer = out['final_estimates'][2]
    assert abs(numer.real - 16.33039729324558) < 1e-9, numer
    assert abs(out['final_estimates'][0].real - 9.75405059997262) < 1e-9
    save('traj_ueg.npz', out)



def make_traj_bp_ueg(name='traj_bp_ueg.npz'):
    """Back-propagated one-body RDM for the UEG (estimators/back_propagation.py:127-226 with
    propagation/planewave.py:141-178), same model as traj_ueg.npz, tau_bp = 4 steps."""
    out = {}
    options = {'verbosity': 0, 'get_sha1': False,
               'qmc': {'timestep': 0.01, 'num_steps': 10, 'blocks': 4, 'rng_seed': 8},
               'model': {'name': "UEG", 'rs': 2.44, 'ecut': 2, 'nup': 7, 'ndown': 7},
               'estimates': {'mixed': {'energy_eval_freq': 1}, 'back_propagated': {'tau_bp': 0.04, 'one_rdm': True}},
               'trial': {'name': 'hartree_fock'}}
    comm = MPI.COMM_WORLD
    afqmc = AFQMC(comm=comm, options=options)
    assert afqmc.estimators.back_propagation and afqmc.estimators.nbp == 4
    ueg_arrays(afqmc.system, out, 'sys_')
    out['nbp'] = afqmc.estimators.nbp
    record_trajectory(afqmc, comm, out)
    store = h5py._STORE[afqmc.estimators.filename]
    dk = sorted((k for k in store if k.startswith('back_propagated/denominator_4/')), key=lambda k: int(k.rsplit('/', 1)[1]))
    rk = sorted((k for k in store if k.startswith('back_propagated/one_rdm_4/')), key=lambda k: int(k.rsplit('/', 1)[1]))
    assert len(dk) == len(rk) and len(dk) > 0
    out['bp_denominator'] = numpy.array([store[k] for k in dk]).reshape(len(dk))
    out['bp_one_rdm'] = numpy.array([store[k] for k in rk])
    save(name, out)


def make_traj_mixed_rdm(name='traj_hubbard_rdm.npz'):
    """Mixed estimator with one_rdm: True (estimators/mixed.py:226-233,279-283): 4x4 U=4 Hubbard, continuous HS,
    10 walkers, comb every 5 steps, energy every 5 steps (so that the accumulated w.G is stale on the steps between:
    it is the Green's function of the walker BEFORE that step's propagation, continuous.py:245)."""
    out = {}
    options = {'verbosity': 0, 'get_sha1': False,
               'qmc': {'timestep': 0.01, 'num_steps': 10, 'blocks': 4, 'rng_seed': 8,
                       'num_walkers': 10, 'pop_control_freq': 5},
               'model': {'name': "Hubbard", 'nx': 4, 'ny': 4, 'nup': 8, "U": 4, 'ndown': 8},
               'trial': {'name': 'UHF'},
               'estimates': {'mixed': {'energy_eval_freq': 5, 'one_rdm': True}},
               'propagator': {'hubbard_stratonovich': 'continuous'}}
    comm = MPI.COMM_WORLD
    afqmc = AFQMC(comm=comm, options=options)
    out['T'] = afqmc.system.T
    out['U'] = afqmc.system.U
    record_trajectory(afqmc, comm, out)
    store = h5py._STORE[afqmc.estimators.filename]
    rk = sorted((k for k in store if k.startswith('basic/one_rdm/')), key=lambda k: int(k.rsplit('/', 1)[1]))
    assert len(rk) > 0
    out['mixed_one_rdm'] = numpy.array([store[k] for k in rk])
    save(name, out)


def make_traj_log_shift(name='traj_hubbard_logshift.npz'):
    """walkers/handler.py:228,456-475 (use_log_shift: True): running-average shifts of log|ot|, log detR applied in
    walkers/single_det.py:159,192,250,320; same 4x4 model, comb every 5 steps, reortho every 5."""
    out = {}
    options = {'verbosity': 0, 'get_sha1': False,
               'qmc': {'timestep': 0.01, 'num_steps': 10, 'blocks': 4, 'rng_seed': 8,
                       'num_walkers': 10, 'pop_control_freq': 5, 'stabilise_freq': 5},
               'model': {'name': "Hubbard", 'nx': 4, 'ny': 4, 'nup': 8, "U": 4, 'ndown': 8},
               'trial': {'name': 'UHF'},
               'walkers': {'use_log_shift': True},
               'estimates': {'mixed': {'energy_eval_freq': 1}},
               'propagator': {'hubbard_stratonovich': 'continuous'}}
    comm = MPI.COMM_WORLD
    afqmc = AFQMC(comm=comm, options=options)
    assert afqmc.psi.use_log_shift
    out['T'] = afqmc.system.T
    out['U'] = afqmc.system.U
    record_trajectory(afqmc, comm, out)
    out['final_log_shift'] = numpy.array([w.log_shift for w in afqmc.psi.walkers])
    out['final_detR_shift'] = numpy.array([w.detR_shift for w in afqmc.psi.walkers])
    save(name, out)


def msd_system(out, tag=''):
    """propagation/tests/test_generic.py:52-62: 10 orbitals, 5+5 electrons, seed 7."""
    numpy.random.seed(7)
    nmo, nelec = 10, (5, 5)
    h1e, chol, enuc, eri = generate_hamiltonian(nmo, nelec, cplx=False)
    system = Generic(nelec=nelec, h1e=numpy.array([h1e, h1e]),
                     chol=chol.reshape((-1, nmo * nmo)).T.copy(), ecore=0)
    out[tag + 'h1e'] = h1e
    out[tag + 'chol'] = system.chol_vecs
    out[tag + 'ecore'] = 0.0
    out[tag + 'nelec'] = numpy.array(nelec)
    return system


def msd_record_trial(out, tag, trial, prop):
    out[tag + 'psi'] = numpy.array(trial.psi)
    out[tag + 'coeffs'] = numpy.array(trial.coeffs)
    out[tag + 'init'] = numpy.array(trial.init)
    out[tag + 'ortho'] = bool(trial.ortho_expansion)
    out[tag + 'BH1'] = prop.propagator.BH1
    out[tag + 'mf_shift'] = prop.propagator.mf_shift


def msd_steps(system, trial, hybrid, out, tag, nsteps=10, eshift=None):
    """The loop of propagation/tests/test_generic.py:52-92 with the drawn fields recorded."""
    qmc = dotdict({'dt': 0.005, 'nstblz': 5})
    prop = Continuous(system, trial, qmc, options={'hybrid': hybrid})
    walker = MultiDetWalker(system, trial)
    msd_record_trial(out, tag, trial, prop)
    out[tag + 'phi0'] = walker.phi.copy()
    out[tag + 'ot0'] = walker.ot
    out[tag + 'ovlps0'] = walker.ovlps.copy()
    out[tag + 'weights0'] = walker.weights.copy()
    out[tag + 'Gi0'] = walker.Gi.copy()
    out[tag + 'xbar0'] = numpy.array(prop.propagator.construct_force_bias(system, walker, trial))
    out[tag + 'energy0'] = numpy.array(walker.local_energy(system))
    xi = []
    _normal = numpy.random.normal

    def normal(*a, **k):
        x = _normal(*a, **k)
        xi.append(numpy.array(x))
        return x

    rec = dict(phi=[], weight=[], ot=[], ehyb=[], eloc=[])
    numpy.random.normal = normal
    try:
        for i in range(nsteps):
            prop.propagate_walker(walker, system, trial, eshift)
            rec['phi'].append(walker.phi.copy())
            rec['weight'].append(walker.weight)
            rec['ot'].append(walker.ot)
            rec['ehyb'].append(walker.hybrid_energy)
            rec['eloc'].append(walker.eloc)
    finally:
        numpy.random.normal = _normal
    out[tag + 'xi'] = numpy.array(xi)
    out[tag + 'eshift'] = eshift
    out[tag + 'step_phi'] = numpy.array(rec['phi'])
    out[tag + 'step_weight'] = numpy.array(rec['weight'], dtype=numpy.float64)
    out[tag + 'step_ot'] = numpy.array(rec['ot'], dtype=numpy.complex128)
    out[tag + 'step_ehyb'] = numpy.array(rec['ehyb'], dtype=numpy.complex128)
    out[tag + 'step_eloc'] = numpy.array(rec['eloc'], dtype=numpy.complex128)
    detR = walker.reortho(trial)
    out[tag + 'phi_qr'] = walker.phi.copy()
    out[tag + 'detR'] = detR
    return walker


def make_msd_ops():
    """Multi-determinant trials: the reference's two pinned PHMSD runs and a
    non-orthogonal (NOMSD) expansion on the same Hamiltonian."""
    out = {}
    # P_: particle-hole expansion, hybrid False (test_generic.py:52-72, weight 0.68797524675701)
    system = msd_system(out)
    wfn, init = get_random_phmsd(system, ndet=3, init=True)
    trial = MultiSlater(system, wfn, init=init)
    trial.calculate_energy(system)
    w = msd_steps(system, trial, False, out, 'PL_', eshift=trial.energy)
    assert abs(w.weight - 0.68797524675701) < 1e-12, w.weight
    # same system and trial, hybrid True (test_generic.py:74-92, weight 0.7430443466368197)
    system = msd_system({})
    wfn, init = get_random_phmsd(system, ndet=3, init=True)
    trial = MultiSlater(system, wfn, init=init)
    trial.calculate_energy(system)
    w = msd_steps(system, trial, True, out, 'PH_', eshift=trial.energy)
    assert abs(w.weight - 0.7430443466368197) < 1e-12, w.weight
    # N_: non-orthogonal complex expansion, walker started from a perturbed first determinant
    system = msd_system({})
    coeffs, wfn = get_random_nomsd(system, ndet=3, cplx=True)
    # determinants that overlap well with each other: common reference + complex noise
    e, v = numpy.linalg.eigh(system.H1[0])
    ref = numpy.concatenate([v[:, :5], v[:, :5]], axis=1)
    wfn = ref[None] + 0.15 * wfn
    init = ref + 0.1 * rand_phi(10, 10)
    trial = MultiSlater(system, (coeffs, wfn), init=init)
    msd_steps(system, trial, True, out, 'N_', eshift=0.3)
    save('msd_ops.npz', out)


def make_traj_msd():
    """The unchanged driver (qmc/afqmc.py) with a 3-determinant NOMSD trial: 8 walkers, comb every
    5 steps, re-orthogonalisation every 5, energy every step."""
    out = {}
    system = msd_system(out)
    coeffs, wfn = get_random_nomsd(system, ndet=3, cplx=True)
    e, v = numpy.linalg.eigh(system.H1[0])
    ref = numpy.concatenate([v[:, :5], v[:, :5]], axis=1)
    wfn = ref[None] + 0.15 * wfn
    trial = MultiSlater(system, (coeffs, wfn), init=ref.astype(numpy.complex128))
    options = {'verbosity': 0, 'get_sha1': False,
               'qmc': {'timestep': 0.005, 'steps': 5, 'blocks': 8, 'rng_seed': 8, 'nwalkers': 8,
                       'pop_control_freq': 5, 'stabilise_freq': 5},
               'estimates': {'mixed': {'energy_eval_freq': 1}}}
    comm = MPI.COMM_WORLD
    afqmc = AFQMC(comm=comm, system=system, trial=trial, options=options)
    out['coeffs'] = numpy.array(trial.coeffs)
    record_trajectory(afqmc, comm, out)
    out['psi'] = numpy.array(trial.psi)
    save('traj_msd.npz', out)



def make_traj_bp(name='traj_bp.npz', restore_weights=None, blocks=10, energy=False, nsplit=1, tau_bp=0.025):
    """qmc/tests/test_afqmc.py:232-278: single-determinant generic run with the back-propagated
    one-body RDM (tau_bp = 5 steps); pinned rdm[11,0,1,3].real == -0.121883381144845."""
    out = {}
    nmo, nelec = 11, (3, 3)
    bp = {'tau_bp': tau_bp, 'one_rdm': True}
    if nsplit != 1:
        bp['nsplit'] = nsplit
    if restore_weights is not None:
        bp['restore_weights'] = restore_weights
    if energy:
        bp['evaluate_energy'] = True
    options = {'verbosity': 0, 'get_sha1': False,
               'qmc': {'timestep': 0.005, 'num_steps': 10, 'blocks': blocks, 'rng_seed': 8},
               'trial': {'name': 'MultiSlater'},
               'estimator': {'back_propagated': bp, 'mixed': {'energy_eval_freq': 1}}}
    numpy.random.seed(7)
    h1e, chol, enuc, eri = generate_hamiltonian(nmo, nelec, cplx=False)
    system = Generic(nelec=nelec, h1e=numpy.array([h1e, h1e]),
                     chol=chol.reshape((-1, nmo * nmo)).T.copy(), ecore=enuc)
    comm = MPI.COMM_WORLD
    afqmc = AFQMC(comm=comm, system=system, options=options)
    assert afqmc.estimators.back_propagation and afqmc.estimators.nbp == int(round(tau_bp / 0.005))
    out['h1e'] = h1e
    out['chol'] = system.chol_vecs
    out['ecore'] = enuc
    out['rchol'] = afqmc.trial._rchol
    out['nbp'] = afqmc.estimators.nbp
    out['restore_weights'] = '' if restore_weights is None else restore_weights
    record_trajectory(afqmc, comm, out)
    store = h5py._STORE[afqmc.estimators.filename]
    nbp = afqmc.estimators.nbp
    if nsplit != 1:
        # one output group per split length (back_propagation.py:288-324)
        splits = [int(x) for x in afqmc.estimators.estimators['back_prop'].splits]
        out['splits'] = numpy.array(splits)
        for sp in splits:
            dk = sorted((k for k in store if k.startswith('back_propagated/denominator_%d/' % sp)), key=lambda k: int(k.rsplit('/', 1)[1]))
            rk = sorted((k for k in store if k.startswith('back_propagated/one_rdm_%d/' % sp)), key=lambda k: int(k.rsplit('/', 1)[1]))
            assert len(dk) == len(rk) and len(dk) > 0, sp
            out['bp_denominator_%d' % sp] = numpy.array([store[k] for k in dk]).reshape(len(dk))
            out['bp_one_rdm_%d' % sp] = numpy.array([store[k] for k in rk])
        save(name, out)
        return
    dk = sorted(k for k in store if k.startswith('back_propagated/denominator_%d/' % nbp))
    rk = sorted(k for k in store if k.startswith('back_propagated/one_rdm_%d/' % nbp))
    assert len(dk) == len(rk) and len(dk) > 0
    out['bp_denominator'] = numpy.array([store[k] for k in dk]).reshape(len(dk))
    out['bp_one_rdm'] = numpy.array([store[k] for k in rk])
    ek = sorted(k for k in store if k.startswith('back_propagated/energies_%d/' % nbp))
    if ek:
        out['bp_energies'] = numpy.array([store[k] for k in ek])
    rdm = out['bp_one_rdm'] / out['bp_denominator'][:, None, None, None]
    if restore_weights is None and blocks == 10:
        assert abs(rdm[0, 0].trace() - nelec[0]) < 1e-10 and abs(rdm[0, 1].trace() - nelec[1]) < 1e-10
        assert abs(rdm[11, 0, 1, 3].real - (-0.121883381144845)) < 1e-10, rdm[11, 0, 1, 3]
        numer = out['final_estimates'][2]
        assert abs(numer.real - 3.8763193646854273) < 1e-9, numer
    save(name, out)



def make_traj_hirsch(name='traj_hubbard_hirsch.npz', charge=False, blocks=10, pin=None, mean_pin=None, walkers=None,
                     bp=None, prop_extra=None):
    """qmc/tests/test_afqmc.py:99-143: discrete Hirsch HS (single-site updates, propagation/hubbard.py:12-343),
    4x4 U=4 with 7+7 electrons, UHF trial.  Every uniform the run draws (one per site per live walker, then the
    comb's) is recorded per step."""
    out = {}
    prop = {'hubbard_stratonovich': 'discrete'}
    if charge:
        prop['charge_decomposition'] = True
    prop.update(prop_extra or {})
    options = {'verbosity': 0, 'get_sha1': False,
               'qmc': {'timestep': 0.01, 'num_steps': 10, 'blocks': blocks, 'rng_seed': 8},
               'model': {'name': "Hubbard", 'nx': 4, 'ny': 4, 'nup': 7, "U": 4, 'ndown': 7},
               'trial': {'name': 'UHF'},
               'estimates': {'mixed': {'energy_eval_freq': 1}},
               'propagator': prop}
    if walkers:
        options['walkers'] = walkers
    if bp:
        options['estimates']['back_propagated'] = bp
    comm = MPI.COMM_WORLD
    afqmc = AFQMC(comm=comm, options=options)
    psi = afqmc.psi
    out['T'] = afqmc.system.T
    out['U'] = afqmc.system.U
    out['charge'] = bool(charge)
    out['phi0'] = numpy.array([w.phi for w in psi.walkers])
    out['psi'] = afqmc.trial.psi
    out['bt2'] = afqmc.propagators.bt2
    out['nelec'] = numpy.array([afqmc.system.nup, afqmc.system.ndown])
    draws = []
    cur = []
    traj = dict(weight=[], unscaled_weight=[], ot=[], phase=[])
    _random = numpy.random.random

    def random(*a, **k):
        x = _random(*a, **k)
        cur.append(x)
        return x

    est_update = afqmc.estimators.update

    def update(system, qmc, trial, psi_, step, fp):
        traj['weight'].append([w.weight for w in psi_.walkers])
        traj['unscaled_weight'].append([w.unscaled_weight for w in psi_.walkers])
        traj['ot'].append([w.ot for w in psi_.walkers])
        traj['phase'].append([w.phase for w in psi_.walkers])
        draws.append(numpy.array(cur))
        del cur[:]
        return est_update(system, qmc, trial, psi_, step, fp)

    comm.bcast_log = []
    numpy.random.random = random
    afqmc.estimators.update = update
    try:
        afqmc.run(comm=comm, verbose=0)
    finally:
        numpy.random.random = _random
        afqmc.estimators.update = est_update
    pix = [d['ix'] for d in comm.bcast_log if isinstance(d, dict) and 'ix' in d]
    out['u'] = numpy.concatenate(draws)
    out['u_off'] = numpy.concatenate([[0], numpy.cumsum([len(d) for d in draws])])
    out['weight'] = numpy.array(traj['weight'], dtype=numpy.float64)
    out['unscaled_weight'] = numpy.array(traj['unscaled_weight'], dtype=numpy.float64)
    out['ot'] = numpy.array(traj['ot'], dtype=numpy.complex128)
    out['phase'] = numpy.array(traj['phase'], dtype=numpy.complex128)
    out['free_projection'] = bool(afqmc.propagators.free_projection)
    out['single_site'] = bool(prop.get('single_site_update', True))
    out['parent_ix'] = numpy.array(pix, dtype=numpy.int32).reshape(len(pix), -1)
    store = h5py._STORE[afqmc.estimators.filename]
    keys = sorted(k for k in store if k.startswith('basic/energies/'))
    out['blocks'] = numpy.array([store[k] for k in keys])
    out['dt'] = afqmc.qmc.dt
    out['nsteps'] = afqmc.qmc.nsteps
    out['nblocks'] = afqmc.qmc.nblocks
    out['nstblz'] = afqmc.qmc.nstblz
    out['npop_control'] = afqmc.qmc.npop_control
    out['energy_eval_freq'] = afqmc.estimators.estimators['mixed'].energy_eval_freq
    mixed = afqmc.estimators.estimators['mixed']
    mixed.update(afqmc.system, afqmc.qmc, afqmc.trial, afqmc.psi, 0, afqmc.propagators.free_projection)
    out['final_estimates'] = mixed.estimates.copy()
    out['final_phi'] = numpy.array([w.phi for w in psi.walkers])
    if bp:
        nbp = afqmc.estimators.nbp
        out['nbp'] = nbp
        dk = sorted((k for k in store if k.startswith('back_propagated/denominator_%d/' % nbp)), key=lambda k: int(k.rsplit('/', 1)[1]))
        rk = sorted((k for k in store if k.startswith('back_propagated/one_rdm_%d/' % nbp)), key=lambda k: int(k.rsplit('/', 1)[1]))
        assert len(dk) == len(rk) and len(dk) > 0
        out['bp_denominator'] = numpy.array([store[k] for k in dk]).reshape(len(dk))
        out['bp_one_rdm'] = numpy.array([store[k] for k in rk])
    out['final_log_shift'] = numpy.array([w.log_shift for w in psi.walkers])
    out['final_detR_shift'] = numpy.array([w.detR_shift for w in psi.walkers])
    if pin is not None:
        assert abs(out['final_estimates'][2].real - pin) < 1e-8, out['final_estimates'][2]
    if mean_pin is not None:
        et = out['blocks'][:, 5]          # ETotal column of the basic/energies rows
        assert abs(numpy.mean(et[:-1]).real - mean_pin) < 1e-9, numpy.mean(et[:-1])
    save(name, out)


def make_io():
    """On-disk formats: what the reference's writers put under each dataset name (captured
    through the in-memory h5py stand-in) and what its readers return, for small random inputs."""
    from pauxy.utils import io as rio
    out = {}
    h5py._STORE.clear()            # the in-memory "files" of whatever ran before in this process are not this fixture's
    rng = numpy.random.RandomState(7)
    M, K, nelec = 5, 7, (2, 2)
    h = rng.rand(M, M)
    h = 0.5 * (h + h.T)
    L = rng.rand(K, M, M) - 0.5
    L = 0.5 * (L + L.transpose(0, 2, 1))
    L[abs(L) < 0.15] = 0.0
    chol = L.reshape(K, M * M).T.copy()
    a = rng.rand(M, M)
    hc = h + 1j * 0.3 * (a - a.T)
    cholc = chol * (1.0 + 0.0j) + 1j * 0.1 * chol[::-1]
    X = rng.rand(M, M)
    out.update(h=h, chol=chol, hc=hc, cholc=cholc, X=X, nelec=numpy.array(nelec), enuc=1.25)
    rio.write_qmcpack_dense(h, chol, nelec, M, enuc=1.25, filename='dense_real.h5', real_chol=True)
    rio.write_qmcpack_dense(hc, cholc, nelec, M, enuc=1.25, filename='dense_cplx.h5', real_chol=False, ortho=X)
    rio.write_qmcpack_sparse(hc, cholc, nelec, M, enuc=1.25, filename='sparse_cplx.h5', real_chol=False)
    rio.write_qmcpack_sparse(h, chol, nelec, M, enuc=1.25, filename='sparse_real.h5', real_chol=True)
    for f in ('dense_real.h5', 'dense_cplx.h5'):
        r = rio.from_qmcpack_dense(f)
        out[f + '|read|hcore'], out[f + '|read|chol'] = numpy.array(r[0]), numpy.array(r[1])
        out[f + '|read|scalars'] = numpy.array([r[2], r[3], r[4], r[5]], dtype=float)
    for f in ('sparse_real.h5', 'sparse_cplx.h5'):
        r = rio.from_qmcpack_sparse(f)
        out[f + '|read|hcore'], out[f + '|read|chol'] = numpy.array(r[0]), r[1].toarray()
        out[f + '|read|scalars'] = numpy.array([r[2], r[3], r[4], r[5]], dtype=float)
    # wavefunctions
    na, nb = nelec
    coeffs = rng.rand(3) + 1j * rng.rand(3)
    dets = rng.rand(3, M, na + nb) + 1j * rng.rand(3, M, na + nb)
    dets[abs(dets.real) < 0.2] = 0.0
    init = [rng.rand(M, na) + 1j * rng.rand(M, na), rng.rand(M, nb) + 1j * rng.rand(M, nb)]
    occa = numpy.array([[0, 1], [0, 2], [1, 3]])
    occb = numpy.array([[0, 1], [1, 2], [0, 4]])
    out.update(coeffs=coeffs, dets=dets, init_a=init[0], init_b=init[1], occa=occa, occb=occb)
    rio.write_qmcpack_wfn('nomsd_uhf.h5', (coeffs.copy(), dets.copy()), 'uhf', nelec, M)
    rio.write_qmcpack_wfn('nomsd_rhf.h5', (coeffs.copy(), dets.copy()), 'rhf', nelec, M)
    rio.write_qmcpack_wfn('nomsd_init.h5', (coeffs.copy(), dets.copy()), 'uhf', nelec, M, init=init)
    rio.write_qmcpack_wfn('phmsd.h5', (coeffs.copy(), occa, occb), 'uhf', nelec, M)
    rio.write_qmcpack_wfn('phmsd_init.h5', (coeffs.copy(), occa, occb), 'uhf', nelec, M, init=init)
    for f in ('nomsd_uhf.h5', 'nomsd_rhf.h5', 'nomsd_init.h5', 'phmsd.h5', 'phmsd_init.h5'):
        wfn, psi0 = rio.read_qmcpack_wfn_hdf(f)
        for i, x in enumerate(wfn):
            out[f + '|read|wfn%d' % i] = numpy.array(x)
        out[f + '|read|psi0'] = psi0
    for fname, store in h5py._STORE.items():
        if not fname.endswith('.h5') or '|' in fname:
            continue
        for path, arr in store.items():
            if arr is not None:
                out[fname + '|' + path] = arr
    save('io_formats.npz', out)
    print('io_formats.npz: %d arrays' % len(out))



# --------------------------------------------------------------------------------------------------------------
# dropin: the GENUINE driver (pauxy/qmc/afqmc.py, unchanged but for the three plug-in imports INTEGRATION.md section 3
# names) over the pauxy_amd plug-in classes.  There is no GPU here, so the classes sit on the test-only numpy stand-in
# for AfqDevice (tests/oracle_device.py, oracle arithmetic); everything the driver touches -- constructors,
# AFQMC.__init__'s to_json/serialise walk, run(), finalise() -- is the reference's own code.  Outputs:
#   * asserts that the run reproduces the committed golden trajectory the genuine classes produced;
#   * dropin_trace.json: which attributes of which plug-in object the reference's code read / wrote / called, and what
#     its serialise made of every object (key -> JSON kind): tests/test_dropin_cpu.py and tests/test_gpu_dropin.py hold
#     the real classes to it wherever they run.
# --------------------------------------------------------------------------------------------------------------
DROPIN_IMPORTS = (
    ("from pauxy.estimators.handler import Estimators", "from pauxy_amd.estimators.handler import Estimators"),
    ("from pauxy.propagation.utils import get_propagator_driver",
     "from pauxy_amd.propagation.continuous import get_propagator_driver"),
    ("from pauxy.walkers.handler import Walkers", "from pauxy_amd.walkers.handler import Walkers"),
)


def dropin_driver_module():
    """qmc/afqmc.py of the scratch copy with exactly the three import lines switched, as module pauxy.qmc.afqmc_dropin."""
    src = open(os.path.join(SCRATCH, "pauxy/qmc/afqmc.py")).read()
    for old, new in DROPIN_IMPORTS:
        assert src.count(old) == 1, old
        src = src.replace(old, new)
    with open(os.path.join(SCRATCH, "pauxy/qmc/afqmc_dropin.py"), "w") as f:
        f.write(src)
    import importlib
    return importlib.import_module("pauxy.qmc.afqmc_dropin")


class AccessTrace(object):
    """Records what code under SCRATCH/pauxy (the reference) does to the traced objects; accesses the plug-in classes
    make on themselves are not the driver's and are left out."""

    def __init__(self):
        self.log = {}

    def entry(self, label):
        return self.log.setdefault(label, {'read': set(), 'written': set(), 'called': set()})

    @staticmethod
    def from_reference(only=None, depth=2):
        """Is the code doing the access the reference's (``only``: that one file of it)?"""
        name = sys._getframe(depth).f_code.co_filename
        if only is not None:
            return name == os.path.join(SCRATCH, "pauxy", only)
        return name.startswith(os.path.join(SCRATCH, "pauxy"))

    def attach(self, obj, label, only=None):
        cls = type(obj)
        if getattr(cls, '_traced', False):
            return obj
        ent = self.entry(label)
        trace = self

        class Traced(cls):
            _traced = True

            def __getattribute__(self, name):
                try:
                    value = cls.__getattribute__(self, name)
                except AttributeError:
                    if not hasattr(cls, '__getattr__'):
                        raise
                    value = cls.__getattr__(self, name)
                if trace.from_reference(only):
                    if callable(value) and not isinstance(value, type) and hasattr(value, '__call__') \
                            and (hasattr(value, '__self__') or hasattr(value, '__func__')):
                        ent['called'].add(name)
                    else:
                        ent['read'].add(name)
                return value

            def __setattr__(self, name, value):
                if trace.from_reference(only):
                    ent['written'].add(name)
                cls.__setattr__(self, name, value)

        Traced.__name__ = cls.__name__
        Traced.__qualname__ = cls.__qualname__
        obj.__class__ = Traced
        return obj


def json_kinds(tree):
    """serialise output -> the same tree with every leaf replaced by the name of its JSON kind."""
    if isinstance(tree, dict):
        return dict((k, json_kinds(v)) for k, v in tree.items())
    if isinstance(tree, (list, tuple)):
        return 'array'
    if isinstance(tree, bool):
        return 'bool'
    if isinstance(tree, (int, float)):
        return 'number'
    if tree is None:
        return 'null'
    return 'string'


def make_dropin():
    import contextlib
    import io
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))            # /root/repo: pauxy_amd, tests, oracle
    os.environ.pop('AFQ_ESTIMATES_FILE', None)
    from tests import oracle_device
    oracle_device.install()
    mod = dropin_driver_module()
    import pauxy_amd.propagation.continuous as amd_prop
    import pauxy_amd.walkers.handler as amd_walkers
    import pauxy_amd.estimators.handler as amd_est
    assert mod.get_propagator_driver is amd_prop.get_propagator_driver
    assert mod.Walkers is amd_walkers.Walkers and mod.Estimators is amd_est.Estimators
    trace = AccessTrace()

    # the traced classes are attached where the driver receives the objects: wrap the three constructors it calls
    def traced_factory(fn, label, also=None):
        def build(*a, **k):
            obj = trace.attach(fn(*a, **k), label)
            if also:
                also(obj)
            return obj
        return build

    def trace_walkers(psi):
        for w in psi.walkers:
            trace.attach(w, 'Walker')

    def trace_estimators(est):
        for name, e in est.estimators.items():
            trace.attach(e, {'mixed': 'Mixed', 'back_prop': 'BackPropagation'}[name])

    def trace_prop(prop):
        if hasattr(prop, 'propagator'):                      # (the discrete Hirsch propagator has no inner object)
            trace.attach(prop.propagator, 'Continuous.propagator')

    mod.get_propagator_driver = traced_factory(amd_prop.get_propagator_driver, 'Propagator', trace_prop)
    mod.Walkers = traced_factory(amd_walkers.Walkers, 'Walkers', trace_walkers)
    mod.Estimators = traced_factory(amd_est.Estimators, 'Estimators', trace_estimators)
    # the genuine system / trial objects: only what the DRIVER reads of them (their own classes read far more), so that
    # pauxy_amd.systems / pauxy_amd.trial objects can be checked for the same surface
    driver_file = "qmc/afqmc_dropin.py"
    get_system, get_trial = mod.get_system, mod.get_trial_wavefunction
    mod.get_system = lambda *a, **k: trace.attach(get_system(*a, **k), 'system', only=driver_file)
    mod.get_trial_wavefunction = lambda *a, **k: trace.attach(get_trial(*a, **k), 'trial', only=driver_file)
    comm = MPI.COMM_WORLD
    doc = {'cases': {}}

    def run_case(name, golden, options, system=None, verbose=0):
        d = numpy.load(os.path.join(HERE, golden))
        with contextlib.redirect_stdout(io.StringIO()):
            afqmc = mod.AFQMC(comm=comm, system=system, options=options, verbose=verbose)
        # AFQMC.__init__ has run to_json(self) -> serialise (utils/misc.py:72-135) over the plug-in objects and stored
        # the string as the metadata of the estimator file
        meta = json.loads(afqmc.estimators.json_string)
        for key in ('propagators', 'estimators', 'psi'):
            assert key in meta, key
        store = h5py._STORE[afqmc.estimators.filename]
        assert json.loads(str(store['metadata'])) == meta
        rec = dict(weight=[], unscaled_weight=[], ot=[], ehyb=[], phase=[], eloc=[], pix=[])
        est_update = afqmc.estimators.update

        def update(system_, qmc, trial, psi_, step, fp):
            rec['weight'].append([w.weight for w in psi_.walkers])
            rec['unscaled_weight'].append([w.unscaled_weight for w in psi_.walkers])
            rec['ot'].append([w.ot for w in psi_.walkers])
            rec['ehyb'].append([w.hybrid_energy for w in psi_.walkers])
            rec['phase'].append([w.phase for w in psi_.walkers])
            rec['eloc'].append([w.eloc for w in psi_.walkers])
            if step % qmc.npop_control == 0 and psi_.ntot_walkers > 1:
                rec['pix'].append(numpy.array(psi_.last_parent_ix))
            return est_update(system_, qmc, trial, psi_, step, fp)

        object.__setattr__(afqmc.estimators, 'update', update)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                afqmc.run(comm=comm, verbose=verbose)
                afqmc.finalise(verbose=True)
        finally:
            object.__delattr__(afqmc.estimators, 'update')
        afqmc.estimators.flush()

        def close(a, b, what, tol=1e-8):
            a, b = numpy.asarray(a), numpy.asarray(b)
            assert a.shape == b.shape, (what, a.shape, b.shape)
            err = float(numpy.max(numpy.abs(a - b))) / max(1.0, float(numpy.max(numpy.abs(b))))
            assert err <= tol, (what, err)
            return err

        errs = {}
        errs['weight'] = close(rec['weight'], d['weight'], 'weight')
        errs['unscaled_weight'] = close(rec['unscaled_weight'], d['unscaled_weight'], 'unscaled_weight')
        errs['ot'] = close(rec['ot'], d['ot'], 'ot')
        for key in ('ehyb', 'phase', 'eloc'):                # (the discrete-field fixtures record fewer walker scalars)
            if key in d.files:
                errs[key] = close(rec[key], d[key], key)
        assert numpy.array_equal(numpy.array(rec['pix']).reshape(d['parent_ix'].shape), d['parent_ix']), 'parent_ix'
        keys = sorted(k for k in store if k.startswith('basic/energies/'))
        blocks = numpy.array([store[k] for k in keys])
        errs['blocks'] = close(blocks[:, 1:10], d['blocks'][:, 1:10], 'blocks')
        errs['final_phi'] = close(numpy.array([w.phi for w in afqmc.psi.walkers]), d['final_phi'], 'final_phi')
        if 'nfb_trig' in d.files:
            assert afqmc.propagators.nfb_trig == int(d['nfb_trig']) and afqmc.propagators.nhe_trig == int(d['nhe_trig'])
        dev = oracle_device.OracleDevice.instances[-1]
        nprop = dev.calls.count('propagate')
        assert nprop == afqmc.qmc.total_steps, (nprop, afqmc.qmc.total_steps)      # ONE batched launch per step
        doc['cases'][name] = {'golden': golden, 'steps': int(afqmc.qmc.total_steps),
                              'max_rel_err': dict((k, float(v)) for k, v in errs.items()),
                              'serialised': dict((k, json_kinds(meta[k])) for k in ('propagators', 'estimators', 'psi'))}
        print('dropin %-12s %s: genuine driver over the plug-in classes == %s  (max rel err %.1e)'
              % (name, 'verbose' if verbose else 'quiet', golden, max(errs.values())))
        import pauxy_amd.context as amd_ctx
        amd_ctx.release_context(afqmc.system, afqmc.trial)
        return afqmc

    # BASELINE configs[0]: 4x4 U=4 half filling, 10 walkers, comb every 5 steps (the options of make_traj_hubbard)
    run_case('hubbard_c1', 'traj_hubbard_c1.npz',
             {'verbosity': 0, 'get_sha1': False,
              'qmc': {'timestep': 0.01, 'num_steps': 10, 'blocks': 10, 'rng_seed': 8, 'num_walkers': 10,
                      'pop_control_freq': 5},
              'model': {'name': "Hubbard", 'nx': 4, 'ny': 4, 'nup': 8, "U": 4, 'ndown': 8},
              'trial': {'name': 'UHF'}, 'estimates': {'mixed': {'energy_eval_freq': 1}},
              'propagator': {'hubbard_stratonovich': 'continuous'}})
    # qmc/tests/test_afqmc.py:190-229: generic Cholesky Hamiltonian, RHF-type MultiSlater trial (the options of make_traj_generic)
    nmo, nelec = 11, (3, 3)
    numpy.random.seed(7)
    h1e, chol, enuc, eri = generate_hamiltonian(nmo, nelec, cplx=False)
    system = Generic(nelec=nelec, h1e=numpy.array([h1e, h1e]), chol=chol.reshape((-1, nmo * nmo)).T.copy(), ecore=enuc)
    run_case('generic', 'traj_generic.npz',
             {'verbosity': 0, 'get_sha1': False, 'qmc': {'timestep': 0.005, 'steps': 10, 'blocks': 10, 'rng_seed': 8},
              'estimates': {'mixed': {'energy_eval_freq': 1}}, 'trial': {'name': 'MultiSlater'}}, system=system)
    # qmc/tests/test_afqmc.py:49-97: the electron gas (plane-wave propagator, sparse density operators, HartreeFock trial)
    run_case('ueg', 'traj_ueg.npz',
             {'verbosity': 0, 'get_sha1': False, 'qmc': {'timestep': 0.01, 'num_steps': 10, 'blocks': 5, 'rng_seed': 8},
              'model': {'name': "UEG", 'rs': 2.44, 'ecut': 2, 'nup': 7, 'ndown': 7},
              'estimates': {'mixed': {'energy_eval_freq': 1}}, 'trial': {'name': 'hartree_fock'}})
    # qmc/tests/test_afqmc.py:232-278: the back-propagated one-body RDM (estimators/back_propagation.py through the driver:
    # Estimators builds BackPropagation, Walkers gets nbp, every step's fields go into the walkers' history)
    numpy.random.seed(7)
    h1e, chol, enuc, eri = generate_hamiltonian(nmo, nelec, cplx=False)
    system = Generic(nelec=nelec, h1e=numpy.array([h1e, h1e]), chol=chol.reshape((-1, nmo * nmo)).T.copy(), ecore=enuc)
    bp_run = run_case('generic_bp', 'traj_bp.npz',
                      {'verbosity': 0, 'get_sha1': False, 'qmc': {'timestep': 0.005, 'num_steps': 10, 'blocks': 10, 'rng_seed': 8},
                       'trial': {'name': 'MultiSlater'},
                       'estimator': {'back_propagated': {'tau_bp': 0.025, 'one_rdm': True}, 'mixed': {'energy_eval_freq': 1}}},
                      system=system)
    d = numpy.load(os.path.join(HERE, 'traj_bp.npz'))
    store = h5py._STORE[bp_run.estimators.filename]
    nbp = bp_run.estimators.nbp
    dk = sorted(k for k in store if k.startswith('back_propagated/denominator_%d/' % nbp))
    rk = sorted(k for k in store if k.startswith('back_propagated/one_rdm_%d/' % nbp))
    got_den = numpy.array([store[k] for k in dk]).reshape(len(dk))
    got_rdm = numpy.array([store[k] for k in rk])
    assert got_den.shape == d['bp_denominator'].shape and got_rdm.shape == d['bp_one_rdm'].shape
    assert numpy.max(numpy.abs(got_den - d['bp_denominator'])) <= 1e-9 * numpy.max(numpy.abs(d['bp_denominator']))
    assert numpy.max(numpy.abs(got_rdm - d['bp_one_rdm'])) <= 1e-9 * numpy.max(numpy.abs(d['bp_one_rdm']))
    rdm = got_rdm / got_den[:, None, None, None]
    assert abs(rdm[11, 0, 1, 3].real - (-0.121883381144845)) < 1e-9           # the value qmc/tests/test_afqmc.py pins
    print('dropin generic_bp   back-propagated one-body RDM of the genuine driver over the plug-in classes == traj_bp.npz')
    # the discrete Hirsch fields (propagation/hubbard.py:12-343 behind get_propagator_driver's 'discrete' branch): the options
    # of make_traj_hirsch; numpy.random.random is drawn M times per surviving walker by the plug-in class, in walker order
    run_case('hubbard_hirsch', 'traj_hubbard_hirsch.npz',
             {'verbosity': 0, 'get_sha1': False,
              'qmc': {'timestep': 0.01, 'num_steps': 10, 'blocks': 10, 'rng_seed': 8},
              'model': {'name': "Hubbard", 'nx': 4, 'ny': 4, 'nup': 7, "U": 4, 'ndown': 7},
              'trial': {'name': 'UHF'}, 'estimates': {'mixed': {'energy_eval_freq': 1}},
              'propagator': {'hubbard_stratonovich': 'discrete'}})
    # the local-energy weight update and free projection take other branches of the driver-facing classes
    for name, golden, extra in (('hubbard_le', 'traj_hubbard_le.npz', {'hybrid': False}),
                                ('hubbard_fp', 'traj_hubbard_fp.npz', {'free_projection': True})):
        run_case(name, golden,
                 {'verbosity': 0, 'get_sha1': False,
                  'qmc': {'timestep': 0.01, 'num_steps': 10, 'blocks': 4, 'rng_seed': 8, 'num_walkers': 10,
                          'pop_control_freq': 5},
                  'model': {'name': "Hubbard", 'nx': 4, 'ny': 4, 'nup': 8, "U": 4, 'ndown': 8},
                  'trial': {'name': 'UHF'}, 'estimates': {'mixed': {'energy_eval_freq': 1}},
                  'propagator': dict({'hubbard_stratonovich': 'continuous'}, **extra)})
    # once more with the driver's verbose branches (print_key, print_header, the step-0 print_step): same numbers apart
    # from the step-0 row, which a verbose run prints and zeroes instead of folding it into the first block
    # (utils/misc.py:265 get_sys_info reads numpy.__config__.blas_opt_info, which numpy 2 no longer has: the one call of
    #  the verbose constructor that cannot run in this image is replaced; it does not touch the plug-in objects)
    mod.get_sys_info = lambda sha1, branch, uuid, nranks: {}
    with contextlib.redirect_stdout(io.StringIO()) as text:
        mod.AFQMC(comm=comm, verbose=1, options={
            'verbosity': 1, 'get_sha1': False,
            'qmc': {'timestep': 0.01, 'num_steps': 5, 'blocks': 2, 'rng_seed': 8, 'num_walkers': 4, 'pop_control_freq': 5},
            'model': {'name': "Hubbard", 'nx': 4, 'ny': 4, 'nup': 8, "U": 4, 'ndown': 8}, 'trial': {'name': 'UHF'},
            'estimates': {'mixed': {'energy_eval_freq': 1}},
            'propagator': {'hubbard_stratonovich': 'continuous'}}).run(comm=comm, verbose=True)
    assert '# Explanation of output column headers:' in text.getvalue() and 'WeightFactor' in text.getvalue()
    doc['trace'] = dict((label, dict((k, sorted(v)) for k, v in ent.items())) for label, ent in sorted(trace.log.items()))
    doc['driver_imports_switched'] = [new for _, new in DROPIN_IMPORTS]
    with open(os.path.join(OUT, 'dropin_trace.json'), 'w') as f:
        json.dump(doc, f, indent=1, sort_keys=True)
        f.write('\n')
    for label, ent in sorted(trace.log.items()):
        print('%-22s read %d, wrote %d, called %d' % (label, len(ent['read']), len(ent['written']), len(ent['called'])))


# Every committed fixture and the ONE call that makes it.  `python make_golden.py` (or `all`) regenerates all of them,
# `python make_golden.py <file or group> ...` some, `--check` regenerates into a scratch directory and compares with the
# committed files array by array (exit status 1 on any difference).
# comb uniforms that make the C3-size trajectories clone walkers (weights differ by 1e-3 there: see record_trajectory)
C3_R = {15: 0.9999, 20: 0.0001, 25: 0.9999, 30: 0.0001}

FIXTURES = [
    ('ops', 'generic_ops.npz', lambda: make_generic_ops()),
    ('ops', 'hubbard_ops.npz', lambda: make_hubbard_ops()),
    ('ops', 'ueg_ops.npz', lambda: make_ueg_ops()),
    ('traj', 'traj_generic.npz', lambda: make_traj_generic()),
    # BASELINE configs[2] at its own size and cadence (closed-shell walkers; and a third of them opened)
    ('traj', 'traj_generic_c3.npz', lambda: make_traj_generic_c3(r_override=C3_R)),
    ('traj', 'traj_generic_c3_open.npz', lambda: make_traj_generic_c3('traj_generic_c3_open.npz', nopen=12, r_override=C3_R)),
    ('traj', 'traj_hubbard.npz', lambda: make_traj_hubbard('traj_hubbard.npz', 7, pin=-152.91937839611)),
    # BASELINE configs[0]: 4x4 U=4 half filling, 10 walkers, comb every 5 steps
    ('traj', 'traj_hubbard_c1.npz', lambda: make_traj_hubbard('traj_hubbard_c1.npz', 8, nwalkers=10, npop=5, blocks=10)),
    ('traj', 'traj_ueg.npz', lambda: make_traj_ueg()),
    # free projection (propagation/continuous.py:175-200, estimators/mixed.py:151-175) and the local-energy weight
    # update (continuous.py:294-318), same 4x4 U=4 model
    ('traj', 'traj_hubbard_fp.npz', lambda: make_traj_hubbard('traj_hubbard_fp.npz', 8, nwalkers=10, npop=5, blocks=4,
                                                              prop_extra={'free_projection': True})),
    ('traj', 'traj_hubbard_le.npz', lambda: make_traj_hubbard('traj_hubbard_le.npz', 8, nwalkers=10, npop=5, blocks=4,
                                                              prop_extra={'hybrid': False})),
    ('msd', 'msd_ops.npz', lambda: make_msd_ops()),
    ('msd', 'traj_msd.npz', lambda: make_traj_msd()),
    ('bp', 'traj_bp.npz', lambda: make_traj_bp()),
    ('bp', 'traj_bp_full.npz', lambda: make_traj_bp('traj_bp_full.npz', restore_weights='full', blocks=4)),
    ('bp', 'traj_bp_split.npz', lambda: make_traj_bp('traj_bp_split.npz', blocks=4, nsplit=2, tau_bp=0.03)),
    ('bp', 'traj_bp_ueg.npz', lambda: make_traj_bp_ueg()),
    ('traj', 'traj_hubbard_rdm.npz', lambda: make_traj_mixed_rdm()),
    ('traj', 'traj_hubbard_logshift.npz', lambda: make_traj_log_shift()),
    ('hirsch', 'traj_hubbard_hirsch.npz', lambda: make_traj_hirsch(pin=-152.68468568462666)),
    ('hirsch', 'traj_hubbard_hirsch_charge.npz', lambda: make_traj_hirsch('traj_hubbard_hirsch_charge.npz', charge=True,
                                                                          blocks=4)),
    # discrete fields + use_log_shift: calc_otrial shifts the determinant of the inverse overlap (single_det.py:159)
    ('hirsch', 'traj_hirsch_logshift.npz', lambda: make_traj_hirsch('traj_hirsch_logshift.npz', blocks=4,
                                                                    walkers={'use_log_shift': True})),
    # propagate_walker_free (propagation/hubbard.py:303-343) through the reference driver, spin and charge decomposition
    ('hirsch', 'traj_hirsch_fp.npz', lambda: make_traj_hirsch('traj_hirsch_fp.npz', blocks=3,
                                                              prop_extra={'free_projection': True})),
    ('hirsch', 'traj_hirsch_fp_charge.npz', lambda: make_traj_hirsch('traj_hirsch_fp_charge.npz', charge=True, blocks=2,
                                                                     prop_extra={'free_projection': True})),
    # propagation/hubbard.py:222-275 (two_body_direct: dynamic force bias, all sites at once), both decompositions
    ('hirsch', 'traj_hirsch_direct.npz', lambda: make_traj_hirsch('traj_hirsch_direct.npz', blocks=3,
                                                                  prop_extra={'single_site_update': False})),
    ('hirsch', 'traj_hirsch_direct_charge.npz', lambda: make_traj_hirsch('traj_hirsch_direct_charge.npz', charge=True,
                                                                         blocks=2,
                                                                         prop_extra={'single_site_update': False})),
    ('hirsch', 'traj_hirsch_bp.npz', lambda: make_traj_hirsch('traj_hirsch_bp.npz', blocks=4,
                                                              bp={'tau_bp': 0.04, 'one_rdm': True})),
    ('io', 'io_formats.npz', lambda: make_io()),
    # (back-propagated energies: the reference raises TypeError at back_propagation.py:160 -- local_energy() has no
    #  'opt' keyword -- so there is no reference output to record for evaluate_energy)
    # the genuine driver over the pauxy_amd plug-in classes; reads the trajectories above, writes dropin_trace.json
    ('dropin', 'dropin_trace.json', lambda: make_dropin()),
]


def compare_fixture(name, fresh_dir):
    """[] when the fresh copy of fixture ``name`` equals the committed one, else what differs."""
    a, b = os.path.join(HERE, name), os.path.join(fresh_dir, name)
    if not os.path.exists(a):
        return ['not committed']
    if name.endswith('.json'):
        import json
        x, y = json.load(open(a)), json.load(open(b))
        for case in list(x.get('cases', {}).values()) + list(y.get('cases', {}).values()):
            case.pop('max_rel_err', None)                      # rounding-level numbers of the run, not its content
        return [] if x == y else ['json content differs']
    x, y = numpy.load(a, allow_pickle=False), numpy.load(b, allow_pickle=False)
    bad = ['only committed: ' + k for k in sorted(set(x.files) - set(y.files))]
    bad += ['only fresh: ' + k for k in sorted(set(y.files) - set(x.files))]
    for k in sorted(set(x.files) & set(y.files)):
        u, v = x[k], y[k]
        if k in ('blocks', 'final_estimates') and u.shape == v.shape:     # fixtures of earlier rounds kept the clock
            u, v = numpy.array(u), numpy.array(v)
            if k == 'blocks':
                u[:, -1] = v[:, -1] = 0
            else:
                u[9] = v[9] = 0
        if u.shape != v.shape or u.dtype != v.dtype or not numpy.array_equal(u, v, equal_nan=u.dtype.kind in 'fc'):
            bad.append('differs: ' + k)
    return bad


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if a != '--check']
    check = '--check' in sys.argv[1:]
    known = set(g for g, _, _ in FIXTURES) | set(n for _, n, _ in FIXTURES) | {'all'}
    for a in args:
        if a not in known:
            sys.exit("unknown fixture or group %r (groups: %s)" % (a, ' '.join(sorted(set(g for g, _, _ in FIXTURES)))))
    chosen = [f for f in FIXTURES if not args or 'all' in args or f[0] in args or f[1] in args]
    if check:
        import tempfile
        OUT = tempfile.mkdtemp(prefix='golden_check_')
    failed = 0
    for group, name, make in chosen:
        make()
        if check:
            bad = compare_fixture(name, OUT)
            print('%-34s %s' % (name, 'identical to the committed fixture' if not bad else 'DIFFERS: ' + '; '.join(bad[:6])))
            failed += bool(bad)
        else:
            print('%-34s %d bytes' % (name, os.path.getsize(os.path.join(HERE, name))))
    if check:
        shutil.rmtree(OUT, ignore_errors=True)
        print('%d of %d fixtures differ' % (failed, len(chosen)) if failed else 'all %d fixtures reproduce' % len(chosen))
        sys.exit(1 if failed else 0)
