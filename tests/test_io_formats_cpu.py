"""On-disk formats either side of the device path, CPU only.

* ``h5lite`` reads the reference's own HDF5 fixture (pauxy/trial_wavefunction/tests/wfn.h5, written by
  h5py/libhdf5; committed here as tests/golden/ref_wfn_nomsd.h5) and the result is structurally consistent;
* the QMCPACK-format writers put exactly the datasets (names, shapes, dtypes, values) the reference's
  writers put, and the readers return what the reference's readers return
  (tests/golden/io_formats.npz, recorded from the genuine reference by make_golden.py io);
* container round trips: nesting, >1000 members per group (multi-level B-tree), dtypes, append / in-place modes.
"""
import json
import os

import numpy
import pytest

from pauxy_amd.utils import h5lite
from pauxy_amd.utils import io as aio
from pauxy_amd.estimators.utils import H5EstimatorHelper

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = numpy.load(os.path.join(HERE, "golden", "io_formats.npz"), allow_pickle=False)
REF_WFN = os.path.join(HERE, "golden", "ref_wfn_nomsd.h5")
REF_EST = os.path.join(HERE, "golden", "ref_analysed_estimates.h5")     # docs/source/tutorials/calcs/hubbard/


def flatten(group, prefix=''):
    out = {}
    for k, v in group.items():
        if isinstance(v, h5lite.Group):
            out.update(flatten(v, prefix + k + '/'))
        else:
            out[prefix + k] = v[()] if v.shape == () else v[:]
    return out


def golden_file(fname):
    pre = fname + '|'
    return {k[len(pre):]: GOLD[k] for k in GOLD.files if k.startswith(pre) and not k.startswith(pre + 'read|')}


def assert_same_file(path, fname):
    with h5lite.File(path, 'r') as f:
        mine = flatten(f)
    want = golden_file(fname)
    assert sorted(mine) == sorted(want)
    for k in want:
        a, b = numpy.asarray(mine[k]), want[k]
        assert a.shape == b.shape, k
        assert a.dtype == b.dtype, (k, a.dtype, b.dtype)
        assert numpy.array_equal(a, b), k


# ------------------------------------------------------------------ real file
def test_reads_reference_hdf5_fixture():
    with h5lite.File(REF_WFN, 'r') as f:
        assert f.keys() == ['Wavefunction'] and f['Wavefunction'].keys() == ['NOMSD']
        g = f['Wavefunction/NOMSD']
        dims = g['dims'][:]
        assert dims.dtype == numpy.int32 and list(dims) == [93, 7, 7, 1, 5]      # UEG rs=5 ecut=4, 7+7, 5 dets
        assert g['ci_coeffs'].shape == (5, 2) and g['Psi0_alpha'].shape == (93, 7, 2)
        assert 'Psi0_beta' not in g
        for d in range(5):
            p = g['PsiT_%d' % d]
            nrow, ncol, nnz = p['dims'][:]
            assert (nrow, ncol) == (7, 93)
            assert p['data_'].shape == (nnz, 2) and p['data_'].dtype == numpy.float64
            j, b, e = p['jdata_'][:], p['pointers_begin_'][:], p['pointers_end_'][:]
            assert j.shape == (nnz,) and j.min() >= 0 and j.max() < 93
            assert b[0] == 0 and e[-1] == nnz and numpy.array_equal(b[1:], e[:-1]) and (e >= b).all()
        with pytest.raises(KeyError):
            f['Wavefunction/PHMSD']
    (coeffs, wfn), psi0 = aio.read_qmcpack_wfn_hdf(REF_WFN)
    assert coeffs.shape == (5,) and wfn.shape == (5, 93, 14) and psi0.shape == (93, 14)
    assert numpy.array_equal(wfn[:, :, :7], wfn[:, :, 7:])                       # closed-shell walker type
    assert numpy.array_equal(psi0[:, :7], psi0[:, 7:])
    assert numpy.isfinite(wfn.view(numpy.float64)).all() and numpy.abs(wfn).max() <= 1.0 + 1e-12
    # every determinant has 7 linearly independent orbitals
    for d in range(5):
        assert numpy.linalg.matrix_rank(wfn[d, :, :7]) == 7
    with pytest.raises(ValueError):
        aio.read_qmcpack_wfn_hdf(REF_WFN, nelec=(7, 6))


def test_reads_reference_estimates_file():
    """A second libhdf5-written file of the reference (its Hubbard tutorial's analysed estimates, written through
    h5py/pandas in 2018): variable-length strings in a global heap collection (the ``metadata`` JSON and the header
    arrays), 5-dimensional float datasets, modification-time and padding messages."""
    with h5lite.File(REF_EST, 'r') as f:
        assert f.keys() == ['back_propagated', 'metadata', 'mixed', 'real_itcf', 'real_itcf_err']
        meta = json.loads(f['metadata'][:][0])
        assert meta['system']['name'] == 'Hubbard' and meta['system']['nup'] == 3 and meta['qmc']['dt'] == 0.05
        assert meta['estimators']['estimators']['back_prop']['nmax'] == 40
        assert [x.decode() for x in f['mixed/headers'][:]][:4] == ['ndets', 'dt', 'E', 'E_error']
        est = f['mixed/estimates'][:]
        assert est.shape == (1, 8) and est[0, 1] == 0.05 and est[0, 2] == pytest.approx(-9.66736746, abs=1e-8)
        bp = f['back_propagated/estimates'][:]
        assert bp.shape == (1, 10) and bp[0, 0] == 40.0 and bp[0, 2] == pytest.approx(-10.17259488, abs=1e-8)
        itcf = f['real_itcf']
        assert itcf.shape == (41, 2, 2, 9, 9) and itcf.dtype == numpy.float64
        g = itcf[:]
        # equal-time Green's function of the 3x3 lattice: G(0)_ii = 1 - n_i, translation invariant
        assert numpy.allclose(numpy.diag(g[0, 0, 0]), g[0, 0, 0, 0, 0], atol=0.1) and 0.0 < g[0, 0, 0, 0, 0] < 1.0
        assert numpy.isfinite(g).all() and (f['real_itcf_err'][:] >= 0).all()


def test_rewrites_reference_fixture_identically(tmp_path):
    """read (libhdf5-written) -> write (h5lite) -> read: same tree, same bytes per dataset."""
    with h5lite.File(REF_WFN, 'r') as f:
        ref = flatten(f)
    out = str(tmp_path / 'copy.h5')
    with h5lite.File(out, 'w') as f:
        for k, v in ref.items():
            f[k] = v
    with h5lite.File(out, 'r') as f:
        got = flatten(f)
    assert sorted(got) == sorted(ref)
    for k in ref:
        assert got[k].dtype == ref[k].dtype and numpy.array_equal(got[k], ref[k])
    a = aio.read_qmcpack_wfn_hdf(out)
    b = aio.read_qmcpack_wfn_hdf(REF_WFN)
    assert numpy.array_equal(a[0][1], b[0][1]) and numpy.array_equal(a[0][0], b[0][0]) and numpy.array_equal(a[1], b[1])


# ------------------------------------------------------- Hamiltonian formats
@pytest.mark.parametrize("fname", ['dense_real.h5', 'dense_cplx.h5', 'sparse_real.h5', 'sparse_cplx.h5'])
def test_hamiltonian_files_match_reference(tmp_path, fname):
    nelec = tuple(int(x) for x in GOLD['nelec'])
    path = str(tmp_path / fname)
    real = 'real' in fname
    h, chol = (GOLD['h'], GOLD['chol']) if real else (GOLD['hc'], GOLD['cholc'])
    if fname.startswith('dense'):
        aio.write_qmcpack_dense(h, chol, nelec, 5, enuc=float(GOLD['enuc']), filename=path, real_chol=real,
                                ortho=None if real else GOLD['X'])
        r = aio.from_qmcpack_dense(path)
        with pytest.raises(KeyError):
            aio.from_qmcpack_sparse(path)
        got_chol = r[1]
    else:
        aio.write_qmcpack_sparse(h, chol, nelec, 5, enuc=float(GOLD['enuc']), filename=path, real_chol=real)
        r = aio.from_qmcpack_sparse(path)
        with pytest.raises(KeyError):
            aio.from_qmcpack_dense(path)
        got_chol = r[1].toarray()
    assert_same_file(path, fname)
    assert numpy.array_equal(r[0], GOLD[fname + '|read|hcore'])
    assert got_chol.dtype == GOLD[fname + '|read|chol'].dtype
    assert numpy.array_equal(got_chol, GOLD[fname + '|read|chol'])
    assert [r[2], r[3], r[4], r[5]] == list(GOLD[fname + '|read|scalars'])
    # dispatching readers
    hcore, dense, ecore = aio.read_integrals(path)
    assert numpy.array_equal(dense, got_chol) and ecore == 1.25
    ham = aio.read_qmcpack_hamiltonian(path)
    assert ham['nmo'] == 5 and ham['nelec'] == nelec and numpy.array_equal(ham['chol'], got_chol)


def test_generic_system_from_file(tmp_path):
    from pauxy_amd.systems import Generic, get_system
    path = str(tmp_path / 'ham.h5')
    aio.write_qmcpack_dense(GOLD['h'], GOLD['chol'], (2, 2), 5, enuc=0.5, filename=path)
    s = get_system({'name': 'Generic', 'integrals': path, 'nup': 2, 'ndown': 2})
    ref = Generic((2, 2), numpy.array([GOLD['h'], GOLD['h']]), GOLD['chol'], 0.5)
    assert s.nbasis == 5 and s.nchol == 7 and s.ecore == 0.5
    assert numpy.array_equal(s.chol_vecs, ref.chol_vecs) and numpy.allclose(s.h1e_mod, ref.h1e_mod, rtol=0, atol=1e-15)
    with pytest.raises(ValueError):
        get_system({'name': 'Generic', 'nup': 2, 'ndown': 2})
    with pytest.raises(OSError):
        aio.read_integrals(str(tmp_path / 'missing.h5'))


# ------------------------------------------------------ wavefunction formats
@pytest.mark.parametrize("fname", ['nomsd_uhf.h5', 'nomsd_rhf.h5', 'nomsd_init.h5', 'phmsd.h5', 'phmsd_init.h5'])
def test_wavefunction_files_match_reference(tmp_path, fname):
    nelec = tuple(int(x) for x in GOLD['nelec'])
    path = str(tmp_path / fname)
    init = [GOLD['init_a'], GOLD['init_b']] if 'init' in fname else None
    if fname.startswith('nomsd'):
        wfn = (GOLD['coeffs'].copy(), GOLD['dets'].copy())
    else:
        wfn = (GOLD['coeffs'].copy(), GOLD['occa'], GOLD['occb'])
    aio.write_qmcpack_wfn(path, wfn, 'rhf' if 'rhf' in fname else 'uhf', nelec, 5, init=init)
    assert_same_file(path, fname)
    got, psi0 = aio.read_qmcpack_wfn_hdf(path)
    n = len(got)
    assert n == sum(1 for k in GOLD.files if k.startswith(fname + '|read|wfn'))
    for i in range(n):
        want = GOLD[fname + '|read|wfn%d' % i]
        if fname in ('nomsd_uhf.h5', 'nomsd_init.h5') and i == 1:
            # the reference's UHF reader takes the alpha block of determinant d from PsiT_<d> instead of
            # PsiT_<2d> (io.py:364-366), i.e. it returns another determinant's beta orbitals for d >= 1.
            # The writer (and QMCPACK) put alpha at 2d, beta at 2d+1; this reader follows the writer.
            thr = numpy.abs(GOLD['dets']) >= 1e-8
            assert numpy.array_equal(got[i][0], want[0])
            assert numpy.array_equal(got[i][:, :, 2:], want[:, :, 2:])
            assert not numpy.array_equal(got[i][1:, :, :2], want[1:, :, :2])
            want = GOLD['dets'] * thr
        assert numpy.array_equal(got[i], want)
    assert numpy.array_equal(psi0, GOLD[fname + '|read|psi0'])
    # writing again in append mode replaces the group (io.py:432-446)
    aio.write_qmcpack_wfn(path, wfn, 'rhf' if 'rhf' in fname else 'uhf', nelec, 5, init=init, mode='a')
    assert_same_file(path, fname)


def test_trial_from_wavefunction_file(tmp_path):
    from pauxy_amd.systems import Generic
    from pauxy_amd.trial import get_trial_wavefunction, MultiDetTrial, SingleDetTrial
    path = str(tmp_path / 'wfn.h5')
    aio.write_qmcpack_wfn(path, (GOLD['coeffs'].copy(), GOLD['dets'].copy()), 'uhf', (2, 2), 5)
    system = Generic((2, 2), numpy.array([GOLD['h'], GOLD['h']]), GOLD['chol'], 0.5)
    t = get_trial_wavefunction(system, {'filename': path})
    assert isinstance(t, MultiDetTrial) and t.ndets == 3
    thr = numpy.abs(GOLD['dets']) >= 1e-8
    assert numpy.array_equal(t.psi, GOLD['dets'] * thr) and numpy.array_equal(t.coeffs, GOLD['coeffs'])
    assert numpy.array_equal(t.init, GOLD['nomsd_uhf.h5|read|psi0'])
    t1 = get_trial_wavefunction(system, {'wavefunction_file': path, 'ndets': 1})
    assert isinstance(t1, SingleDetTrial) and numpy.array_equal(t1.psi, t.psi[0])
    t0 = get_trial_wavefunction(system, {})
    assert numpy.array_equal(t0.psi[:, :2], numpy.eye(5)[:, :2])


# ------------------------------------------------------------ the container
def test_container_round_trip(tmp_path):
    path = str(tmp_path / 't.h5')
    rng = numpy.random.RandomState(3)
    arrays = {
        'f8': rng.rand(3, 4), 'f4': rng.rand(5).astype(numpy.float32), 'i8': numpy.arange(-3, 4),
        'i4': numpy.arange(6, dtype=numpy.int32).reshape(2, 3), 'u1': numpy.arange(5, dtype=numpy.uint8),
        'c16': rng.rand(2, 2) + 1j * rng.rand(2, 2), 'c8': (rng.rand(3) + 1j * rng.rand(3)).astype(numpy.complex64),
        'S': numpy.array(['Iteration', 'Weight', 'E']).astype('S'), 'empty': numpy.zeros((0, 3)),
        'scalar': numpy.float64(2.5), 'be': numpy.arange(4, dtype='>f8'),
        'deep/er/and/deeper': numpy.ones(2),
    }
    with h5lite.File(path, 'w') as f:
        for k, v in arrays.items():
            f[k] = v
        f['text'] = json.dumps({'a': [1, 2]})
        f['pyint'] = 7
        g = f.create_group('grp')
        g['x'] = numpy.arange(3)
        with pytest.raises(ValueError):
            f.create_group('grp')
        with pytest.raises(ValueError):
            f['f8'] = 1.0
        for i in range(1100):
            f['many/%09d' % i] = numpy.array([i, 0.5 * i])
        assert len(f['many']) == 1100
    with h5lite.File(path, 'r') as f:
        for k, v in arrays.items():
            got = f[k][()] if numpy.ndim(v) == 0 else f[k][:]
            assert numpy.asarray(got).dtype == numpy.asarray(v).dtype and numpy.array_equal(got, v), k
            assert f[k].shape == numpy.shape(v)
        assert json.loads(f['text'][()]) == {'a': [1, 2]} and f['pyint'][()] == 7
        assert f['grp/x'][1] == 1 and f['/grp']['x'][-1] == 2
        assert f['many'].keys() == ['%09d' % i for i in range(1100)]
        assert all(f['many/%09d' % i][0] == i for i in range(0, 1100, 37))
        assert 'nope' not in f and 'grp/x' in f and 'grp/y' not in f
        with pytest.raises(KeyError):
            f['many/x']
        with pytest.raises(OSError):
            f['new'] = 1
    # in-place update and append, then everything else is still there
    with h5lite.File(path, 'r+') as f:
        f['c16'][0, :] = [1j, 2j]
        f['f8'][...] = 0.0
    with h5lite.File(path, 'a') as f:
        f['many/%09d' % 1100] = numpy.array([1100, 0.0])
        del f['grp']
        del f['deep/er']
    with h5lite.File(path, 'r') as f:
        assert list(f['c16'][0]) == [1j, 2j] and numpy.array_equal(f['c16'][1], arrays['c16'][1])
        assert not f['f8'][:].any() and len(f['many']) == 1101 and 'grp' not in f
        assert f['deep'].keys() == [] and numpy.array_equal(f['i4'][:], arrays['i4'])
        assert json.loads(f['text'][()]) == {'a': [1, 2]}
    with pytest.raises(OSError):
        h5lite.File(str(tmp_path / 'missing.h5'), 'r')
    with open(str(tmp_path / 'junk.h5'), 'wb') as fh:
        fh.write(b'not an hdf5 file' * 10)
    with pytest.raises(IOError):
        h5lite.File(str(tmp_path / 'junk.h5'), 'r')


def test_estimator_file_layout(tmp_path):
    """estimators/utils.py:279-327: <base>/<name>/<nine-digit block>; queued pushes reach the file on flush."""
    path = str(tmp_path / 'estimates.0.h5')
    with aio.h5.File(path, 'w'):
        pass
    with aio.h5.File(path, 'a') as f:
        f['basic/headers'] = numpy.array(['Iteration', 'Weight']).astype('S')
        f['metadata'] = json.dumps({'qmc': {'dt': 0.01}})
    out = H5EstimatorHelper(path, 'basic', flush_every=4)
    rows = [numpy.array([i, i + 0.25j]) for i in range(10)]
    for r in rows:
        out.push(r, 'energies')
        out.increment()
    with aio.h5.File(path, 'r') as f:
        assert len(f['basic/energies']) == 8                  # two batches of four written so far
    out.flush()
    with aio.h5.File(path, 'r') as f:
        assert f['basic/energies'].keys() == ['%09d' % i for i in range(10)]
        for i, r in enumerate(rows):
            assert numpy.array_equal(f['basic/energies/%09d' % i][:], r)
        assert list(f['basic/headers'][:]) == [b'Iteration', b'Weight']
        assert json.loads(f['metadata'][()])['qmc']['dt'] == 0.01
