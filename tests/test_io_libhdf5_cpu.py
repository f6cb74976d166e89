"""The genuine HDF5 library against pauxy_amd/utils/h5lite.py, both directions (tests/hdf5_ctypes.py drives the
libhdf5 1.10 the image carries under /opt/conda/lib; there is no h5py):

* files the writers of the path produce through h5lite (Hamiltonian dense / sparse, wavefunction, estimator output
  with appended blocks, restart walkers) are opened by libhdf5 and every dataset it reads equals what was written;
* contiguous, chunked, chunked + deflate and chunked + shuffle + deflate datasets written by libhdf5 are read by
  h5lite (the layouts a user's h5py may have produced for the reference's input files)."""
import json
import os
import subprocess

import numpy
import pytest

from pauxy_amd.utils import h5lite
from pauxy_amd.utils import io as aio
from pauxy_amd.estimators.utils import H5EstimatorHelper
from tests import hdf5_ctypes as h5c

pytestmark = pytest.mark.skipif(not h5c.available(), reason="no libhdf5 in this image")
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = numpy.load(os.path.join(HERE, "golden", "io_formats.npz"), allow_pickle=False)


def flatten(group, prefix=''):
    out = {}
    for k, v in group.items():
        if isinstance(v, h5lite.Group):
            out.update(flatten(v, prefix + k + '/'))
        else:
            out[prefix + k] = v[()] if v.shape == () else v[:]
    return out


def check_with_libhdf5(path):
    """every fixed-size dataset h5lite finds in the file, re-read through libhdf5"""
    with h5lite.File(path, 'r') as f:
        mine = flatten(f)
    assert mine
    n = 0
    for name, want in mine.items():
        want = numpy.asarray(want)
        if want.dtype.kind in 'OU' or isinstance(want[()] if want.shape == () else None, (str, bytes)):
            continue                                           # variable-length strings: checked through h5dump below
        got = h5c.read_dataset(path, '/' + name)
        assert got.shape == want.shape, name
        if want.dtype.kind == 'S':
            assert [x.rstrip(b'\0') for x in got.ravel()] == [x.rstrip(b'\0') for x in want.ravel()], name
        else:
            assert got.dtype == want.dtype, (name, got.dtype, want.dtype)
            assert numpy.array_equal(got, want), name
        n += 1
    return n


def test_libhdf5_reads_hamiltonian_and_wavefunction_files(tmp_path):
    nelec = tuple(int(x) for x in GOLD['nelec'])
    p = str(tmp_path / 'dense.h5')
    aio.write_qmcpack_dense(GOLD['hc'], GOLD['cholc'], nelec, 5, enuc=1.25, filename=p, real_chol=False, ortho=GOLD['X'])
    assert check_with_libhdf5(p) >= 4
    p = str(tmp_path / 'sparse.h5')
    aio.write_qmcpack_sparse(GOLD['h'], GOLD['chol'], nelec, 5, enuc=1.25, filename=p, real_chol=True)
    assert check_with_libhdf5(p) >= 5
    p = str(tmp_path / 'wfn.h5')
    aio.write_qmcpack_wfn(p, (GOLD['coeffs'].copy(), GOLD['dets'].copy()), 'uhf', (2, 2), 5)
    assert check_with_libhdf5(p) >= 3


def test_libhdf5_reads_estimator_file_with_appended_blocks(tmp_path):
    """the file grows block by block in 'a' mode (estimators/utils.py:279-327): groups gain entries after the first
    close, which is where a hand-written symbol-table writer would go wrong first"""
    path = str(tmp_path / 'estimates.0.h5')
    with aio.h5.File(path, 'w'):
        pass
    with aio.h5.File(path, 'a') as f:
        f['basic/headers'] = numpy.array(['Iteration', 'Weight', 'ETotal']).astype('S')
        f['metadata'] = json.dumps({'qmc': {'dt': 0.01}})
    out = H5EstimatorHelper(path, 'basic', flush_every=3)
    rows = [numpy.array([i, i + 0.25j, -1.5 * i]) for i in range(40)]
    for r in rows:
        out.push(r, 'energies')
        out.push(numpy.full((2, 3, 3), r[1], dtype=numpy.complex128), 'one_rdm')
        out.increment()
    out.flush()
    assert check_with_libhdf5(path) == 81
    for i in (0, 17, 39):
        assert numpy.array_equal(h5c.read_dataset(path, '/basic/energies/%09d' % i), rows[i])
    # the variable-length metadata string and the tree, through the library's own tool
    tool = '/opt/conda/bin/h5dump'
    if os.path.exists(tool):
        txt = subprocess.run([tool, '-d', '/metadata', path], capture_output=True, text=True, timeout=60)
        assert txt.returncode == 0 and '"dt": 0.01' in txt.stdout.replace('\\"', '"')
        hdr = subprocess.run([tool, '-H', path], capture_output=True, text=True, timeout=60)
        assert hdr.returncode == 0 and hdr.stdout.count('DATASET "0000000') == 80


def test_libhdf5_reads_restart_style_file(tmp_path):
    """walkers/handler.py:146-161 layout: one flat complex dataset per walker"""
    path = str(tmp_path / 'restart.h5')
    rng = numpy.random.RandomState(3)
    bufs = [rng.rand(1 + 1 + 1 + 12 * 6) + 1j * rng.rand(75) for _ in range(7)]
    with aio.h5.File(path, 'w') as f:
        for i, b in enumerate(bufs):
            f['walker_%d' % i] = b
    assert check_with_libhdf5(path) == 7


@pytest.mark.parametrize("chunks,deflate,shuffle", [(None, None, False), ((4, 5), None, False), ((4, 5), 4, False),
                                                     ((3, 7), 6, True), ((13, 11), 1, True)])
def test_h5lite_reads_what_libhdf5_writes(tmp_path, chunks, deflate, shuffle):
    rng = numpy.random.RandomState(11)
    a = rng.rand(13, 11)
    c = rng.rand(13, 11) + 1j * rng.rand(13, 11)
    k = numpy.arange(13 * 11, dtype=numpy.int32).reshape(13, 11) - 40
    path = str(tmp_path / 'lib.h5')
    h5c.write_file(path, {'a': (a, chunks, deflate, shuffle), 'c': (c, chunks, deflate, shuffle),
                          'k': (k, chunks, deflate, shuffle)})
    with h5lite.File(path, 'r') as f:
        assert sorted(f.keys()) == ['a', 'c', 'k']
        assert numpy.array_equal(f['a'][:], a)
        assert f['c'][:].dtype == numpy.complex128 and numpy.array_equal(f['c'][:], c)
        assert f['k'][:].dtype == numpy.int32 and numpy.array_equal(f['k'][:], k)
