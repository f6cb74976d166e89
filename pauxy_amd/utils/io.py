"""On-disk formats either side of the device path.

Input side -- QMCPACK-style HDF5 Hamiltonians and trial wavefunctions, the files
PAUXY's ``Generic`` system and ``MultiSlater`` trial are built from:

  Hamiltonian/{Energies, hcore, dims, DenseFactorized/L}                 dense   (pauxy/utils/io.py:176-214)
  Hamiltonian/{Energies, hcore, dims, occups, Factorized/{block_sizes,
               index_i, vals_i}}                                          sparse  (pauxy/utils/io.py:81-174)
  Wavefunction/NOMSD/{dims, ci_coeffs, Psi0_alpha[, Psi0_beta],
               PsiT_i/{dims, data_, jdata_, pointers_begin_, pointers_end_}}     (pauxy/utils/io.py:339-374,454-515)
  Wavefunction/PHMSD/{dims, ci_coeffs, occs, Psi0_alpha, Psi0_beta, fullmo, type} (pauxy/utils/io.py:376-405,517-544)

Complex arrays are stored the QMCPACK way, as float64 with a trailing axis of
length 2 (pauxy/utils/io.py:126-127,562-564).  Function names and argument order
follow the reference so a PAUXY script can import them from here unchanged.

The HDF5 container is ``h5py`` when it is importable, otherwise the built-in
``pauxy_amd.utils.h5lite`` (same file layout, same tiny API).
"""
import numpy
import scipy.sparse

try:                                        # pragma: no cover - depends on the image
    import h5py as h5
    HAVE_H5PY = True
except ImportError:
    from pauxy_amd.utils import h5lite as h5
    HAVE_H5PY = False


# ------------------------------------------------------------------ complex
def to_qmcpack_complex(array):
    """complex128 [..] -> float64 [.., 2] (io.py:562-564)."""
    array = numpy.ascontiguousarray(array, dtype=numpy.complex128)
    return array.view(numpy.float64).reshape(array.shape + (2,))


def from_qmcpack_complex(data, shape):
    """float64 [.., 2] -> complex128 reshaped to ``shape`` (io.py:126-127)."""
    data = numpy.ascontiguousarray(data, dtype=numpy.float64)
    return data.view(numpy.complex128).ravel().reshape(shape)


def _is_pair_array(a):
    return a.ndim >= 2 and a.shape[-1] == 2


# -------------------------------------------------------------- Hamiltonian
def to_sparse(vals, offset=0, cutoff=1e-8):
    """Non-zeros of a 2-D array as (interleaved [row, col, row, col, ...] int32, complex values) (io.py:73-79)."""
    rows, cols = numpy.nonzero(numpy.abs(vals) > cutoff)
    ix = numpy.empty(2 * rows.size, dtype=numpy.int32)
    ix[0::2] = rows
    ix[1::2] = cols
    return ix, numpy.asarray(vals[rows, cols], dtype=numpy.complex128)


def write_qmcpack_dense(hcore, chol, nelec, nmo, enuc=0.0, filename='hamiltonian.h5', real_chol=True,
                        verbose=False, ortho=None):
    """Dense Cholesky Hamiltonian; ``chol`` is [nmo*nmo, nchol] (io.py:176-193)."""
    chol = numpy.asarray(chol)
    if chol.ndim != 2 or chol.shape[0] != nmo * nmo:
        raise ValueError("chol must have shape (nmo*nmo, nchol)")
    with h5.File(filename, 'w') as fh5:
        fh5['Hamiltonian/Energies'] = numpy.array([enuc, 0.0])
        if real_chol:
            fh5['Hamiltonian/hcore'] = numpy.ascontiguousarray(numpy.real(hcore), dtype=numpy.float64)
            fh5['Hamiltonian/DenseFactorized/L'] = numpy.ascontiguousarray(numpy.real(chol), dtype=numpy.float64)
        else:
            fh5['Hamiltonian/hcore'] = to_qmcpack_complex(hcore)
            fh5['Hamiltonian/DenseFactorized/L'] = to_qmcpack_complex(chol)
        fh5['Hamiltonian/dims'] = numpy.array([0, 0, 0, nmo, nelec[0], nelec[1], 0, chol.shape[-1]])
        if ortho is not None:
            fh5['Hamiltonian/X'] = numpy.asarray(ortho)


def from_qmcpack_dense(filename):
    """-> (hcore [nmo,nmo], chol [nmo*nmo, nchol], enuc, nmo, nalpha, nbeta) (io.py:195-214).

    Raises KeyError when the file holds the sparse layout (the reference's dispatch relies on that)."""
    with h5.File(filename, 'r') as fh5:
        enuc = float(fh5['Hamiltonian/Energies'][:][0])
        dims = fh5['Hamiltonian/dims'][:]
        nmo = int(dims[3])
        chol = fh5['Hamiltonian/DenseFactorized/L'][:]
        hcore = fh5['Hamiltonian/hcore'][:]
    if hcore.ndim == 3 and hcore.shape[-1] == 2:
        hcore = from_qmcpack_complex(hcore, (nmo, nmo))
        chol = from_qmcpack_complex(chol, (nmo * nmo, -1))
    return hcore, chol, enuc, nmo, int(dims[4]), int(dims[5])


def write_qmcpack_sparse(hcore, chol, nelec, nmo, enuc=0.0, filename='hamiltonian.h5', real_chol=False,
                         verbose=False, cutoff=1e-16, ortho=None):
    """Sparse (COO triplets in one block) Cholesky Hamiltonian (io.py:81-124)."""
    chol = numpy.asarray(chol)
    ix, vals = to_sparse(chol, cutoff=cutoff)
    nnz = len(vals)
    nalpha, nbeta = nelec
    with h5.File(filename, 'w') as fh5:
        fh5['Hamiltonian/Energies'] = numpy.array([enuc, 0.0])
        if real_chol:
            fh5['Hamiltonian/hcore'] = numpy.ascontiguousarray(numpy.real(hcore), dtype=numpy.float64)
        else:
            fh5['Hamiltonian/hcore'] = to_qmcpack_complex(hcore)
        if ortho is not None:
            fh5['Hamiltonian/X'] = numpy.asarray(ortho)
        fh5['Hamiltonian/Factorized/block_sizes'] = numpy.array([nnz])
        fh5['Hamiltonian/Factorized/index_0'] = ix
        # the reference stores complex128 values here even with real_chol (io.py:107-108)
        fh5['Hamiltonian/Factorized/vals_0'] = vals if real_chol else to_qmcpack_complex(vals)
        fh5['Hamiltonian/dims'] = numpy.array([0, nnz, 1, nmo, nalpha, nbeta, 0, chol.shape[-1]])
        fh5['Hamiltonian/occups'] = numpy.array(list(range(nalpha)) + [i + nmo for i in range(nbeta)])
    if verbose:
        print(" # Non-zero elements of the Cholesky tensor: %d (sparsity %f)" % (nnz, 1.0 - nnz / max(chol.size, 1)))


def from_qmcpack_sparse(filename):
    """-> (hcore, chol as scipy.sparse.csr_matrix [nmo*nmo, nchol], enuc, nmo, nalpha, nbeta) (io.py:129-174).

    Raises KeyError when the file holds the dense layout."""
    with h5.File(filename, 'r') as fh5:
        enuc = float(fh5['Hamiltonian/Energies'][:][0])
        dims = fh5['Hamiltonian/dims'][:]
        nmo = int(dims[3])
        nchol = int(dims[7])
        block_sizes = fh5['Hamiltonian/Factorized/block_sizes'][:]
        if 'Hamiltonian/hcore' in fh5:
            hcore = fh5['Hamiltonian/hcore'][:]
            real_ints = not (hcore.ndim == 3 and hcore.shape[-1] == 2)
            if not real_ints:
                hcore = from_qmcpack_complex(hcore, (nmo, nmo))
        else:
            # older files: lower triangle of hcore as COO triplets (io.py:139-146)
            v = from_qmcpack_complex(fh5['Hamiltonian/H1'][:], (-1,))
            idx = fh5['Hamiltonian/H1_indx'][:]
            low = scipy.sparse.csr_matrix((v, (idx[::2], idx[1::2])), shape=(nmo, nmo)).toarray()
            hcore = numpy.tril(low, -1) + numpy.tril(low, 0).conj().T
            real_ints = False
        rows, cols, vals = [], [], []
        for ic, bs in enumerate(block_sizes):
            ixs = fh5['Hamiltonian/Factorized/index_%i' % ic][:]
            v = fh5['Hamiltonian/Factorized/vals_%i' % ic][:]
            if numpy.iscomplexobj(v):
                v = v.ravel()
            elif _is_pair_array(v):
                v = from_qmcpack_complex(v, (-1,))
            else:
                v = v.ravel()
            rows.append(ixs[0:2 * bs:2])
            cols.append(ixs[1:2 * bs:2])
            vals.append(numpy.real(v[:bs]) if real_ints else v[:bs])
    vals = numpy.concatenate(vals) if vals else numpy.zeros(0)
    rows = numpy.concatenate(rows).astype(numpy.int64) if rows else numpy.zeros(0, dtype=numpy.int64)
    cols = numpy.concatenate(cols).astype(numpy.int64) if cols else numpy.zeros(0, dtype=numpy.int64)
    chol = scipy.sparse.csr_matrix((vals, (rows, cols)), shape=(nmo * nmo, nchol))
    return hcore, chol, enuc, nmo, int(dims[4]), int(dims[5])


def read_integrals(integral_file):
    """-> (hcore, dense chol [nmo*nmo, nchol], ecore); sparse layout first, dense second
    (pauxy/systems/generic.py:184-199).  Unlike the reference an unreadable file raises."""
    try:
        hcore, schol, ecore, nmo, na, nb = from_qmcpack_sparse(integral_file)
        return hcore, schol.toarray(), ecore
    except KeyError:
        hcore, chol, ecore, nmo, na, nb = from_qmcpack_dense(integral_file)
        return hcore, chol, ecore


def read_qmcpack_hamiltonian(filename):
    """-> dict(hcore, chol, enuc, nmo, nelec) whichever layout the file uses."""
    try:
        hcore, chol, enuc, nmo, na, nb = from_qmcpack_sparse(filename)
        chol = chol.toarray()
    except KeyError:
        hcore, chol, enuc, nmo, na, nb = from_qmcpack_dense(filename)
    return dict(hcore=hcore, chol=chol, enuc=enuc, nmo=nmo, nelec=(na, nb))


# ------------------------------------------------------------ wavefunctions
def orbs_from_dset(dset):
    """One NOMSD component.  The file holds A^H as CSR triplets; returns A [nmo, nocc] (io.py:546-560)."""
    dims = dset['dims'][:]
    nrow, ncol, nnz = int(dims[0]), int(dims[1]), int(dims[2])
    data = from_qmcpack_complex(dset['data_'][:], (nnz,))
    indices = dset['jdata_'][:]
    indptr = numpy.zeros(nrow + 1, dtype=numpy.int64)
    indptr[:-1] = dset['pointers_begin_'][:]
    indptr[-1] = dset['pointers_end_'][:][-1] if nrow else 0
    ah = scipy.sparse.csr_matrix((data, indices, indptr), shape=(nrow, ncol))
    return ah.toarray().conj().T.copy()


def _check_nelec(na, nb, nelec):
    # the reference compares nb with nelec[0] (io.py:347,384); the intended check is per spin
    if nelec is not None and (na != nelec[0] or nb != nelec[1]):
        raise ValueError("Number of electrons does not match wavefunction: (%d, %d) vs %s." % (na, nb, tuple(nelec)))


def _read_psi0(wgroup, nmo, na, nb, uhf, closed_shell_copy):
    psi0 = numpy.zeros((nmo, na + nb), dtype=numpy.complex128)
    psi0a = from_qmcpack_complex(wgroup['Psi0_alpha'][:], (nmo, na))
    psi0[:, :na] = psi0a
    if uhf:
        psi0[:, na:] = from_qmcpack_complex(wgroup['Psi0_beta'][:], (nmo, nb))
    else:
        psi0[:, na:] = psi0a[:, :nb] if closed_shell_copy else psi0a
    return psi0


def read_qmcpack_nomsd_hdf5(wgroup, nelec=None):
    """-> ((coeffs [nci], wfn [nci, nmo, na+nb]), psi0 [nmo, na+nb]) (io.py:339-374).

    UHF files hold alpha of determinant d in PsiT_<2d> and beta in PsiT_<2d+1> (write_nomsd, io.py:486-494).
    The reference reader computes that index but then reads alpha from PsiT_<d> (io.py:364-366), which is
    only right for d = 0; this reader uses the writer's (QMCPACK's) convention for every determinant."""
    dims = wgroup['dims'][:]
    nmo, na, nb, walker_type, nci = (int(x) for x in dims[:5])
    _check_nelec(na, nb, nelec)
    uhf = walker_type == 2
    coeffs = from_qmcpack_complex(wgroup['ci_coeffs'][:], (nci,))
    psi0 = _read_psi0(wgroup, nmo, na, nb, uhf, True)
    wfn = numpy.zeros((nci, nmo, na + nb), dtype=numpy.complex128)
    for idet in range(nci):
        pa = orbs_from_dset(wgroup['PsiT_%d' % (2 * idet if uhf else idet)])
        wfn[idet, :, :na] = pa
        wfn[idet, :, na:] = orbs_from_dset(wgroup['PsiT_%d' % (2 * idet + 1)]) if uhf else pa[:, :nb]
    return (coeffs, wfn), psi0


def read_qmcpack_phmsd_hdf5(wgroup, nelec=None):
    """-> ((coeffs, occa [nci,na], occb [nci,nb]), psi0) (io.py:376-405)."""
    dims = wgroup['dims'][:]
    nmo, na, nb, walker_type, nci = (int(x) for x in dims[:5])
    _check_nelec(na, nb, nelec)
    coeffs = from_qmcpack_complex(wgroup['ci_coeffs'][:], (nci,))
    occs = wgroup['occs'][:].reshape((nci, na + nb))
    psi0 = _read_psi0(wgroup, nmo, na, nb, walker_type == 2, False)
    return (coeffs, occs[:, :na], occs[:, na:] - nmo), psi0


def read_qmcpack_wfn_hdf(filename, nelec=None):
    """NOMSD first, PHMSD second (io.py:325-337); a file with neither raises KeyError."""
    with h5.File(filename, 'r') as fh5:
        if 'Wavefunction/NOMSD' in fh5:
            return read_qmcpack_nomsd_hdf5(fh5['Wavefunction/NOMSD'], nelec=nelec)
        if 'Wavefunction/PHMSD' in fh5:
            return read_qmcpack_phmsd_hdf5(fh5['Wavefunction/PHMSD'], nelec=nelec)
    raise KeyError("no Wavefunction/NOMSD or Wavefunction/PHMSD group in %s" % filename)


def write_nomsd_single(fh5, psi, idet):
    """CSR triplets of one component (``psi`` = A^H as scipy CSR) under PsiT_<idet>/ (io.py:497-515)."""
    base = 'PsiT_%d/' % idet
    fh5[base + 'dims'] = numpy.array([psi.shape[0], psi.shape[1], psi.nnz], dtype=numpy.int32)
    fh5[base + 'data_'] = to_qmcpack_complex(psi.data)
    fh5[base + 'jdata_'] = numpy.asarray(psi.indices, dtype=numpy.int32)
    fh5[base + 'pointers_begin_'] = numpy.asarray(psi.indptr[:-1], dtype=numpy.int32)
    fh5[base + 'pointers_end_'] = numpy.asarray(psi.indptr[1:], dtype=numpy.int32)


def write_nomsd(fh5, wfn, uhf, nelec, thresh=1e-8, init=None):
    """NOMSD determinants ``wfn`` [ndet, nmo, na+nb] (or one [nmo, nel]) into group ``fh5`` (io.py:454-495)."""
    nalpha, nbeta = nelec
    wfn = numpy.array(wfn, dtype=numpy.complex128)
    if wfn.ndim == 2:
        wfn = wfn[None]
    wfn[numpy.abs(wfn) < thresh] = 0.0
    if init is not None:
        fh5['Psi0_alpha'] = to_qmcpack_complex(init[0])
        fh5['Psi0_beta'] = to_qmcpack_complex(init[1])
    else:
        fh5['Psi0_alpha'] = to_qmcpack_complex(wfn[0, :, :nalpha].copy())
        if uhf:
            fh5['Psi0_beta'] = to_qmcpack_complex(wfn[0, :, nalpha:].copy())
    for idet, w in enumerate(wfn):
        write_nomsd_single(fh5, scipy.sparse.csr_matrix(w[:, :nalpha].conj().T), 2 * idet if uhf else idet)
        if uhf:
            write_nomsd_single(fh5, scipy.sparse.csr_matrix(w[:, nalpha:].conj().T), 2 * idet + 1)


def write_phmsd(fh5, occa, occb, nelec, norb, init=None):
    """Particle-hole expansion as occupation lists (beta orbitals offset by ``norb``) (io.py:517-544)."""
    na, nb = nelec
    if init is not None:
        fh5['Psi0_alpha'] = to_qmcpack_complex(numpy.array(init[0], dtype=numpy.complex128))
        fh5['Psi0_beta'] = to_qmcpack_complex(numpy.array(init[1], dtype=numpy.complex128))
    else:
        eye = numpy.eye(norb, dtype=numpy.complex128)
        fh5['Psi0_alpha'] = to_qmcpack_complex(eye[:, numpy.asarray(occa[0])].copy())
        fh5['Psi0_beta'] = to_qmcpack_complex(eye[:, numpy.asarray(occb[0])].copy())
    fh5['fullmo'] = numpy.array([0], dtype=numpy.int32)
    fh5['type'] = 0
    occs = numpy.zeros((len(occa), na + nb), dtype=numpy.int32)
    occs[:, :na] = numpy.array(occa)
    occs[:, na:] = norb + numpy.array(occb)
    fh5['occs'] = occs.ravel()


def write_qmcpack_wfn(filename, wfn, walker_type, nelec, norb, init=None, mode='w'):
    """``wfn`` = (coeffs, dets) for NOMSD or (coeffs, occa, occb) for PHMSD; ``walker_type`` in
    'rhf' / 'uhf' / 'ghf' (io.py:407-452).  An existing group of the same kind is replaced."""
    if len(wfn) == 3:
        coeffs, occa, occb = wfn
        kind = 'PHMSD'
    elif len(wfn) == 2:
        coeffs, dets = wfn
        kind = 'NOMSD'
    else:
        raise ValueError("Unknown wavefunction type passed.")
    uhf = walker_type == 'uhf'
    wtype = {'ghf': 3, 'uhf': 2}.get(walker_type, 1)
    if kind == 'PHMSD':
        wtype = 2
    coeffs = numpy.array(coeffs, dtype=numpy.complex128)
    with h5.File(filename, mode) as fh5:
        path = 'Wavefunction/' + kind
        if path in fh5:
            del fh5[path]
        group = fh5.create_group(path)
        if kind == 'NOMSD':
            write_nomsd(group, dets, uhf, nelec, init=init)
        else:
            write_phmsd(group, occa, occb, nelec, norb, init=init)
        group['ci_coeffs'] = to_qmcpack_complex(coeffs)
        group['dims'] = numpy.array([norb, nelec[0], nelec[1], wtype, len(coeffs)], dtype=numpy.int32)


# ------------------------------------------------------- estimator file readers
def get_param(filename, path):
    """Entry of the ``metadata`` JSON of an estimator file (pauxy/analysis/extraction.py:86-95)."""
    import json
    with h5.File(filename, 'r') as fh5:
        doc = json.loads(fh5['metadata'][()])
    for key in path:
        doc = doc[key]
    return doc


def extract_data(filename, group, estimator, raw=False):
    """Blocks of one estimator (pauxy/analysis/extraction.py:14-28) without pandas: ``raw`` or an ``*rdm*``
    estimator -> array [nblocks, ...]; otherwise a dict {header name: column [nblocks]} with the real parts
    (complex columns for a free-projection run, as in the reference)."""
    try:
        fp = bool(get_param(filename, ['propagators', 'free_projection']))
    except KeyError:
        fp = False
    with h5.File(filename, 'r') as fh5:
        names = sorted(fh5[group][estimator].keys())
        data = numpy.array([fh5[group][estimator][d][:] for d in names])
        if 'rdm' in estimator or raw:
            return data
        header = [x.decode('utf-8') for x in fh5[group]['headers'][:]]
    if not fp:
        data = numpy.real(data)
    return {name: data[:, i] for i, name in enumerate(header)}


def extract_mixed_estimates(filename, skip=0):
    """analysis/extraction.py:30-31."""
    return {k: v[skip:] for k, v in extract_data(filename, 'basic', 'energies').items()}


def extract_bp_estimates(filename, skip=0):
    """analysis/extraction.py:33-34."""
    return {k: v[skip:] for k, v in extract_data(filename, 'back_propagated', 'energies').items()}


def extract_rdm(filename, est_type='back_propagated', rdm_type='one_rdm', ix=None):
    """Back-propagated density matrices divided by their denominators (analysis/extraction.py:36-62)."""
    if ix is None:
        ix = get_param(filename, ['estimators', 'estimators', 'back_prop', 'splits'])[0][-1]
    one_rdm = extract_data(filename, est_type, '%s_%d' % (rdm_type, ix), raw=True)
    denom = extract_data(filename, est_type, 'denominator_%d' % ix, raw=True)
    return one_rdm / denom.reshape((-1,) + (1,) * (one_rdm.ndim - 1))
