"""A few libhdf5 calls through ctypes (tests only): the image carries a libhdf5 1.10 under /opt/conda/lib but no h5py,
so this is how the tests let the genuine library read what pauxy_amd/utils/h5lite.py writes, and write chunked /
deflated / shuffled datasets for h5lite to read.  ``available()`` is False when the library is not there."""
import ctypes
import glob

import numpy

_lib = None
H5P_DEFAULT = 0
H5S_ALL = 0
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_COMPOUND = 0, 1, 3, 6
hid_t = ctypes.c_int64


def _load():
    global _lib
    if _lib is not None:
        return _lib
    names = sorted(glob.glob('/opt/conda/lib/libhdf5.so*')) + ['libhdf5.so']
    for n in names:
        try:
            lib = ctypes.CDLL(n)
        except OSError:
            continue
        for f in ('H5Fopen', 'H5Fcreate', 'H5Dopen2', 'H5Dget_space', 'H5Dget_type', 'H5Tcreate', 'H5Tcopy',
                  'H5Screate_simple', 'H5Pcreate', 'H5Dcreate2', 'H5Dget_create_plist'):
            getattr(lib, f).restype = hid_t
        lib.H5Tget_size.restype = ctypes.c_size_t
        lib.H5open()
        _lib = lib
        return lib
    _lib = False
    return False


def available():
    return bool(_load())


def _gid(name):
    return hid_t.in_dll(_lib, name).value


def _ck(x, what):
    if x < 0:
        raise RuntimeError("libhdf5: %s failed" % what)
    return x


def read_dataset(path, name):
    """-> numpy array read by libhdf5 (f64, i32, i64, {r,i} complex128, fixed-length strings)."""
    lib = _load()
    f = _ck(lib.H5Fopen(path.encode(), H5F_ACC_RDONLY, hid_t(H5P_DEFAULT)), 'H5Fopen')
    try:
        d = _ck(lib.H5Dopen2(hid_t(f), name.encode(), hid_t(H5P_DEFAULT)), 'H5Dopen2 ' + name)
        sp = _ck(lib.H5Dget_space(hid_t(d)), 'H5Dget_space')
        nd = _ck(lib.H5Sget_simple_extent_ndims(hid_t(sp)), 'ndims')
        dims = (ctypes.c_uint64 * max(nd, 1))()
        lib.H5Sget_simple_extent_dims(hid_t(sp), dims, None)
        shape = tuple(int(dims[i]) for i in range(nd))
        t = _ck(lib.H5Dget_type(hid_t(d)), 'H5Dget_type')
        cls = lib.H5Tget_class(hid_t(t))
        size = int(lib.H5Tget_size(hid_t(t)))
        if cls == H5T_FLOAT and size == 8:
            out, mt = numpy.empty(shape, numpy.float64), _gid('H5T_NATIVE_DOUBLE_g')
        elif cls == H5T_INTEGER and size == 4:
            out, mt = numpy.empty(shape, numpy.int32), _gid('H5T_NATIVE_INT32_g')
        elif cls == H5T_INTEGER and size == 8:
            out, mt = numpy.empty(shape, numpy.int64), _gid('H5T_NATIVE_INT64_g')
        elif cls == H5T_COMPOUND and size == 16:
            mt = _ck(lib.H5Tcreate(H5T_COMPOUND, ctypes.c_size_t(16)), 'H5Tcreate')
            lib.H5Tinsert(hid_t(mt), b'r', ctypes.c_size_t(0), hid_t(_gid('H5T_NATIVE_DOUBLE_g')))
            lib.H5Tinsert(hid_t(mt), b'i', ctypes.c_size_t(8), hid_t(_gid('H5T_NATIVE_DOUBLE_g')))
            out = numpy.empty(shape, numpy.complex128)
        elif cls == H5T_STRING and not lib.H5Tis_variable_str(hid_t(t)):
            out, mt = numpy.empty(shape, 'S%d' % size), t
        else:
            raise NotImplementedError("class %d size %d" % (cls, size))
        _ck(lib.H5Dread(hid_t(d), hid_t(mt), hid_t(H5S_ALL), hid_t(H5S_ALL), hid_t(H5P_DEFAULT),
                        out.ctypes.data_as(ctypes.c_void_p)), 'H5Dread ' + name)
        lib.H5Dclose(hid_t(d))
        return out
    finally:
        lib.H5Fclose(hid_t(f))


def write_file(path, datasets):
    """datasets: {name (no groups): (array, chunks or None, deflate level or None, shuffle bool)}; f64 / i32 / c128."""
    lib = _load()
    f = _ck(lib.H5Fcreate(path.encode(), H5F_ACC_TRUNC, hid_t(H5P_DEFAULT), hid_t(H5P_DEFAULT)), 'H5Fcreate')
    try:
        for name, (a, chunks, deflate, shuffle) in datasets.items():
            a = numpy.ascontiguousarray(a)
            if a.dtype == numpy.float64:
                t = _gid('H5T_NATIVE_DOUBLE_g')
            elif a.dtype == numpy.int32:
                t = _gid('H5T_NATIVE_INT32_g')
            elif a.dtype == numpy.complex128:
                t = _ck(lib.H5Tcreate(H5T_COMPOUND, ctypes.c_size_t(16)), 'H5Tcreate')
                lib.H5Tinsert(hid_t(t), b'r', ctypes.c_size_t(0), hid_t(_gid('H5T_NATIVE_DOUBLE_g')))
                lib.H5Tinsert(hid_t(t), b'i', ctypes.c_size_t(8), hid_t(_gid('H5T_NATIVE_DOUBLE_g')))
            else:
                raise NotImplementedError(a.dtype)
            dims = (ctypes.c_uint64 * a.ndim)(*a.shape)
            sp = _ck(lib.H5Screate_simple(a.ndim, dims, None), 'H5Screate_simple')
            pl = _ck(lib.H5Pcreate(hid_t(_gid('H5P_CLS_DATASET_CREATE_ID_g'))), 'H5Pcreate')
            if chunks is not None:
                _ck(lib.H5Pset_chunk(hid_t(pl), a.ndim, (ctypes.c_uint64 * a.ndim)(*chunks)), 'H5Pset_chunk')
                if shuffle:
                    _ck(lib.H5Pset_shuffle(hid_t(pl)), 'H5Pset_shuffle')
                if deflate is not None:
                    _ck(lib.H5Pset_deflate(hid_t(pl), ctypes.c_uint(deflate)), 'H5Pset_deflate')
            d = _ck(lib.H5Dcreate2(hid_t(f), name.encode(), hid_t(t), hid_t(sp), hid_t(H5P_DEFAULT), hid_t(pl),
                                   hid_t(H5P_DEFAULT)), 'H5Dcreate2 ' + name)
            _ck(lib.H5Dwrite(hid_t(d), hid_t(t), hid_t(H5S_ALL), hid_t(H5S_ALL), hid_t(H5P_DEFAULT),
                             a.ctypes.data_as(ctypes.c_void_p)), 'H5Dwrite ' + name)
            lib.H5Dclose(hid_t(d))
            lib.H5Pclose(hid_t(pl))
            lib.H5Sclose(hid_t(sp))
    finally:
        lib.H5Fclose(hid_t(f))
