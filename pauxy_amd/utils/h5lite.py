"""Self-contained reader / writer for the subset of HDF5 that PAUXY's files use.

PAUXY reads and writes every on-disk object through ``h5py`` (Hamiltonian and
wavefunction input: pauxy/utils/io.py:81-214,325-545; estimator output:
pauxy/estimators/utils.py:279-327, estimators/handler.py:70,119-120; walker
restart files: walkers/handler.py:151-155,444-485).  ``h5py`` is an optional
dependency here: ``pauxy_amd.utils.io`` uses it when it imports and this module
otherwise, so the formats either side of the device path do not depend on a
package the target image does not ship.

Supported on *read* (what libhdf5 writes with default settings, which is what
h5py produces for PAUXY/QMCPACK files): superblock versions 0-3, version-1 object
headers with continuation blocks, version-2 object headers with compact link
messages, symbol-table groups (v1 B-tree + local heap + symbol nodes), simple and
scalar dataspaces, fixed-point / IEEE float / fixed string / compound (complex)
/ variable-length string datatypes, compact, contiguous and chunked layouts with
the deflate and shuffle filters, attributes (v1-v3).  Anything else raises
``NotImplementedError`` naming the feature.

Written files use the same "classic" structures (superblock 0, v1 object headers,
symbol-table groups, contiguous datasets, complex as ``{r,i}`` compounds,
``str`` as variable-length UTF-8 in a global heap collection), i.e. the layout of
the reference's own fixture pauxy/trial_wavefunction/tests/wfn.h5.

The API is the slice of h5py's that PAUXY touches: ``File(name, mode)`` as a
context manager, ``f['a/b/c']``, ``f['a/b'] = array``, ``in``, ``del``, ``keys``,
``create_group``, ``create_dataset``, ``dset[...]``, ``dset[...] = x``,
``dset.shape/dtype``, ``KeyError`` for a missing name.
"""
import os
import struct
import zlib

import numpy

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


# ============================================================================ reading
class _Reader(object):
    def __init__(self, fh):
        self.fh = fh
        self.so = 8          # size of offsets
        self.sl = 8          # size of lengths
        self.base = 0
        self._gcol = {}

    def read(self, addr, n):
        self.fh.seek(self.base + addr)
        b = self.fh.read(n)
        if len(b) != n:
            raise IOError("truncated HDF5 file (wanted %d bytes at %d)" % (n, addr))
        return b

    def uint(self, buf, pos, n):
        return int.from_bytes(buf[pos:pos + n], 'little')

    # ------------------------------------------------------------ superblock
    def superblock(self):
        pos = 0
        while True:                      # the superblock may sit at 0, 512, 1024, ...
            self.fh.seek(pos)
            if self.fh.read(8) == SIGNATURE:
                break
            pos = 512 if pos == 0 else pos * 2
            if pos > (1 << 24):
                raise IOError("not an HDF5 file")
        self.fh.seek(pos)
        head = self.fh.read(128)
        ver = head[8]
        if ver in (0, 1):
            self.so, self.sl = head[13], head[14]
            p = 24 if ver == 0 else 28
            self.base = self.uint(head, p, self.so)
            p += 4 * self.so
            # root symbol table entry: name offset, header address, cache type, reserved, scratch
            root = self.uint(head, p + self.so, self.so)
            return root
        if ver in (2, 3):
            self.so, self.sl = head[9], head[10]
            p = 12
            self.base = self.uint(head, p, self.so)
            return self.uint(head, p + 3 * self.so, self.so)
        raise NotImplementedError("HDF5 superblock version %d" % ver)

    # -------------------------------------------------------- object headers
    def messages(self, addr):
        """[(type, flags, bytes)] of the object header at ``addr`` (continuations followed)."""
        head = self.read(addr, 16)
        if head[:4] == b'OHDR':
            return self._messages_v2(addr)
        ver, nmsg, size = head[0], struct.unpack_from('<H', head, 2)[0], struct.unpack_from('<I', head, 8)[0]
        if ver != 1:
            raise NotImplementedError("object header version %d" % ver)
        out = []
        blocks = [(addr + 16, size)]
        while blocks and len(out) < nmsg:
            a, n = blocks.pop(0)
            buf = self.read(a, n)
            p = 0
            while p + 8 <= n and len(out) < nmsg:
                t, s, fl = struct.unpack_from('<HHB', buf, p)
                body = buf[p + 8:p + 8 + s]
                p += 8 + s
                if t == 0x10:
                    blocks.append((self.uint(body, 0, self.so), self.uint(body, self.so, self.sl)))
                out.append((t, fl, body))
        return out

    def _messages_v2(self, addr):
        head = self.read(addr, 6)
        flags = head[5]
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        nsz = 1 << (flags & 3)
        size = self.uint(self.read(p, nsz), 0, nsz)
        p += nsz
        out = []
        blocks = [(p, size)]
        while blocks:
            a, n = blocks.pop(0)
            buf = self.read(a, n)
            q = 0
            while q + 4 <= n:
                t = buf[q]
                s = struct.unpack_from('<H', buf, q + 1)[0]
                fl = buf[q + 3]
                q += 4
                if flags & 0x04:
                    q += 2
                body = buf[q:q + s]
                q += s
                if t == 0x10:
                    ca, cn = self.uint(body, 0, self.so), self.uint(body, self.so, self.sl)
                    blocks.append((ca + 4, cn - 8))           # skip "OCHK", drop the checksum
                elif t != 0:
                    out.append((t, fl, body))
        return out

    # ---------------------------------------------------------------- groups
    def group_links(self, msgs):
        """{name: object header address} of a group given its header messages."""
        links = {}
        for t, fl, body in msgs:
            if t == 0x11:
                btree = self.uint(body, 0, self.so)
                heap = self.uint(body, self.so, self.so)
                self._walk_group_btree(btree, self._local_heap(heap), links)
            elif t == 0x06:
                name, target = self._link_message(body)
                links[name] = target
            elif t == 0x02:
                # link info: dense storage if the fractal heap address is defined
                p = 2 + (8 if body[1] & 1 else 0)
                if self.uint(body, p, self.so) != UNDEF >> (64 - 8 * self.so):
                    raise NotImplementedError("HDF5 dense link storage (fractal heap groups)")
        return links

    def _link_message(self, body):
        flags = body[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = body[p]
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        nsz = 1 << (flags & 3)
        n = self.uint(body, p, nsz)
        p += nsz
        name = body[p:p + n].decode('utf-8')
        p += n
        if ltype != 0:
            raise NotImplementedError("HDF5 soft/external link '%s'" % name)
        return name, self.uint(body, p, self.so)

    def _local_heap(self, addr):
        head = self.read(addr, 8 + 2 * self.sl + self.so)
        if head[:4] != b'HEAP':
            raise IOError("bad local heap signature at %d" % addr)
        size = self.uint(head, 8, self.sl)
        data = self.uint(head, 8 + 2 * self.sl, self.so)
        return self.read(data, size)

    def _walk_group_btree(self, addr, heap, links):
        head = self.read(addr, 8 + 2 * self.so)
        if head[:4] != b'TREE' or head[4] != 0:
            raise IOError("bad group B-tree node at %d" % addr)
        level, used = head[5], struct.unpack_from('<H', head, 6)[0]
        body = self.read(addr + 8 + 2 * self.so, (2 * used + 1) * max(self.so, self.sl))
        for i in range(used):
            child = self.uint(body, self.sl + i * (self.sl + self.so), self.so)
            if level > 0:
                self._walk_group_btree(child, heap, links)
            else:
                self._symbol_node(child, heap, links)

    def _symbol_node(self, addr, heap, links):
        head = self.read(addr, 8)
        if head[:4] != b'SNOD':
            raise IOError("bad symbol node at %d" % addr)
        n = struct.unpack_from('<H', head, 6)[0]
        esz = 2 * self.so + 24
        buf = self.read(addr + 8, n * esz)
        for i in range(n):
            off = self.uint(buf, i * esz, self.so)
            target = self.uint(buf, i * esz + self.so, self.so)
            end = heap.index(b'\0', off)
            links[heap[off:end].decode('utf-8')] = target

    # ------------------------------------------------------------- datatypes
    def datatype(self, body, p=0):
        """-> (numpy dtype or ('vlen_str',) marker, bytes consumed)."""
        cls, ver = body[p] & 0x0F, body[p] >> 4
        b0, b1, b2 = body[p + 1], body[p + 2], body[p + 3]
        size = struct.unpack_from('<I', body, p + 4)[0]
        q = p + 8
        order = '>' if b0 & 1 else '<'
        if cls == 0:
            kind = 'i' if b0 & 0x08 else 'u'
            return numpy.dtype('%s%s%d' % (order, kind, size)), q + 4 - p
        if cls == 1:
            return numpy.dtype('%sf%d' % (order, size)), q + 12 - p
        if cls == 3:
            return numpy.dtype('S%d' % size), q - p
        if cls == 6:
            nmemb = b0 | (b1 << 8)
            names, formats, offsets = [], [], []
            for _ in range(nmemb):
                end = body.index(b'\0', q)
                name = body[q:end].decode('utf-8')
                if ver < 3:
                    q += ((end - q) // 8 + 1) * 8
                else:
                    q = end + 1
                if ver < 3:
                    off = struct.unpack_from('<I', body, q)[0]
                    q += 4
                else:
                    nb = 1 if size < 256 else 2 if size < 65536 else 3 if size < 16777216 else 4
                    off = self.uint(body, q, nb)
                    q += nb
                if ver == 1:
                    q += 28                     # dimensionality, reserved, permutation, reserved, 4 dim sizes
                mt, used = self.datatype(body, q)
                q += used
                names.append(name)
                formats.append(mt)
                offsets.append(off)
            if names == ['r', 'i'] and formats[0] == formats[1] and formats[0].kind == 'f' and \
                    offsets == [0, formats[0].itemsize] and size == 2 * formats[0].itemsize:
                return numpy.dtype('%sc%d' % (formats[0].byteorder.replace('=', '<').replace('|', '<'), size)), q - p
            return numpy.dtype({'names': names, 'formats': formats, 'offsets': offsets, 'itemsize': size}), q - p
        if cls == 9:
            base, used = self.datatype(body, q)
            if (b0 & 0x0F) == 1:
                return ('vlen_str',), q + used - p
            raise NotImplementedError("HDF5 variable-length sequence datatype")
        if cls == 8:
            base, used = self.datatype(body, q + (0 if ver == 3 else 0))
            # enum (h5py stores bool this way): read as its integer base type
            nmemb = b0 | (b1 << 8)
            r = q + used
            for _ in range(nmemb):
                end = body.index(b'\0', r)
                r = r + ((end - r) // 8 + 1) * 8 if ver < 3 else end + 1
            r += nmemb * base.itemsize
            return base, r - p
        if cls == 10:
            ndim = body[q]
            q += 1 if ver == 3 else 4
            dims = struct.unpack_from('<%dI' % ndim, body, q)
            q += 4 * ndim
            if ver < 3:
                q += 4 * ndim
            base, used = self.datatype(body, q)
            return numpy.dtype((base, tuple(dims))), q + used - p
        raise NotImplementedError("HDF5 datatype class %d" % cls)

    def dataspace(self, body):
        ver, rank, flags = body[0], body[1], body[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            if body[3] == 2:
                return None                  # null dataspace
            p = 4
        else:
            raise NotImplementedError("dataspace message version %d" % ver)
        return tuple(self.uint(body, p + i * self.sl, self.sl) for i in range(rank))

    # ------------------------------------------------------------- raw data
    def global_heap_object(self, addr, index):
        if addr not in self._gcol:
            head = self.read(addr, 8 + self.sl)
            if head[:4] != b'GCOL':
                raise IOError("bad global heap collection at %d" % addr)
            size = self.uint(head, 8, self.sl)
            buf = self.read(addr, size)
            objs = {}
            p = 8 + self.sl
            while p + 8 + self.sl <= size:
                idx = struct.unpack_from('<H', buf, p)[0]
                n = self.uint(buf, p + 8, self.sl)
                if idx == 0:
                    break
                objs[idx] = buf[p + 8 + self.sl:p + 8 + self.sl + n]
                p += 8 + self.sl + ((n + 7) // 8) * 8
            self._gcol[addr] = objs
        return self._gcol[addr][index]

    def filters(self, body):
        ver, n = body[0], body[1]
        out = []
        p = 8 if ver == 1 else 2
        for _ in range(n):
            fid = struct.unpack_from('<H', body, p)[0]
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = struct.unpack_from('<H', body, p)[0]
                p += 2
            p += 2                                    # flags
            ncd = struct.unpack_from('<H', body, p)[0]
            p += 2
            p += ((nlen + 7) // 8) * 8 if ver == 1 else nlen
            cd = struct.unpack_from('<%dI' % ncd, body, p)
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, cd))
        return out

    def chunk_index(self, addr, rank, out):
        head = self.read(addr, 8 + 2 * self.so)
        if head[:4] != b'TREE' or head[4] != 1:
            raise IOError("bad chunk B-tree node at %d" % addr)
        level, used = head[5], struct.unpack_from('<H', head, 6)[0]
        ksz = 8 + 8 * (rank + 1)
        buf = self.read(addr + 8 + 2 * self.so, used * (ksz + self.so) + ksz)
        for i in range(used):
            p = i * (ksz + self.so)
            nbytes, mask = struct.unpack_from('<II', buf, p)
            offs = struct.unpack_from('<%dQ' % rank, buf, p + 8)
            child = self.uint(buf, p + ksz, self.so)
            if level > 0:
                self.chunk_index(child, rank, out)
            else:
                out.append((offs, nbytes, mask, child))


def _unshuffle(buf, itemsize):
    a = numpy.frombuffer(buf, dtype=numpy.uint8)
    n = a.size // itemsize
    return a[:n * itemsize].reshape(itemsize, n).T.tobytes() + bytes(a[n * itemsize:])


# ============================================================================ objects
class Dataset(object):
    def __init__(self, file, name, data=None, info=None):
        self.file = file
        self.name = name
        self._data = data             # ndarray / bytes / str when held in memory
        self._info = info             # (shape, dtype, layout dict, filters) when still on disk
        self.attrs = {}

    # ------------------------------------------------------------------ meta
    @property
    def shape(self):
        if self._data is not None:
            return numpy.shape(self._data) if isinstance(self._data, numpy.ndarray) else ()
        return self._info[0]

    @property
    def dtype(self):
        if self._data is not None:
            return self._data.dtype if isinstance(self._data, numpy.ndarray) else numpy.dtype('O')
        dt = self._info[1]
        return numpy.dtype('O') if isinstance(dt, tuple) else dt

    @property
    def size(self):
        return int(numpy.prod(self.shape, dtype=numpy.int64))

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        return self.shape[0]

    # ------------------------------------------------------------------ data
    def _load(self):
        if self._data is not None:
            return self._data
        shape, dt, layout, filters = self._info
        rd = self.file._reader
        if shape is None:
            return numpy.empty((0,), dtype=dt)
        vlen = isinstance(dt, tuple)
        edt = numpy.dtype([('n', '<u4'), ('a', '<u8'), ('i', '<u4')]) if vlen else dt
        count = int(numpy.prod(shape, dtype=numpy.int64))
        kind = layout['class']
        if kind == 0:
            raw = numpy.frombuffer(layout['data'][:count * edt.itemsize], dtype=edt).reshape(shape)
        elif kind == 1:
            if layout['addr'] == UNDEF or count == 0:
                raw = numpy.zeros(shape, dtype=edt)
            else:
                rd.fh.seek(rd.base + layout['addr'])
                raw = numpy.fromfile(rd.fh, dtype=edt, count=count).reshape(shape)
        else:
            raw = self._load_chunked(rd, shape, edt, layout, filters)
        if vlen:
            flat = [rd.global_heap_object(int(e['a']), int(e['i']))[:int(e['n'])] if e['n'] else b''
                    for e in raw.ravel()]
            if shape == ():
                return flat[0]
            out = numpy.empty(len(flat), dtype=object)
            out[:] = flat
            return out.reshape(shape)
        return raw

    def _load_chunked(self, rd, shape, edt, layout, filters):
        out = numpy.zeros(shape, dtype=edt)
        cdims = layout['chunk']
        rank = len(shape)
        chunks = []
        if layout['addr'] != UNDEF:
            rd.chunk_index(layout['addr'], rank, chunks)
        for offs, nbytes, mask, addr in chunks:
            buf = rd.read(addr, nbytes)
            for k, (fid, cd) in reversed(list(enumerate(filters))):
                if mask & (1 << k):
                    continue
                if fid == 1:
                    buf = zlib.decompress(buf)
                elif fid == 2:
                    buf = _unshuffle(buf, edt.itemsize)
                elif fid == 3:
                    buf = buf[:-4]            # fletcher32 checksum appended
                else:
                    raise NotImplementedError("HDF5 filter id %d" % fid)
            block = numpy.frombuffer(buf, dtype=edt, count=int(numpy.prod(cdims))).reshape(cdims)
            sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
            sel_in = tuple(slice(0, s.stop - s.start) for s in sel_out)
            out[sel_out] = block[sel_in]
        return out

    def __getitem__(self, key):
        a = self._load()
        if not isinstance(a, numpy.ndarray):
            if key == () or key is Ellipsis:
                return a
            raise ValueError("scalar string dataset")
        if a.shape == ():
            if key == () or key is Ellipsis:
                return a[()]
            if isinstance(key, slice) and key == slice(None):
                raise ValueError("Illegal slicing argument for scalar dataspace")
        r = a[key]
        return r.copy() if isinstance(r, numpy.ndarray) else r

    def __setitem__(self, key, value):
        self.file._check_writable()
        a = self._load()
        if not isinstance(a, numpy.ndarray):
            raise TypeError("cannot assign into a string dataset")
        if not a.flags.writeable:
            a = a.copy()
        a[key] = value
        self._data = a
        self._info = None
        self.file._dirty = True

    def __array__(self, dtype=None, copy=None):
        a = numpy.asarray(self._load())
        return a.astype(dtype) if dtype is not None else a

    def __repr__(self):
        return '<h5lite dataset "%s": shape %s, type "%s">' % (self.name, self.shape, self.dtype)


class Group(object):
    def __init__(self, file, name):
        self.file = file
        self.name = name
        self._members = None          # dict name -> Group | Dataset  (None: not read yet)
        self._addr = None
        self.attrs = {}

    # -------------------------------------------------------------- loading
    def _ensure(self):
        if self._members is None:
            self._members = {}
            if self._addr is not None:
                rd = self.file._reader
                msgs = rd.messages(self._addr)
                self.attrs = self.file._attributes(msgs)
                for name, addr in rd.group_links(msgs).items():
                    self._members[name] = self.file._object(addr, self._join(name))
        return self._members

    def _join(self, name):
        return (self.name.rstrip('/') + '/' + name)

    def _split(self, path):
        if isinstance(path, bytes):
            path = path.decode('utf-8')
        parts = [p for p in path.split('/') if p]
        start = self.file if path.startswith('/') else self
        return start, parts

    # ------------------------------------------------------------ mapping API
    def __getitem__(self, path):
        node, parts = self._split(path)
        for p in parts:
            if not isinstance(node, Group) or p not in node._ensure():
                raise KeyError("Unable to open object (object '%s' doesn't exist)" % p)
            node = node._members[p]
        return node

    def get(self, path, default=None):
        try:
            return self[path]
        except KeyError:
            return default

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def keys(self):
        return sorted(self._ensure().keys())

    def values(self):
        return [self._members[k] for k in self.keys()]

    def items(self):
        return [(k, self._members[k]) for k in self.keys()]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self._ensure())

    def _parent_for(self, path, create):
        node, parts = self._split(path)
        if not parts:
            raise ValueError("empty object name")
        for p in parts[:-1]:
            members = node._ensure()
            if p not in members:
                if not create:
                    raise KeyError(p)
                members[p] = Group(self.file, node._join(p))
                members[p]._members = {}
            node = members[p]
            if not isinstance(node, Group):
                raise ValueError("'%s' is not a group" % p)
        return node, parts[-1]

    def create_group(self, path):
        self.file._check_writable()
        parent, leaf = self._parent_for(path, True)
        if leaf in parent._ensure():
            raise ValueError("Unable to create group (name already exists)")
        g = Group(self.file, parent._join(leaf))
        g._members = {}
        parent._members[leaf] = g
        self.file._dirty = True
        return g

    def require_group(self, path):
        return self[path] if path in self else self.create_group(path)

    def create_dataset(self, path, shape=None, dtype=None, data=None, **unused):
        self.file._check_writable()
        if data is None:
            if shape is None:
                raise TypeError("one of data or shape is required")
            data = numpy.zeros(shape, dtype=dtype if dtype is not None else numpy.float32)
        elif isinstance(data, (str, bytes)):
            pass
        else:
            data = numpy.array(data, dtype=dtype)
            if shape is not None:
                data = data.reshape(shape)
        parent, leaf = self._parent_for(path, True)
        if leaf in parent._ensure():
            raise ValueError("Unable to create dataset (name already exists)")
        if isinstance(data, numpy.ndarray):
            data = _storable(data)
        d = Dataset(self.file, parent._join(leaf), data=data)
        parent._members[leaf] = d
        self.file._dirty = True
        return d

    def __setitem__(self, path, value):
        if isinstance(value, (Group, Dataset)):
            raise NotImplementedError("hard links to existing objects")
        self.create_dataset(path, data=value)

    def __delitem__(self, path):
        self.file._check_writable()
        parent, leaf = self._parent_for(path, False)
        if leaf not in parent._ensure():
            raise KeyError("Couldn't delete link (name doesn't exist)")
        del parent._members[leaf]
        self.file._dirty = True

    def __repr__(self):
        return '<h5lite group "%s" (%d members)>' % (self.name, len(self))


def _storable(a):
    """Array in a dtype this writer can describe (h5py conversions: bool -> enum is not supported)."""
    if a.dtype.kind == 'U':
        a = numpy.char.encode(a, 'utf-8')
    if a.dtype.kind == 'O':
        raise TypeError("object arrays cannot be stored")
    if a.dtype.kind not in 'iufcS':
        raise TypeError("dtype %s is not supported by h5lite" % a.dtype)
    return a if a.flags.c_contiguous else a.copy(order='C')


class File(Group):
    """h5py.File look-alike.  Modes: 'r', 'r+', 'a', 'w', 'w-'/'x'."""

    def __init__(self, name, mode='r', **unused):
        Group.__init__(self, self, '/')
        self.filename = os.fspath(name)
        self.mode = mode
        self._fh = None
        self._reader = None
        self._dirty = False
        exists = os.path.isfile(self.filename)
        if mode in ('r', 'r+') and not exists:
            raise OSError("Unable to open file (unable to open file: name = '%s')" % self.filename)
        if mode in ('w-', 'x') and exists:
            raise OSError("Unable to create file (file exists)")
        if mode not in ('r', 'r+', 'a', 'w', 'w-', 'x'):
            raise ValueError("invalid mode %r" % mode)
        if mode in ('r', 'r+') or (mode == 'a' and exists):
            self._fh = open(self.filename, 'rb')
            self._reader = _Reader(self._fh)
            self._addr = self._reader.superblock()
        else:
            self._members = {}
            self._dirty = True           # a new (possibly empty) file is written on close
        self._writable = mode != 'r'

    def _check_writable(self):
        if not self._writable:
            raise OSError("file is open read-only")
        if self._fh is None and self._members is None:
            raise ValueError("file is closed")

    # --------------------------------------------------------------- objects
    def _attributes(self, msgs):
        rd = self._reader
        out = {}
        for t, fl, body in msgs:
            if t != 0x0C:
                continue
            try:
                ver = body[0]
                nlen, tlen, slen = struct.unpack_from('<HHH', body, 2)
                p = 8 + (1 if ver == 3 else 0)
                pad = (lambda n: ((n + 7) // 8) * 8) if ver == 1 else (lambda n: n)
                name = body[p:p + nlen].split(b'\0')[0].decode('utf-8')
                p += pad(nlen)
                dt, _ = rd.datatype(body, p)
                p += pad(tlen)
                shape = rd.dataspace(body[p:p + slen])
                p += pad(slen)
                if isinstance(dt, tuple):
                    n, a, i = struct.unpack_from('<IQI', body, p)
                    out[name] = rd.global_heap_object(a, i)[:n].decode('utf-8')
                elif shape is not None:
                    cnt = int(numpy.prod(shape, dtype=numpy.int64))
                    val = numpy.frombuffer(body, dtype=dt, count=cnt, offset=p).reshape(shape)
                    out[name] = val[()] if shape == () else val.copy()
            except (NotImplementedError, ValueError, IndexError, struct.error):
                continue                    # attributes are informational on this path
        return out

    def _object(self, addr, name):
        rd = self._reader
        msgs = rd.messages(addr)
        types = set(t for t, _, _ in msgs)
        if 0x08 not in types:               # no data layout message: a group
            g = Group(self, name)
            g._addr = addr
            return g
        shape = dt = layout = None
        filters = []
        for t, fl, body in msgs:
            if t == 0x01:
                shape = rd.dataspace(body)
            elif t == 0x03:
                dt, _ = rd.datatype(body)
            elif t == 0x0B:
                filters = rd.filters(body)
            elif t == 0x08:
                layout = self._layout(rd, body)
        d = Dataset(self, name, info=(shape, dt, layout, filters))
        d.attrs = self._attributes(msgs)
        return d

    @staticmethod
    def _layout(rd, body):
        ver = body[0]
        if ver == 3:
            cls = body[1]
            if cls == 0:
                n = struct.unpack_from('<H', body, 2)[0]
                return {'class': 0, 'data': body[4:4 + n]}
            if cls == 1:
                return {'class': 1, 'addr': rd.uint(body, 2, rd.so), 'size': rd.uint(body, 2 + rd.so, rd.sl)}
            if cls == 2:
                nd = body[2]
                addr = rd.uint(body, 3, rd.so)
                dims = struct.unpack_from('<%dI' % nd, body, 3 + rd.so)
                return {'class': 2, 'addr': addr, 'chunk': tuple(dims[:-1])}
            raise NotImplementedError("HDF5 layout class %d" % cls)
        if ver in (1, 2):
            nd, cls = body[1], body[2]
            p = 8
            addr = UNDEF
            if cls != 0:
                addr = rd.uint(body, p, rd.so)
                p += rd.so
            dims = struct.unpack_from('<%dI' % nd, body, p)
            p += 4 * nd
            if cls == 0:
                n = struct.unpack_from('<I', body, p)[0]
                return {'class': 0, 'data': body[p + 4:p + 4 + n]}
            if cls == 1:
                return {'class': 1, 'addr': addr, 'size': None}
            return {'class': 2, 'addr': addr, 'chunk': tuple(dims[:-1])}
        raise NotImplementedError("HDF5 data layout message version %d (libver 'latest' files)" % ver)

    # ----------------------------------------------------------------- close
    def flush(self):
        if self._writable and self._dirty:
            tmp = self.filename + '.h5lite-tmp'
            with open(tmp, 'wb') as out:
                _Writer(out).write(self)
            if self._fh is not None:
                # everything still on disk was pulled into memory by the writer
                self._fh.close()
                self._fh = None
                self._reader = None
            os.replace(tmp, self.filename)
            self._dirty = False

    def close(self):
        if self._members is None and self._fh is None:
            return
        self.flush()
        if self._fh is not None:
            self._fh.close()
        self._fh = None
        self._reader = None
        self._members = None
        self._addr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __repr__(self):
        return '<h5lite file "%s" (mode %s)>' % (os.path.basename(self.filename), self.mode)


# ============================================================================ writing
def _pad8(b):
    return b + b'\0' * (-len(b) % 8)


def _msg(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack('<HHB3x', mtype, len(body), flags) + body


def _datatype_message(dt):
    if dt.kind in 'iu':
        bits = (0x08 if dt.kind == 'i' else 0) | (1 if dt.byteorder == '>' else 0)
        return struct.pack('<BBBBI', 0x10, bits, 0, 0, dt.itemsize) + struct.pack('<HH', 0, 8 * dt.itemsize)
    if dt.kind == 'f':
        spec = {2: (15, 10, 5, 0, 10, 15), 4: (31, 23, 8, 0, 23, 127), 8: (63, 52, 11, 0, 52, 1023)}
        if dt.itemsize not in spec:
            raise TypeError("float%d is not supported" % (8 * dt.itemsize))
        sign, eloc, esz, mloc, msz, bias = spec[dt.itemsize]
        b0 = 0x20 | (1 if dt.byteorder == '>' else 0)
        return struct.pack('<BBBBI', 0x11, b0, sign, 0, dt.itemsize) + \
            struct.pack('<HHBBBBI', 0, 8 * dt.itemsize, eloc, esz, mloc, msz, bias)
    if dt.kind == 'c':
        half = numpy.dtype('%sf%d' % ('>' if dt.byteorder == '>' else '<', dt.itemsize // 2))
        body = b''
        for k, name in enumerate((b'r', b'i')):
            body += _pad8(name + b'\0') + struct.pack('<I', k * half.itemsize) + b'\0' * 28 + _datatype_message(half)
        return struct.pack('<BBBBI', 0x16, 2, 0, 0, dt.itemsize) + body
    if dt.kind == 'S':
        # null-padded ASCII, the mapping h5py uses for numpy 'S' arrays
        return struct.pack('<BBBBI', 0x13, 0x01, 0, 0, max(dt.itemsize, 1))
    raise TypeError("dtype %s is not supported" % dt)


# variable-length UTF-8 string over a 1-byte base type, byte for byte what libhdf5 emits for h5py's str
_VLEN_STR = struct.pack('<BBBBI', 0x19, 0x01, 0x01, 0, 16) + struct.pack('<BBBBIHH', 0x10, 0, 0, 0, 1, 0, 8)


class _Writer(object):
    """Serialise a File tree: superblock 0, v1 headers, symbol-table groups, contiguous data."""
    LEAF_K = 4
    NODE_K = 16

    def __init__(self, out):
        self.out = out
        self.pos = 0

    def alloc(self, n):
        a = self.pos
        self.pos += n + (-n % 8)
        return a

    def put(self, addr, b):
        self.out.seek(addr)
        self.out.write(b)

    def write(self, root):
        self.alloc(96)                                  # superblock + root symbol table entry
        hdr, btree, heap = self.group(root)
        eof = self.pos
        sb = SIGNATURE + struct.pack('<BBBBBBBBHHI', 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, self.NODE_K, 0)
        sb += struct.pack('<QQQQ', 0, UNDEF, eof, UNDEF)
        sb += struct.pack('<QQII', 0, hdr, 1, 0) + struct.pack('<QQ', btree, heap)
        self.put(0, sb)
        self.out.seek(0, 2)
        if self.out.tell() < eof:
            self.out.write(b'\0' * (eof - self.out.tell()))

    # ---------------------------------------------------------------- groups
    def group(self, g):
        members = g._ensure()
        names = sorted(members.keys(), key=lambda s: s.encode('utf-8'))
        # local heap data segment: "" at offset 0, then the names, then one free block
        seg = bytearray(8)
        offs = {}
        for n in names:
            offs[n] = len(seg)
            seg += _pad8(n.encode('utf-8') + b'\0')
        free_off = len(seg)
        seg += struct.pack('<QQ', 1, 16)                # next = 1 (none), size of this free block
        hdr = self.alloc(16 + 24)
        heap = self.alloc(32)
        heap_data = self.alloc(len(seg))
        # children first (their header addresses go into the symbol nodes)
        entries = []
        for n in names:
            obj = members[n]
            if isinstance(obj, Group):
                h, b, hp = self.group(obj)
                entries.append(struct.pack('<QQII', offs[n], h, 1, 0) + struct.pack('<QQ', b, hp))
            else:
                h = self.dataset(obj)
                entries.append(struct.pack('<QQII', offs[n], h, 0, 0) + b'\0' * 16)
        # symbol nodes of <= 2*LEAF_K entries
        cap = 2 * self.LEAF_K
        nleaf = max(1, -(-len(names) // cap))
        per = -(-len(names) // nleaf) if names else 0
        level = []                                       # (address, heap offset of the largest name below)
        for i in range(nleaf):
            chunk = entries[i * per:(i + 1) * per]
            a = self.alloc(8 + cap * 40)
            self.put(a, b'SNOD' + struct.pack('<BBH', 1, 0, len(chunk)) + b''.join(chunk) +
                     b'\0' * (40 * (cap - len(chunk))))
            last = names[min((i + 1) * per, len(names)) - 1] if chunk else None
            level.append((a, offs[last] if last is not None else 0))
        if not names:
            level = []
        # B-tree levels above the symbol nodes
        depth = 0
        fan = 2 * self.NODE_K
        while True:
            nnode = max(1, -(-len(level) // fan))
            pern = -(-len(level) // nnode) if level else 0
            addrs = [self.alloc(24 + (2 * fan + 1) * 8) for _ in range(nnode)]
            nxt = []
            first_key = 0
            for i in range(nnode):
                kids = level[i * pern:(i + 1) * pern]
                body = struct.pack('<Q', first_key)
                for a, k in kids:
                    body += struct.pack('<QQ', a, k)
                left = addrs[i - 1] if i > 0 else UNDEF
                right = addrs[i + 1] if i + 1 < nnode else UNDEF
                node = b'TREE' + struct.pack('<BBHQQ', 0, depth, len(kids), left, right) + body
                self.put(addrs[i], node + b'\0' * (24 + (2 * fan + 1) * 8 - len(node)))
                last_key = kids[-1][1] if kids else 0
                nxt.append((addrs[i], last_key))
                first_key = last_key
            if nnode == 1:
                btree = addrs[0]
                break
            level = nxt
            depth += 1
        self.put(heap, b'HEAP' + struct.pack('<B3xQQQ', 0, len(seg), free_off, heap_data))
        self.put(heap_data, bytes(seg))
        body = _msg(0x11, struct.pack('<QQ', btree, heap))
        self.put(hdr, struct.pack('<BBHII4x', 1, 0, 1, 1, len(body)) + body)
        return hdr, btree, heap

    # -------------------------------------------------------------- datasets
    def dataset(self, d):
        data = d._load()
        if isinstance(data, (str, bytes)):
            raw = data.encode('utf-8') if isinstance(data, str) else data
            # one global heap collection holding the string, referenced by a scalar vlen element
            need = 16 + 16 + len(raw) + (-len(raw) % 8) + 16
            size = max(4096, need + (-need % 8))
            gcol = self.alloc(size)
            buf = b'GCOL' + struct.pack('<B3xQ', 1, size)
            buf += struct.pack('<HHIQ', 1, 0, 0, len(raw)) + _pad8(raw)
            buf += struct.pack('<HHIQ', 0, 0, 0, size - len(buf))          # free space object (size incl. header)
            self.put(gcol, buf + b'\0' * (size - len(buf)))
            elem = struct.pack('<IQI', len(raw), gcol, 1)
            return self._dataset_header((), _VLEN_STR, elem, fill_time=0)
        a = _storable(numpy.asarray(data))
        return self._dataset_header(a.shape, _datatype_message(a.dtype), a.tobytes() if a.size else b'',
                                    empty=(a.size == 0))

    def _dataset_header(self, shape, dtmsg, raw, empty=False, fill_time=2):
        space = struct.pack('<BBBB4x', 1, len(shape), 1 if shape else 0, 0)
        space += b''.join(struct.pack('<Q', s) for s in shape)
        space += b''.join(struct.pack('<Q', s) for s in shape)             # max dims = dims
        if not shape:
            space = struct.pack('<BBBB4x', 1, 0, 0, 0)
        data_addr = UNDEF
        if raw:
            data_addr = self.alloc(len(raw))
            self.put(data_addr, raw)
        msgs = _msg(0x01, space) + _msg(0x03, dtmsg, flags=1)
        msgs += _msg(0x05, struct.pack('<BBBBI', 2, 2, fill_time, 1, 0), flags=1)    # fill value: late alloc, if-set, default
        msgs += _msg(0x08, struct.pack('<BBQQ', 3, 1, data_addr, len(raw)))
        hdr = self.alloc(16 + len(msgs))
        self.put(hdr, struct.pack('<BBHII4x', 1, 0, 4, 1, len(msgs)) + msgs)
        return hdr
