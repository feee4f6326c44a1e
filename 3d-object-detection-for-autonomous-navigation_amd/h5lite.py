"""Read-only HDF5 access for Keras checkpoints, in plain Python + numpy (no h5py, no libhdf5).

The reference keeps its checkpoints as Keras `save_weights` files (`model_weights_<epoch>.h5`, train.py:407,436) and
reads them back with `net.load_weights` (train.py:731-734).  h5py is not part of this image (nor of every deployment
box), so the importer (`weights.from_keras_h5`) would be unusable exactly where it is needed.  This module reads the
subset of the HDF5 file format such files -- and ordinary h5py / PyTables files -- are made of, and shows it through
the slice of the h5py interface the importer uses:

    with h5lite.File(path) as f:
        f.attrs["layer_names"]            # numpy array of bytes
        g = f["rpn"]; g.attrs["weight_names"]; np.asarray(g["rpn/block1/.../kernel:0"])

Supported (HDF5 File Format Specification, versions 0-3 of the superblock):
  * superblock versions 0, 1, 2, 3 (with a user block in front);
  * object headers version 1 and version 2 (`OHDR` / `OCHK`), continuation blocks;
  * groups: symbol tables (B-tree v1 + local heap, what libver="earliest" writes) and compact link messages;
  * datasets: contiguous, compact and chunked (B-tree v1 chunk index) layouts, with the deflate and shuffle filters;
  * datatypes: fixed-point, IEEE floating point (either byte order), fixed-length strings, variable-length strings
    (global heap); scalar, simple and null dataspaces;
  * attributes (message versions 1, 2, 3) of those types.
Anything else (dense / fractal-heap groups and attributes, compound / enum / array / reference types, other filters,
version-4 chunk indexes, external or soft links) raises `Unsupported` with the name of the feature: a wrong guess is
never returned.  `tests/test_h5lite.py` checks it against files written by the real library (h5py 3.3.0 / libhdf5
1.10.6, `tools/gen_golden_h5.py`).
"""
import mmap
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"


class Unsupported(NotImplementedError):
    """The file uses a part of the format this reader does not implement."""


class FormatError(ValueError):
    """The bytes are not what the format specification says they should be."""


def _guarded(fn):
    """A truncated or damaged file shows up as a read past the end or a nonsense length: report it as FormatError."""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **k):
        try:
            return fn(*a, **k)
        except (FormatError, Unsupported, KeyError):
            raise
        except (struct.error, IndexError, OverflowError, MemoryError, RecursionError, zlib.error, UnicodeDecodeError,
                ValueError, TypeError) as ex:
            raise FormatError(f"damaged or truncated HDF5 file ({type(ex).__name__}: {ex})") from ex
    return wrapper


# ------------------------------------------------------------------------------------------------------------
# datatypes
# ------------------------------------------------------------------------------------------------------------
class _Type:
    """dtype: numpy dtype of one element as stored; vlen_str: elements are global-heap references to strings."""

    def __init__(self, dtype, vlen_str=False, size=None):
        self.dtype = dtype
        self.vlen_str = vlen_str
        self.size = size if size is not None else dtype.itemsize


def _parse_datatype(buf, off, offsize):
    cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", buf, off)
    cls, ver = cv & 0x0F, cv >> 4
    if ver not in (1, 2, 3):
        raise Unsupported(f"datatype message version {ver}")
    if cls == 0:                                   # fixed-point
        order = ">" if (b0 & 1) else "<"
        signed = bool(b0 & 0x08)
        if size not in (1, 2, 4, 8):
            raise Unsupported(f"{size}-byte integers")
        return _Type(np.dtype(f"{order}{'i' if signed else 'u'}{size}"))
    if cls == 1:                                   # floating point
        if b0 & 0x40:
            raise Unsupported("VAX byte order")
        order = ">" if (b0 & 1) else "<"
        if size not in (2, 4, 8):
            raise Unsupported(f"{size}-byte floating point")
        bit_off, prec, eloc, esize, mloc, msize, ebias = struct.unpack_from("<HHBBBBI", buf, off + 8)
        ieee = {2: (10, 5, 0, 10, 15), 4: (23, 8, 0, 23, 127), 8: (52, 11, 0, 52, 1023)}[size]
        if (eloc, esize, mloc, msize, ebias) != ieee or bit_off != 0 or prec != 8 * size:
            raise Unsupported("non-IEEE floating-point layout")
        return _Type(np.dtype(f"{order}f{size}"))
    if cls == 3:                                   # fixed-length string
        return _Type(np.dtype(f"S{size}"))
    if cls == 9:                                   # variable length
        if (b0 & 0x0F) != 1:
            raise Unsupported("variable-length sequences")
        return _Type(np.dtype(f"V{size}"), vlen_str=True, size=size)
    names = {2: "time", 4: "bitfield", 5: "opaque", 6: "compound", 7: "reference", 8: "enum", 10: "array"}
    raise Unsupported(f"{names.get(cls, cls)} datatype")


def _parse_dataspace(buf, off, lensize):
    ver, rank, flags = struct.unpack_from("<BBB", buf, off)
    if ver == 1:
        p = off + 8
    elif ver == 2:
        kind = buf[off + 3]
        if kind == 2:                              # null dataspace
            return None
        p = off + 4
    else:
        raise Unsupported(f"dataspace message version {ver}")
    fmt = {4: "<I", 8: "<Q", 2: "<H"}[lensize]
    return tuple(struct.unpack_from(fmt, buf, p + i * lensize)[0] for i in range(rank))


# ------------------------------------------------------------------------------------------------------------
# file
# ------------------------------------------------------------------------------------------------------------
class _Msg:
    __slots__ = ("type", "off", "size", "flags")

    def __init__(self, type_, off, size, flags):
        self.type, self.off, self.size, self.flags = type_, off, size, flags


class AttributeManager(dict):
    """attrs mapping (h5py spells `.get`, `in`, `[]`, iteration: a dict does)."""


class _Node:
    def __init__(self, file, addr, name):
        self._f, self._addr, self.name = file, addr, name
        self._msgs = file._object_messages(addr)
        self._attrs = None

    @property
    @_guarded
    def attrs(self):
        if self._attrs is None:
            a = AttributeManager()
            for m in self._msgs:
                if m.type == 0x000C:
                    k, v = self._f._parse_attribute(m)
                    a[k] = v
                elif m.type == 0x0015:
                    ver, flags = self._f._buf[m.off], self._f._buf[m.off + 1]
                    p = m.off + 2 + (2 if flags & 1 else 0)
                    heap = self._f._addr_at(p)
                    if heap is not None:
                        raise Unsupported("dense attribute storage (fractal heap)")
            self._attrs = a
        return self._attrs


class Group(_Node):
    def __init__(self, file, addr, name):
        super().__init__(file, addr, name)
        self._links = None

    @_guarded
    def _children(self):
        if self._links is None:
            f, links = self._f, {}
            for m in self._msgs:
                if m.type == 0x0011:               # symbol table: B-tree v1 + local heap
                    btree, heap = f._addr_at(m.off), f._addr_at(m.off + f._O)
                    f._walk_group_btree(btree, f._local_heap_data(heap), links)
                elif m.type == 0x0006:             # link message (compact new-style group)
                    k, v = f._parse_link(m)
                    links[k] = v
                elif m.type == 0x0002:             # link info: dense storage when the heap address is defined
                    flags = f._buf[m.off + 1]
                    p = m.off + 2 + (8 if flags & 1 else 0)
                    if f._addr_at(p) is not None:
                        raise Unsupported("dense link storage (fractal heap)")
            self._links = links
        return self._links

    def keys(self):
        return list(self._children().keys())

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self._children())

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    @_guarded
    def __getitem__(self, path):
        if isinstance(path, bytes):
            path = path.decode()
        node = self._f if path.startswith("/") else self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            kids = node._children()
            if part not in kids:
                raise KeyError(f"{path!r}: no member {part!r} in {node.name!r}")
            child_name = (node.name.rstrip("/") + "/" + part)
            node = node._f._open(kids[part], child_name)
        return node

    def visit_datasets(self):
        """[(path, Dataset)] of every dataset below this group, depth first in name order."""
        out = []
        for k in sorted(self.keys()):
            n = self[k]
            if isinstance(n, Group):
                out.extend(n.visit_datasets())
            else:
                out.append((n.name, n))
        return out


class Dataset(_Node):
    def __init__(self, file, addr, name):
        super().__init__(file, addr, name)
        f = file
        self._type = self._layout = None
        self.shape = ()
        self._filters = []
        for m in self._msgs:
            if m.type == 0x0003:
                try:
                    self._type = _parse_datatype(f._buf, m.off, f._O)
                except Unsupported as ex:          # the object can be listed; reading it raises
                    self._type = ex
            elif m.type == 0x0001:
                self.shape = _parse_dataspace(f._buf, m.off, f._L)
            elif m.type == 0x0008:
                self._layout = m
            elif m.type == 0x000B:
                self._filters = f._parse_filters(m)
        if self._type is None or self._layout is None:
            raise FormatError(f"{name}: dataset without a datatype or a layout message")

    @property
    def dtype(self):
        if isinstance(self._type, Exception):
            raise self._type
        if self._type.vlen_str:
            return np.dtype(object)
        t = self._type.dtype
        return t.newbyteorder("=") if t.kind in "iuf" else t

    @_guarded
    def _raw(self):
        f, m, t = self._f, self._layout, self._type
        if isinstance(t, Exception):
            raise t
        if self.shape is None:
            return np.zeros((0,), t.dtype)
        count = int(np.prod(self.shape, dtype=np.int64)) if len(self.shape) else 1
        nbytes = count * t.size
        buf = f._buf
        if nbytes > max(len(buf) * 1024, 1 << 20):     # (deflate cannot expand by more than ~1000x)
            raise FormatError(f"{self.name}: a dataspace of {nbytes} bytes in a file of {len(buf)}")
        ver = buf[m.off]
        if ver in (3, 4):
            cls = buf[m.off + 1]
            if ver == 4 and cls == 2:
                raise Unsupported("version-4 chunk indexes (libver='latest')")
            if cls == 0:
                size = struct.unpack_from("<H", buf, m.off + 2)[0]
                data = bytes(buf[m.off + 4:m.off + 4 + size])
            elif cls == 1:
                addr = f._addr_at(m.off + 2)
                data = b"\0" * nbytes if addr is None else bytes(buf[addr:addr + nbytes])   # never written: fill value 0
            elif cls == 2:
                ndim = buf[m.off + 2]
                btree = f._addr_at(m.off + 3)
                dims = struct.unpack_from(f"<{ndim}I", buf, m.off + 3 + f._O)
                data = f._read_chunked(btree, self.shape, dims[:-1], t.size, self._filters)
            else:
                raise Unsupported(f"data layout class {cls}")
        elif ver in (1, 2):
            ndim, cls = buf[m.off + 1], buf[m.off + 2]
            p = m.off + 8
            addr = None
            if cls != 0:
                addr = f._addr_at(p)
                p += f._O
            dims = struct.unpack_from(f"<{ndim}I", buf, p)
            p += 4 * ndim
            if cls == 0:
                size = struct.unpack_from("<I", buf, p)[0]
                data = bytes(buf[p + 4:p + 4 + size])
            elif cls == 1:
                data = b"\0" * nbytes if addr is None else bytes(buf[addr:addr + nbytes])
            elif cls == 2:
                data = f._read_chunked(addr, self.shape, dims[:-1], t.size, self._filters)
            else:
                raise Unsupported(f"data layout class {cls}")
        else:
            raise Unsupported(f"data layout message version {ver}")
        if len(data) < nbytes:
            raise FormatError(f"{self.name}: {len(data)} bytes stored, {nbytes} expected")
        return f._decode_elements(data[:nbytes], t, self.shape)

    def __array__(self, dtype=None, copy=None):
        a = self._raw()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, key):
        return self._raw()[key]


class File(Group):
    def __init__(self, path, mode="r"):
        if mode != "r":
            raise Unsupported("h5lite reads only")
        self._fh = open(path, "rb")
        try:
            self._buf = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:                         # empty file
            self._fh.close()
            raise FormatError(f"{path}: empty file")
        self.filename = path
        self._cache = {}
        try:
            root = self._read_superblock()
            super().__init__(self, root, "/")
        except Exception:
            self.close()
            raise

    # -- context manager ------------------------------------------------------------------------------------
    def close(self):
        if self._buf is not None:
            try:
                self._buf.close()
            except (BufferError, ValueError):
                pass
            self._buf = None
        if self._fh is not None:
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # -- low level ------------------------------------------------------------------------------------------
    def _uint(self, off, size):
        return int.from_bytes(self._buf[off:off + size], "little")

    def _addr_at(self, off):
        """Absolute file offset stored at `off`, or None for the undefined address (all ones)."""
        v = self._uint(off, self._O)
        return None if v == (1 << (8 * self._O)) - 1 else v + self._base

    @_guarded
    def _read_superblock(self):
        buf, n = self._buf, len(self._buf)
        start = 0
        while True:
            if start + 8 > n:
                raise FormatError("no HDF5 signature")
            if buf[start:start + 8] == SIGNATURE:
                break
            start = 512 if start == 0 else start * 2
        ver = buf[start + 8]
        self._base = 0
        if ver in (0, 1):
            self._O, self._L = buf[start + 13], buf[start + 14]
            p = start + 24 + (4 if ver == 1 else 0)
            base = self._uint(p, self._O)
            self._base = base
            p += 4 * self._O                       # base, free-space info, end of file, driver info
            root = self._addr_at(p + self._O)      # symbol table entry: link name offset, object header address
        elif ver in (2, 3):
            self._O, self._L = buf[start + 9], buf[start + 10]
            p = start + 12
            self._base = self._uint(p, self._O)
            root = self._addr_at(p + 3 * self._O)
        else:
            raise Unsupported(f"superblock version {ver}")
        if self._O not in (4, 8) or self._L not in (4, 8):
            raise Unsupported(f"{self._O}-byte offsets / {self._L}-byte lengths")
        if root is None:
            raise FormatError("no root group")
        return root

    @_guarded
    def _object_messages(self, addr):
        buf = self._buf
        msgs = []
        if buf[addr:addr + 4] == b"OHDR":          # version 2
            if buf[addr + 4] != 2:
                raise Unsupported(f"object header version {buf[addr + 4]}")
            flags = buf[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szsz = 1 << (flags & 3)
            size0 = self._uint(p, szsz)
            p += szsz
            blocks = [(p, size0)]
            tracked = bool(flags & 0x04)
            i = 0
            while i < len(blocks):
                if i > 4096:
                    raise FormatError("object header continuation loop")
                q, size = blocks[i]
                end = q + size
                i += 1
                while q + 4 <= end:
                    mtype = buf[q]
                    msize, mflags = struct.unpack_from("<HB", buf, q + 1)
                    q += 4 + (2 if tracked else 0)
                    if mtype == 0x10:
                        coff, clen = self._addr_at(q), self._uint(q + self._O, self._L)
                        if buf[coff:coff + 4] != b"OCHK":
                            raise FormatError("continuation block without OCHK")
                        blocks.append((coff + 4, clen - 8))      # signature in front, checksum behind
                    elif mtype != 0:
                        msgs.append(_Msg(mtype, q, msize, mflags))
                    q += msize
            return msgs
        ver = buf[addr]
        if ver != 1:
            raise FormatError(f"object header at {addr}: version {ver}")
        nmsgs = struct.unpack_from("<H", buf, addr + 2)[0]
        hsize = struct.unpack_from("<I", buf, addr + 8)[0]
        blocks = [(addr + 16, hsize)]
        i = seen = 0
        while i < len(blocks):
            if i > 4096:
                raise FormatError("object header continuation loop")
            q, size = blocks[i]
            end = q + size
            i += 1
            while q + 8 <= end and seen < nmsgs:
                mtype, msize, mflags = struct.unpack_from("<HHB", buf, q)
                q += 8
                seen += 1
                if mtype == 0x0010:
                    blocks.append((self._addr_at(q), self._uint(q + self._O, self._L)))
                elif mtype != 0:
                    if mflags & 0x02:
                        raise Unsupported("shared object header messages")
                    msgs.append(_Msg(mtype, q, msize, mflags))
                q += msize
        return msgs

    def _open(self, addr, name):
        node = self._cache.get(addr)
        if node is None:
            msgs = self._object_messages(addr)
            is_dataset = any(m.type == 0x0008 for m in msgs)
            node = (Dataset if is_dataset else Group)(self, addr, name)
            self._cache[addr] = node
        return node

    # -- groups ---------------------------------------------------------------------------------------------
    def _local_heap_data(self, addr):
        buf = self._buf
        if buf[addr:addr + 4] != b"HEAP":
            raise FormatError("local heap signature")
        return self._addr_at(addr + 8 + 2 * self._L)

    def _heap_string(self, heap_data, off):
        buf = self._buf
        end = buf.find(b"\0", heap_data + off)
        return bytes(buf[heap_data + off:end]).decode("utf-8")

    def _walk_group_btree(self, addr, heap_data, links):
        buf = self._buf
        if addr is None:
            return
        if buf[addr:addr + 4] != b"TREE":
            raise FormatError("B-tree signature")
        ntype, level, used = struct.unpack_from("<BBH", buf, addr + 4)
        if ntype != 0:
            raise FormatError("group B-tree of the wrong node type")
        p = addr + 8 + 2 * self._O + self._L        # behind the siblings and key 0
        for _ in range(used):
            child = self._addr_at(p)
            p += self._O + self._L
            if level > 0:
                self._walk_group_btree(child, heap_data, links)
                continue
            if buf[child:child + 4] != b"SNOD":
                raise FormatError("symbol table node signature")
            nsym = struct.unpack_from("<H", buf, child + 6)[0]
            q = child + 8
            for _ in range(nsym):
                name_off = self._uint(q, self._O)
                obj = self._addr_at(q + self._O)
                cache_type = struct.unpack_from("<I", buf, q + 2 * self._O)[0]
                if cache_type == 2:
                    raise Unsupported("symbolic links")
                links[self._heap_string(heap_data, name_off)] = obj
                q += 2 * self._O + 24

    def _parse_link(self, m):
        buf = self._buf
        ver, flags = buf[m.off], buf[m.off + 1]
        if ver != 1:
            raise Unsupported(f"link message version {ver}")
        p = m.off + 2
        ltype = 0
        if flags & 0x08:
            ltype = buf[p]
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        lsz = 1 << (flags & 3)
        nlen = self._uint(p, lsz)
        p += lsz
        name = bytes(buf[p:p + nlen]).decode("utf-8")
        p += nlen
        if ltype != 0:
            raise Unsupported("soft / external links")
        return name, self._addr_at(p)

    # -- attributes and element decoding ------------------------------------------------------------------------
    def _parse_attribute(self, m):
        buf = self._buf
        ver = buf[m.off]
        if m.flags & 0x02:
            raise Unsupported("shared attribute messages")
        nsz, tsz, ssz = struct.unpack_from("<HHH", buf, m.off + 2)
        if ver == 1:
            pad = lambda n: (n + 7) & ~7
            p = m.off + 8
        elif ver in (2, 3):
            if buf[m.off + 1] & 0x03:
                raise Unsupported("shared datatype / dataspace in an attribute")
            pad = lambda n: n
            p = m.off + 8 + (1 if ver == 3 else 0)
        else:
            raise Unsupported(f"attribute message version {ver}")
        name = bytes(buf[p:p + nsz]).split(b"\0")[0].decode("utf-8")
        p += pad(nsz)
        t = _parse_datatype(buf, p, self._O)
        p += pad(tsz)
        shape = _parse_dataspace(buf, p, self._L)
        p += pad(ssz)
        if shape is None:
            return name, np.zeros((0,), t.dtype)
        count = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
        data = bytes(buf[p:p + count * t.size])
        v = self._decode_elements(data, t, shape)
        if v.shape == ():
            v = v[()]
            if isinstance(v, np.bytes_):
                v = bytes(v)
        return name, v

    def _decode_elements(self, data, t, shape):
        if not t.vlen_str:
            a = np.frombuffer(data, dtype=t.dtype).reshape(shape)
            if t.dtype.kind in "iuf" and t.dtype.byteorder == ">":
                return a.astype(t.dtype.newbyteorder("="))
            return a.copy()                        # (frombuffer gives a read-only view of the bytes)
        out = np.empty(int(np.prod(shape, dtype=np.int64)) if len(shape) else 1, dtype=object)
        step = t.size                              # 4-byte length + global heap id (address + 4-byte index)
        for i in range(out.size):
            rec = data[i * step:(i + 1) * step]
            length = struct.unpack_from("<I", rec, 0)[0]
            coll = int.from_bytes(rec[4:4 + self._O], "little")
            index = struct.unpack_from("<I", rec, 4 + self._O)[0]
            if coll == 0 or coll == (1 << (8 * self._O)) - 1:
                out[i] = ""
            else:
                out[i] = self._global_heap_object(coll + self._base, index)[:length].decode("utf-8")
        return out.reshape(shape)

    def _global_heap_object(self, coll, index):
        buf = self._buf
        if buf[coll:coll + 4] != b"GCOL":
            raise FormatError("global heap collection signature")
        size = self._uint(coll + 8, self._L)
        p, end = coll + 8 + self._L, coll + size
        while p + 8 + self._L <= end:
            idx = struct.unpack_from("<H", buf, p)[0]
            osize = self._uint(p + 8, self._L)
            if idx == 0:
                break
            if idx == index:
                return bytes(buf[p + 8 + self._L:p + 8 + self._L + osize])
            p += 8 + self._L + ((osize + 7) & ~7)
        raise FormatError(f"global heap object {index} not found")

    # -- chunked storage ------------------------------------------------------------------------------------
    def _parse_filters(self, m):
        buf = self._buf
        ver, n = buf[m.off], buf[m.off + 1]
        p = m.off + (8 if ver == 1 else 2)
        out = []
        for _ in range(n):
            fid = struct.unpack_from("<H", buf, p)[0]
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = struct.unpack_from("<H", buf, p)[0]
                p += 2
            flags, ncd = struct.unpack_from("<HH", buf, p)
            p += 4
            if ver == 1:
                nlen = (nlen + 7) & ~7
            p += nlen
            cd = struct.unpack_from(f"<{ncd}I", buf, p)
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, cd))
        return out

    def _read_chunked(self, btree, shape, chunk, elsize, filters):
        for fid, _ in filters:
            if fid not in (1, 2, 3):
                raise Unsupported(f"filter {fid} (only deflate, shuffle and fletcher32 are read)")
        rank = len(shape)
        out = np.zeros(shape, dtype=f"V{elsize}")
        if btree is None:
            return out.tobytes()
        chunk_bytes = int(np.prod(chunk, dtype=np.int64)) * elsize
        for offs, size, mask, addr in self._walk_chunk_btree(btree, rank):
            raw = bytes(self._buf[addr:addr + size])
            for k in range(len(filters) - 1, -1, -1):          # the pipeline is undone back to front
                if mask & (1 << k):
                    continue
                fid, cd = filters[k]
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    es = cd[0] if cd else elsize
                    n = len(raw) // es
                    raw = np.frombuffer(raw[:n * es], np.uint8).reshape(es, n).T.tobytes() + raw[n * es:]
                elif fid == 3:
                    raw = raw[:-4]
            if len(raw) < chunk_bytes:
                raise FormatError("short chunk")
            block = np.frombuffer(raw[:chunk_bytes], dtype=f"V{elsize}").reshape(chunk)
            sel_out, sel_in = [], []
            for d in range(rank):
                lo = offs[d]
                hi = min(lo + chunk[d], shape[d])
                sel_out.append(slice(lo, hi))
                sel_in.append(slice(0, hi - lo))
            out[tuple(sel_out)] = block[tuple(sel_in)]
        return out.tobytes()

    def _walk_chunk_btree(self, addr, rank):
        buf = self._buf
        if buf[addr:addr + 4] != b"TREE":
            raise FormatError("B-tree signature")
        ntype, level, used = struct.unpack_from("<BBH", buf, addr + 4)
        if ntype != 1:
            raise FormatError("chunk B-tree of the wrong node type")
        keysize = 8 + 8 * (rank + 1)
        p = addr + 8 + 2 * self._O
        for _ in range(used):
            size, mask = struct.unpack_from("<II", buf, p)
            offs = struct.unpack_from(f"<{rank}Q", buf, p + 8)
            child = self._addr_at(p + keysize)
            p += keysize + self._O
            if level > 0:
                yield from self._walk_chunk_btree(child, rank)
            else:
                yield offs, size, mask, child


def is_hdf5(path):
    try:
        with open(path, "rb") as fh:
            head = fh.read(8)
        return head == SIGNATURE
    except OSError:
        return False
