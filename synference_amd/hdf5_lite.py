"""Minimal read-only HDF5 decoder for Synference library files.

The reference stores its model libraries with h5py (writer: ref: src/synference/library.py:4074-4153; reader:
ref: src/synference/utils.py:37-112): gzip-compressed 2-D float datasets ``Grid/Photometry`` (C, N),
``Grid/Parameters`` (D, N), optionally ``Grid/SupplementaryParameters`` / ``Grid/Spectra``, and root attributes
``FilterCodes``, ``ParameterNames``, ``ParameterUnits``, ``PhotometryUnits`` ... (lists of strings).  Neither
libhdf5 nor h5py exists in the build image, so this module decodes the on-disk structures that h5py's default
settings (``libver='earliest'``) produce, straight from the HDF5 File Format Specification (version 3.0):

    superblock v0 (also v2/v3)        II.A      object header v1 (+ continuation blocks)   IV.A.1.a
    symbol-table groups               III.A-D   v1 B-trees, local heaps, symbol nodes
    dataspace v1/v2                   IV.A.2.b  datatype: fixed-point, float, string, vlen  IV.A.2.d
    data layout v3                    IV.A.2.i  contiguous / compact / chunked (v1 chunk B-tree, III.A.1)
    filter pipeline v1/v2             IV.A.2.l  deflate (zlib), shuffle, fletcher32 (checksum skipped)
    attribute v1/v2/v3                IV.A.2.m  global heap for variable-length strings    III.E

Not handled (an ``Hdf5Error`` says which): v2 object headers / link-message groups (libver='latest'), v2 B-tree
chunk indices, compound / array / reference types, external storage, szip / other filters.
Pure Python + numpy + zlib; a 1e6-row library decodes at zlib speed (the chunks are inflated one by one).
"""
from __future__ import annotations

import os
import struct
import zlib
from typing import Dict, List, Optional, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(ValueError):
    pass


class _Datatype:
    def __init__(self, cls, size, np_dtype=None, vlen_string=False, str_pad=None, base=None):
        self.cls, self.size, self.np_dtype, self.vlen_string, self.str_pad, self.base = cls, size, np_dtype, vlen_string, str_pad, base


class Dataset:
    def __init__(self, f: "File", name: str, msgs):
        self.file, self.name = f, name
        self.shape: Tuple[int, ...] = ()
        self.dtype: Optional[_Datatype] = None
        self.layout = None
        self.filters: List[Tuple[int, List[int]]] = []
        self.attrs: Dict[str, object] = {}
        for typ, data in msgs:
            if typ == 0x0001:
                self.shape = f._dataspace(data)
            elif typ == 0x0003:
                self.dtype = f._datatype(data)[0]
            elif typ == 0x0008:
                self.layout = f._layout(data)
            elif typ == 0x000B:
                self.filters = f._filters(data)
            elif typ == 0x000C:
                k, v = f._attribute(data)
                self.attrs[k] = v

    def __getitem__(self, key):
        arr = self.read()
        return arr[key]

    def read(self, out: "np.ndarray | None" = None, workers: "int | None" = None) -> np.ndarray:
        """The whole dataset.  ``out``: a C-contiguous array of the dataset's shape and dtype to fill instead of a fresh
        one (e.g. the numpy view of a pinned torch tensor, so that the host-to-device copy that follows is one DMA);
        ``workers``: threads that inflate the chunks of a chunked dataset (default: the usable cores, at most 16)."""
        f = self.file
        dt = self.dtype
        if dt is None or self.layout is None:
            raise Hdf5Error(f"{self.name}: not a dataset")
        n = int(np.prod(self.shape)) if self.shape else 1
        if dt.np_dtype is None:
            raise Hdf5Error(f"{self.name}: datatype class {dt.cls} is not supported for datasets")
        kind = self.layout[0]
        if kind == "compact":
            raw = self.layout[1]
        elif kind == "contiguous":
            addr, size = self.layout[1], self.layout[2]
            raw = b"\x00" * (n * dt.size) if addr == UNDEF else f._read(addr, size)
        else:
            return self._read_chunked(out, workers)
        arr = np.frombuffer(raw[: n * dt.size], dtype=dt.np_dtype).reshape(self.shape)
        if out is not None:
            self._check_out(out)
            out[...] = arr
            return out
        return arr.copy()

    def _check_out(self, out):
        if tuple(out.shape) != tuple(self.shape) or out.dtype != self.dtype.np_dtype.newbyteorder("=") and out.dtype != self.dtype.np_dtype:
            raise Hdf5Error(f"{self.name}: out must have shape {self.shape} and dtype {self.dtype.np_dtype}")

    def _read_chunked(self, out=None, workers=None) -> np.ndarray:
        f, dt = self.file, self.dtype
        btree, cdims = self.layout[1], self.layout[2]          # cdims: chunk shape (without the element-size entry)
        rank = len(self.shape)
        if out is None:
            out = np.zeros(self.shape, dtype=dt.np_dtype)
        else:
            self._check_out(out)
            out[...] = 0
        if btree == UNDEF:
            return out
        csize = int(np.prod(cdims)) * dt.size
        chunks = list(f._chunk_btree(btree, rank))
        # Chunks are independent: inflate them on a thread pool (zlib and the numpy byte shuffles release the GIL; file
        # reads go through os.pread, which needs no shared file position) and let every task write its own slice of `out`.
        if workers is None:
            try:
                workers = len(os.sched_getaffinity(0))
            except AttributeError:
                workers = os.cpu_count() or 1
            workers = max(1, min(16, workers))
        if workers > 1 and len(chunks) >= 4 * workers:
            from concurrent.futures import ThreadPoolExecutor
            fd = f._fh.fileno()

            def task(part):
                for c in part:
                    self._decode_chunk(os.pread(fd, c[2], f._base + c[1]), c, out, csize, cdims, rank)
            step = max(1, len(chunks) // (8 * workers))
            with ThreadPoolExecutor(workers) as ex:
                list(ex.map(task, [chunks[i:i + step] for i in range(0, len(chunks), step)]))
            return out
        for c in chunks:
            self._decode_chunk(f._read(c[1], c[2]), c, out, csize, cdims, rank)
        return out

    def _decode_chunk(self, raw, c, out, csize, cdims, rank):
        dt = self.dtype
        offs, addr, nbytes, mask = c
        if True:
            for i in range(len(self.filters) - 1, -1, -1):     # undo the pipeline in reverse order
                if mask & (1 << i):
                    continue
                fid, cd = self.filters[i]
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    es = cd[0] if cd else dt.size
                    a = np.frombuffer(raw, dtype=np.uint8)
                    m = len(a) // es
                    raw = a[: m * es].reshape(es, m).T.tobytes() + a[m * es:].tobytes()
                elif fid == 3:
                    raw = raw[:-4]                               # fletcher32 checksum (not verified)
                else:
                    raise Hdf5Error(f"{self.name}: filter id {fid} is not supported")
            chunk = np.frombuffer(raw[:csize], dtype=dt.np_dtype).reshape(cdims)
            sl_out, sl_in = [], []
            for d in range(rank):
                hi = min(offs[d] + cdims[d], self.shape[d])
                sl_out.append(slice(offs[d], hi))
                sl_in.append(slice(0, hi - offs[d]))
            out[tuple(sl_out)] = chunk[tuple(sl_in)]


class Group:
    def __init__(self, f: "File", name: str, msgs):
        self.file, self.name = f, name
        self.attrs: Dict[str, object] = {}
        self._links: Dict[str, int] = {}
        for typ, data in msgs:
            if typ == 0x0011:
                btree, heap = struct.unpack_from("<QQ", data, 0)
                self._links.update(f._symbol_table(btree, heap))
            elif typ == 0x000C:
                k, v = f._attribute(data)
                self.attrs[k] = v
            elif typ in (0x0002, 0x0006):
                raise Hdf5Error(f"{name or '/'}: link-message groups (libver='latest') are not supported; write the file with "
                                "h5py's default libver")

    def keys(self):
        return list(self._links)

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def __getitem__(self, path: str):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(path)
            node = node.file._object(node._links[part], (node.name + "/" + part).lstrip("/"))
        return node


class File(Group):
    """``with File(path) as f: f["Grid/Parameters"][:]; f.attrs["ParameterNames"]`` -- the subset of h5py's reading
    API that the reference's ``load_library_from_hdf5`` uses."""

    def __init__(self, path: str):
        self._fh = open(path, "rb")
        self._cache: Dict[int, object] = {}
        self._base = 0
        head = self._read(0, 8)
        base = 0
        while head != SIGNATURE:                                   # the superblock may sit at 512, 1024, 2048, ...
            base = 512 if base == 0 else base * 2
            head = self._read(base, 8)
            if base > (1 << 24) or len(head) < 8:
                raise Hdf5Error(f"{path}: no HDF5 signature")
        ver = self._read(base + 8, 1)[0]
        if ver in (0, 1):
            so, sl = struct.unpack_from("<BB", self._read(base + 13, 2))
            if (so, sl) != (8, 8):
                raise Hdf5Error("only 8-byte offsets / lengths are supported")
            p = base + 24 + (4 if ver == 1 else 0)                 # base address, free-space, EOF, driver: 4 x 8 bytes
            base_addr = struct.unpack_from("<Q", self._read(p, 8))[0]
            ste = self._read(p + 32, 40)                           # root group symbol table entry
            root_addr = struct.unpack_from("<Q", ste, 8)[0]
            self._base = base_addr
        elif ver in (2, 3):
            so, sl = struct.unpack_from("<BB", self._read(base + 9, 2))
            if (so, sl) != (8, 8):
                raise Hdf5Error("only 8-byte offsets / lengths are supported")
            base_addr, _ext, _eof, root_addr = struct.unpack_from("<QQQQ", self._read(base + 12, 32))
            self._base = base_addr
        else:
            raise Hdf5Error(f"superblock version {ver} is not supported")
        root = self._object(root_addr, "")
        if not isinstance(root, Group):
            raise Hdf5Error("the root object is not a group")
        Group.__init__(self, self, "", [])
        self.attrs, self._links = root.attrs, root._links

    # ---- context manager ---------------------------------------------------------------------------------------
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        self._fh.close()

    # ---- low level ----------------------------------------------------------------------------------------------
    def _read(self, addr: int, n: int) -> bytes:
        self._fh.seek(self._base + addr)                           # file addresses are relative to the base address
        return self._fh.read(n)

    def _object(self, addr: int, name: str):
        if addr in self._cache:
            return self._cache[addr]
        msgs = self._object_header(addr)
        types = {t for t, _ in msgs}
        obj = Dataset(self, name, msgs) if 0x0008 in types else Group(self, name, msgs)
        self._cache[addr] = obj
        return obj

    def _object_header(self, addr: int):
        head = self._read(addr, 16)
        if head[:4] == b"OHDR":
            raise Hdf5Error("version 2 object headers (libver='latest') are not supported; write the file with h5py's default libver")
        ver, _, nmsg, _ref, hsize = struct.unpack_from("<BBHII", head, 0)
        if ver != 1:
            raise Hdf5Error(f"object header version {ver} at {addr} is not supported")
        msgs = []
        blocks = [(addr + 16, hsize)]
        while blocks and len(msgs) < nmsg + 64:
            a, size = blocks.pop(0)
            buf = self._read(a, size)
            p = 0
            while p + 8 <= len(buf):
                typ, msize, _flags = struct.unpack_from("<HHB", buf, p)
                data = buf[p + 8: p + 8 + msize]
                p += 8 + msize
                if typ == 0x0010:                                  # continuation
                    blocks.append(struct.unpack_from("<QQ", data, 0))
                elif typ != 0:
                    msgs.append((typ, data))
        return msgs

    # ---- groups ---------------------------------------------------------------------------------------------------
    def _heap_string(self, heap_data: bytes, off: int) -> str:
        end = heap_data.index(b"\x00", off)
        return heap_data[off:end].decode("utf-8")

    def _symbol_table(self, btree: int, heap: int) -> Dict[str, int]:
        h = self._read(heap, 32)
        if h[:4] != b"HEAP":
            raise Hdf5Error("bad local heap signature")
        dsize, _free, daddr = struct.unpack_from("<QQQ", h, 8)
        hdata = self._read(daddr, dsize)
        out: Dict[str, int] = {}

        def walk(addr):
            n = self._read(addr, 24)
            if n[:4] == b"TREE":
                ntype, level, used = struct.unpack_from("<BBH", n, 4)
                if ntype != 0:
                    raise Hdf5Error("group B-tree expected")
                body = self._read(addr + 24, (2 * used + 1) * 8)
                for i in range(used):
                    walk(struct.unpack_from("<Q", body, (2 * i + 1) * 8)[0])
            elif n[:4] == b"SNOD":
                nsym = struct.unpack_from("<H", n, 6)[0]
                ent = self._read(addr + 8, nsym * 40)
                for i in range(nsym):
                    name_off, ohdr = struct.unpack_from("<QQ", ent, i * 40)
                    out[self._heap_string(hdata, name_off)] = ohdr
            else:
                raise Hdf5Error("bad group node signature")
        walk(btree)
        return out

    # ---- messages -------------------------------------------------------------------------------------------------
    def _dataspace(self, d: bytes) -> Tuple[int, ...]:
        ver, rank, flags = struct.unpack_from("<BBB", d, 0)
        if ver == 1:
            p = 8
        elif ver == 2:
            p = 4
            if d[3] == 2:                                          # null dataspace
                return (0,)
        else:
            raise Hdf5Error(f"dataspace version {ver}")
        return tuple(struct.unpack_from("<" + "Q" * rank, d, p)) if rank else ()

    def _datatype(self, d: bytes) -> Tuple[_Datatype, int]:
        cv, b0, b1, _b2, size = struct.unpack_from("<BBBBI", d, 0)
        cls = cv & 0x0F
        if cls == 0:                                               # fixed point
            dt = np.dtype(("<" if not (b0 & 1) else ">") + ("i" if b0 & 8 else "u") + str(size))
            return _Datatype(0, size, dt), 8 + 4
        if cls == 1:                                               # floating point (IEEE assumed)
            dt = np.dtype(("<" if not (b0 & 1) else ">") + "f" + str(size))
            return _Datatype(1, size, dt), 8 + 12
        if cls == 3:                                               # fixed-length string
            return _Datatype(3, size, np.dtype(f"S{size}"), str_pad=b0 & 0x0F), 8
        if cls == 9:                                               # variable length
            base, used = self._datatype(d[8:])
            return _Datatype(9, size, None, vlen_string=(b0 & 0x0F) == 1, base=base), 8 + used
        raise Hdf5Error(f"datatype class {cls} is not supported")

    def _layout(self, d: bytes):
        ver, cls = struct.unpack_from("<BB", d, 0)
        if ver != 3:
            raise Hdf5Error(f"data layout version {ver} is not supported (written with a newer libver?)")
        if cls == 0:
            size = struct.unpack_from("<H", d, 2)[0]
            return ("compact", d[4:4 + size])
        if cls == 1:
            addr, size = struct.unpack_from("<QQ", d, 2)
            return ("contiguous", addr, size)
        if cls == 2:
            ndim = d[2]
            addr = struct.unpack_from("<Q", d, 3)[0]
            dims = struct.unpack_from("<" + "I" * ndim, d, 11)
            return ("chunked", addr, tuple(dims[:-1]))
        raise Hdf5Error(f"layout class {cls}")

    def _filters(self, d: bytes):
        ver, n = struct.unpack_from("<BB", d, 0)
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = struct.unpack_from("<H", d, p)[0]
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = struct.unpack_from("<H", d, p)[0]
                p += 2
            _flags, ncd = struct.unpack_from("<HH", d, p)
            p += 4
            if nlen:
                p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            cd = list(struct.unpack_from("<" + "I" * ncd, d, p))
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, cd))
        return out

    def _global_heap_object(self, addr: int, index: int) -> bytes:
        head = self._read(addr, 16)
        if head[:4] != b"GCOL":
            raise Hdf5Error("bad global heap signature")
        size = struct.unpack_from("<Q", head, 8)[0]
        buf = self._read(addr, size)
        p = 16
        while p + 16 <= size:
            idx, _ref, _res, osize = struct.unpack_from("<HHIQ", buf, p)
            if idx == 0:
                break
            if idx == index:
                return buf[p + 16: p + 16 + osize]
            p += 16 + (osize + 7) // 8 * 8
        raise Hdf5Error(f"global heap object {index} not found")

    def _decode_values(self, dt: _Datatype, shape: Tuple[int, ...], raw: bytes):
        n = int(np.prod(shape)) if shape else 1
        if dt.cls == 9:
            if not dt.vlen_string:
                raise Hdf5Error("variable-length sequences are not supported")
            vals = []
            for i in range(n):
                length, gaddr, gidx = struct.unpack_from("<IQI", raw, i * 16)
                vals.append(self._global_heap_object(gaddr, gidx)[:length].decode("utf-8") if length else "")
            return vals[0] if not shape else np.array(vals, dtype=object).reshape(shape)
        if dt.cls == 3:
            arr = np.frombuffer(raw[: n * dt.size], dtype=dt.np_dtype)
            vals = [v.split(b"\x00")[0].decode("utf-8") for v in arr.tolist()]
            return vals[0] if not shape else np.array(vals, dtype=object).reshape(shape)
        arr = np.frombuffer(raw[: n * dt.size], dtype=dt.np_dtype)
        return arr[0] if not shape else arr.reshape(shape).copy()

    def _attribute(self, d: bytes):
        ver = d[0]
        if ver == 1:
            nsz, tsz, ssz = struct.unpack_from("<HHH", d, 2)
            p = 8
            pad = lambda v: (v + 7) // 8 * 8
        elif ver in (2, 3):
            nsz, tsz, ssz = struct.unpack_from("<HHH", d, 2)
            p = 8 + (1 if ver == 3 else 0)
            pad = lambda v: v
        else:
            raise Hdf5Error(f"attribute version {ver}")
        name = d[p:p + nsz].split(b"\x00")[0].decode("utf-8")
        p += pad(nsz)
        dt, _ = self._datatype(d[p:p + tsz])
        p += pad(tsz)
        shape = self._dataspace(d[p:p + ssz])
        p += pad(ssz)
        return name, self._decode_values(dt, shape, d[p:])

    # ---- chunk index ------------------------------------------------------------------------------------------------
    def _chunk_btree(self, addr: int, rank: int):
        n = self._read(addr, 24)
        if n[:4] != b"TREE":
            raise Hdf5Error("bad chunk B-tree signature")
        ntype, level, used = struct.unpack_from("<BBH", n, 4)
        if ntype != 1:
            raise Hdf5Error("raw-data chunk B-tree expected")
        ksize = 8 + 8 * (rank + 1)
        body = self._read(addr + 24, used * (ksize + 8) + ksize)
        for i in range(used):
            p = i * (ksize + 8)
            nbytes, mask = struct.unpack_from("<II", body, p)
            offs = struct.unpack_from("<" + "Q" * (rank + 1), body, p + 8)
            child = struct.unpack_from("<Q", body, p + ksize)[0]
            if level > 0:
                yield from self._chunk_btree(child, rank)
            else:
                yield offs[:rank], child, nbytes, mask
