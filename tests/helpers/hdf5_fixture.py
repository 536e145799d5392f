"""Byte-level writer of small HDF5 files for the tests of synference_amd.hdf5_lite (test infrastructure only).

Neither h5py nor libhdf5 is available here, so the fixtures are assembled structure by structure from the HDF5 File
Format Specification 3.0 in the shape h5py's default settings give a Synference library
(ref: src/synference/library.py:4074-4153): superblock v0, symbol-table groups (v1 B-tree + local heap + symbol node),
v1 object headers, chunked datasets with a deflate (and optionally shuffle) pipeline indexed by a v1 chunk B-tree,
v1 attributes with variable-length UTF-8 strings in a global heap.  Reader and writer share one reading of the
specification: files written by real h5py remain the decisive check (it cannot be run in this image)."""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


def _pad8(b: bytes) -> bytes:
    return b + b"\x00" * (-len(b) % 8)


class Writer:
    def __init__(self):
        self.buf = bytearray(b"\x00" * 96)   # superblock v0 (8+16+32+40 = 96 bytes) is filled in at the end
        self.gheap = []                      # global heap objects (bytes)
        self.gheap_addr = None

    def alloc(self, data: bytes) -> int:
        self.buf += b"\x00" * (-len(self.buf) % 8)
        a = len(self.buf)
        self.buf += data
        return a

    # ---- datatypes / dataspaces --------------------------------------------------------------------------------
    @staticmethod
    def dt_float(size=8):
        if size == 8:
            return struct.pack("<BBBBI", 0x11, 0x20, 0x3F, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        return struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)

    @staticmethod
    def dt_vlen_str():
        base = struct.pack("<BBBBI", 0x13, 0x10, 0, 0, 1)           # string, null-terminated, UTF-8, size 1
        return struct.pack("<BBBBI", 0x19, 0x01, 0x01, 0, 16) + base  # vlen: type = string, charset UTF-8, size 16

    @staticmethod
    def dt_fixed_str(n):
        return struct.pack("<BBBBI", 0x13, 0x00, 0, 0, n)

    @staticmethod
    def dataspace(shape):
        return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", d) for d in shape)

    # ---- messages ------------------------------------------------------------------------------------------------
    @staticmethod
    def msg(typ, data):
        data = _pad8(data)
        return struct.pack("<HHB3x", typ, len(data), 0) + data

    def vlen_ref(self, s: str) -> bytes:
        b = s.encode("utf-8")
        self.gheap.append(b)
        return struct.pack("<IQI", len(b), 0xABABABABABABABAB, len(self.gheap))   # heap address patched at finish()

    def attr(self, name, value):
        nm = name.encode() + b"\x00"
        if isinstance(value, str):
            dt, sp, data = self.dt_vlen_str(), self.dataspace(()), self.vlen_ref(value)
        elif isinstance(value, (list, tuple)) and value and isinstance(value[0], str):
            dt, sp, data = self.dt_vlen_str(), self.dataspace((len(value),)), b"".join(self.vlen_ref(v) for v in value)
        elif isinstance(value, np.ndarray) and value.dtype.kind == "S":
            dt, sp, data = self.dt_fixed_str(value.dtype.itemsize), self.dataspace(value.shape), value.tobytes()
        else:
            v = np.asarray(value, dtype=np.float64)
            dt, sp, data = self.dt_float(8), self.dataspace(v.shape), v.tobytes()
        body = struct.pack("<BxHHH", 1, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp) + data
        return self.msg(0x000C, body)

    def object_header(self, msgs, split=False):
        """v1 object header; split=True puts the later messages into a continuation block (h5py does that when
        attributes are added after creation)."""
        if split and len(msgs) > 1:
            tail = b"".join(msgs[1:])
            cont_addr = self.alloc(tail)
            first = msgs[0] + self.msg(0x0010, struct.pack("<QQ", cont_addr, len(tail)))
            body, n = first, len(msgs) + 1
        else:
            body, n = b"".join(msgs), len(msgs)
        return self.alloc(struct.pack("<BxHII4x", 1, n, 1, len(body)) + body)

    # ---- datasets --------------------------------------------------------------------------------------------------
    def dataset(self, arr, chunks=None, gzip=4, shuffle=False, attrs=None):
        arr = np.ascontiguousarray(arr)
        dt = self.dt_float(arr.dtype.itemsize)
        msgs = [self.msg(0x0001, self.dataspace(arr.shape)), self.msg(0x0003, dt)]
        if chunks is None:
            addr = self.alloc(arr.tobytes())
            msgs.append(self.msg(0x0008, struct.pack("<BBQQ", 3, 1, addr, arr.nbytes)))
        else:
            rank, es = arr.ndim, arr.dtype.itemsize
            entries = []
            grid = [range(0, arr.shape[d], chunks[d]) for d in range(rank)]
            for offs in np.array(np.meshgrid(*grid, indexing="ij")).reshape(rank, -1).T:
                block = np.zeros(chunks, dtype=arr.dtype)
                sl = tuple(slice(int(o), min(int(o) + c, s)) for o, c, s in zip(offs, chunks, arr.shape))
                block[tuple(slice(0, s.stop - s.start) for s in sl)] = arr[sl]
                raw = block.tobytes()
                if shuffle:
                    a = np.frombuffer(raw, dtype=np.uint8)
                    raw = a.reshape(-1, es).T.tobytes()
                if gzip:
                    raw = zlib.compress(raw, gzip)
                entries.append((tuple(int(o) for o in offs), self.alloc(raw), len(raw)))
            ksz = 8 + 8 * (rank + 1)
            node = struct.pack("<4sBBHQQ", b"TREE", 1, 0, len(entries), UNDEF, UNDEF)
            for offs, addr, nb in entries:
                node += struct.pack("<II", nb, 0) + b"".join(struct.pack("<Q", o) for o in offs) + struct.pack("<Q", 0)
                node += struct.pack("<Q", addr)
            node += struct.pack("<II", 0, 0) + b"".join(struct.pack("<Q", s) for s in arr.shape) + struct.pack("<Q", 0)
            bt = self.alloc(node)
            msgs.append(self.msg(0x0008, struct.pack("<BBBQ", 3, 2, rank + 1, bt) +
                                 b"".join(struct.pack("<I", c) for c in chunks) + struct.pack("<I", es)))
            filt = b""
            nf = 0
            if shuffle:
                filt += struct.pack("<HHHH", 2, 0, 1, 1) + struct.pack("<II", es, 0)
                nf += 1
            if gzip:
                filt += struct.pack("<HHHH", 1, 8, 1, 1) + b"deflate\x00" + struct.pack("<II", gzip, 0)
                nf += 1
            if nf:
                msgs.append(self.msg(0x000B, struct.pack("<BB6x", 1, nf) + filt))
        for k, v in (attrs or {}).items():
            msgs.append(self.attr(k, v))
        return self.object_header(msgs)

    # ---- groups ------------------------------------------------------------------------------------------------------
    def group(self, links, attrs=None, split=False):
        """links: {name: object header address}"""
        names = sorted(links)
        heap = bytearray(b"\x00" * 8)
        offs = {}
        for n in names:
            offs[n] = len(heap)
            heap += _pad8(n.encode() + b"\x00")
        free_off = len(heap)
        heap += struct.pack("<QQ", 1, 16)       # a free block: next = 1 (none), size 16
        hdata = self.alloc(bytes(heap))
        haddr = self.alloc(struct.pack("<4sB3xQQQ", b"HEAP", 0, len(heap), free_off, hdata))
        snod = struct.pack("<4sBBH", b"SNOD", 1, 0, len(names))
        for n in names:
            snod += struct.pack("<QQII16x", offs[n], links[n], 0, 0)
        saddr = self.alloc(snod)
        tree = struct.pack("<4sBBHQQ", b"TREE", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, saddr, offs[names[-1]] if names else 0)
        taddr = self.alloc(tree)
        msgs = [self.msg(0x0011, struct.pack("<QQ", taddr, haddr))]
        for k, v in (attrs or {}).items():
            msgs.append(self.attr(k, v))
        return self.object_header(msgs, split=split), taddr, haddr

    def finish(self, root_links, root_attrs, path):
        root, taddr, haddr = self.group(root_links, root_attrs, split=True)
        # global heap collection with every variable-length string
        body = b""
        for i, b in enumerate(self.gheap):
            body += struct.pack("<HHIQ", i + 1, 1, 0, len(b)) + _pad8(b)
        size = max(4096, (16 + len(body) + 16 + 7) // 8 * 8)
        free = size - 16 - len(body)
        coll = struct.pack("<4sB3xQ", b"GCOL", 1, size) + body + struct.pack("<HHIQ", 0, 0, 0, free)
        coll += b"\x00" * (size - len(coll))
        gaddr = self.alloc(coll)
        self.buf = bytearray(bytes(self.buf).replace(struct.pack("<Q", 0xABABABABABABABAB), struct.pack("<Q", gaddr)))
        eof = len(self.buf)
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", taddr, haddr)
        assert len(sb) == 96
        self.buf[:96] = sb
        with open(path, "wb") as fh:
            fh.write(bytes(self.buf))


def write_library(path, photometry, parameters, filter_codes, parameter_names, parameter_units=None, supplementary=None,
                  supp_names=None, supp_units=None, chunks=None, gzip=4, shuffle=False):
    """A Synference library file: Grid/Photometry (C,N), Grid/Parameters (D,N) (+ supplementary), root attributes."""
    w = Writer()
    ch = lambda a: chunks if chunks is not None else (min(a.shape[0], 4), min(a.shape[1], 1024))
    links = {"Photometry": w.dataset(photometry, ch(photometry), gzip, shuffle),
             "Parameters": w.dataset(parameters, ch(parameters), gzip, shuffle)}
    if supplementary is not None:
        links["SupplementaryParameters"] = w.dataset(supplementary, ch(supplementary), gzip, shuffle)
    grid, _, _ = w.group(links)
    attrs = {"ParameterNames": list(parameter_names), "FilterCodes": list(filter_codes), "PhotometryUnits": "nJy"}
    if parameter_units is not None:
        attrs["ParameterUnits"] = list(parameter_units)
    if supplementary is not None:
        attrs["SupplementaryParameterNames"] = list(supp_names)
        attrs["SupplementaryParameterUnits"] = list(supp_units)
    attrs["CreationDT"] = "20260101_000000"
    w.finish({"Grid": grid}, attrs, path)
