"""h5lite — the product's own reader / writer for the HDF5 subset the reference's episode files use.

The reference stores episodes with h5py: `create_dataset(name, data=..., compression='lzf')`, one group per sensor folder
(VLA/data/franka_data/4_convert_to_hdf5.py:30-167; read back in residual_controller/controller_dataset.py:71-170 and
data/create_controller_dataset_episode.py:161-213).  h5py / libhdf5 are not available to the deployment interpreter, so
this module restates the published HDF5 file format (HDF5 File Format Specification v2/v3) for exactly what those files
contain, and is pinned to REAL h5py output by tests/golden/episodes_h5/*.h5 (written by h5py 3.3.0 / HDF5 1.10.6,
tools/make_h5_fixtures.py):
  read : superblock v0-v3, v1 object headers (+ continuation blocks), v2 object headers without creation-order / fractal
         heaps, old-style groups (symbol table: v1 B-tree + local heap + SNOD), datasets with contiguous / compact /
         chunked (v1 B-tree index, any depth) layouts, filter pipelines LZF (h5py's filter 32000), gzip (deflate), shuffle;
         fixed-point and IEEE float types of either byte order; simple attributes on v1 headers.
  write: superblock v0 + old-style groups + chunked LZF datasets, the layout `4_convert_to_hdf5.py` produces — validated by
         reading the result back with the real h5py in the build container (tests/test_h5lite.py).
LZF itself (Marc Lehmann's liblzf format, what h5py's filter wraps) is csrc/vt_lzf.c -> libvtlzf.so (ctypes) with a pure
Python statement of the decoder as the fallback.
API (h5py-like, read side):  with File(path) as f: f["ee_poses"][:], f["gelsight_force"]["forces"][10:20], "x" in f, f.keys()
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import zlib
from typing import Dict, Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
LZF_FILTER, GZIP_FILTER, SHUFFLE_FILTER = 32000, 1, 2

_lzf_lib = None


def _lzf():
    global _lzf_lib
    if _lzf_lib is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvtlzf.so")
        try:
            lib = C.CDLL(path)
            lib.vt_lzf_decompress.restype = C.c_long
            lib.vt_lzf_decompress.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long]
            lib.vt_lzf_compress.restype = C.c_long
            lib.vt_lzf_compress.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long]
            _lzf_lib = lib
        except OSError:
            _lzf_lib = False
    return _lzf_lib


def lzf_decompress_py(src: bytes, out_len: int) -> bytes:
    """liblzf stream: ctrl < 32 -> literal run of ctrl+1 bytes; else back reference: len = ctrl>>5 (7 -> + next byte) + 2,
    offset = ((ctrl & 31) << 8 | next byte) + 1 before the write position (may overlap)."""
    out = bytearray(out_len)
    ip, op, n = 0, 0, len(src)
    while ip < n:
        ctrl = src[ip]; ip += 1
        if ctrl < 32:
            ctrl += 1
            if op + ctrl > out_len or ip + ctrl > n:
                raise ValueError("lzf: literal run overflows")
            out[op:op + ctrl] = src[ip:ip + ctrl]
            ip += ctrl; op += ctrl
        else:
            ln = ctrl >> 5
            ref = op - ((ctrl & 0x1F) << 8) - 1
            if ip + (2 if ln == 7 else 1) > n:
                raise ValueError("lzf: truncated back reference")
            if ln == 7:
                ln += src[ip]; ip += 1
            ref -= src[ip]; ip += 1
            ln += 2
            if ref < 0 or op + ln > out_len:
                raise ValueError("lzf: bad back reference")
            if ref + ln <= op:
                out[op:op + ln] = out[ref:ref + ln]
            else:
                for k in range(ln):
                    out[op + k] = out[ref + k]
            op += ln
    if op != out_len:
        raise ValueError(f"lzf: decoded {op} bytes, expected {out_len}")
    return bytes(out)


def lzf_decompress(src: bytes, out_len: int) -> bytes:
    lib = _lzf()
    if not lib:
        return lzf_decompress_py(src, out_len)
    out = C.create_string_buffer(out_len)
    n = lib.vt_lzf_decompress(src, len(src), out, out_len)
    if n != out_len:
        raise ValueError(f"lzf: decoded {n} bytes, expected {out_len}")
    return out.raw


def lzf_compress(src: bytes) -> Optional[bytes]:
    """-> compressed bytes, or None when the data does not shrink (HDF5 then stores the chunk raw with the filter's mask bit set)."""
    lib = _lzf()
    if not lib or len(src) < 4:
        return None
    out = C.create_string_buffer(len(src))
    n = lib.vt_lzf_compress(src, len(src), out, len(src) - 1)
    return out.raw[:n] if n > 0 else None


# ===================================================================================== reading
class _Reader:
    def __init__(self, buf: bytes):
        self.b = buf
        base = 0
        while self.b[base:base + 8] != SIG:
            base = 512 if base == 0 else base * 2
            if base + 8 > len(self.b):
                raise ValueError("not an HDF5 file (signature not found)")
        self.base = base
        ver = self.b[base + 8]
        if ver in (0, 1):
            self.O, self.L = self.b[base + 13], self.b[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            p += 4 * self.O                                  # base address, free-space info, end of file, driver info
            self.root_header = self.addr(p + self.O)          # root symbol table entry: link name offset, object header address
        elif ver in (2, 3):
            self.O, self.L = self.b[base + 9], self.b[base + 10]
            self.root_header = self.addr(base + 12 + 3 * self.O)
        else:
            raise NotImplementedError(f"HDF5 superblock version {ver}")
        if self.O != 8 or self.L != 8:
            raise NotImplementedError("only 8-byte offsets / lengths")

    def addr(self, p: int) -> int:
        return struct.unpack_from("<Q", self.b, p)[0]

    def u(self, p: int, n: int) -> int:
        return int.from_bytes(self.b[p:p + n], "little")

    # ---- object headers -> list of (type, flags, payload offset, payload size)
    def messages(self, a: int) -> List[Tuple[int, int, int, int]]:
        a += self.base
        out: List[Tuple[int, int, int, int]] = []
        if self.b[a:a + 4] == b"OHDR":                      # version 2
            flags = self.b[a + 5]
            p = a + 6
            if flags & 0x20:
                p += 16                                      # four timestamps
            if flags & 0x10:
                p += 4                                       # max compact / min dense attribute counts
            szb = 1 << (flags & 3)
            chunk0 = self.u(p, szb); p += szb
            blocks = [(p, chunk0)]
            order = bool(flags & 0x04)
            while blocks:
                q, n = blocks.pop(0)
                end = q + n
                while q + 4 + (2 if order else 0) <= end:
                    mt, ms, mf = self.b[q], self.u(q + 1, 2), self.b[q + 3]
                    q += 4 + (2 if order else 0)
                    if mt == 0x10:
                        ca, cl = self.addr(q) + self.base, self.addr(q + 8)
                        blocks.append((ca + 4, cl - 8))      # "OCHK" signature ... 4-byte checksum
                    elif mt != 0:
                        out.append((mt, mf, q, ms))
                    q += ms
            return out
        ver = self.b[a]
        if ver != 1:
            raise NotImplementedError(f"object header version {ver}")
        nmsg, hsize = self.u(a + 2, 2), self.u(a + 8, 4)
        blocks = [(a + 16, hsize)]
        while blocks and len(out) < nmsg + 64:
            q, n = blocks.pop(0)
            end = q + n
            while q + 8 <= end:
                mt, ms, mf = self.u(q, 2), self.u(q + 2, 2), self.b[q + 4]
                q += 8
                if mt == 0x10:
                    blocks.append((self.addr(q) + self.base, self.addr(q + 8)))
                elif mt != 0:
                    out.append((mt, mf, q, ms))
                q += ms
        return out

    # ---- old-style group: symbol table message -> {name: object header address}
    def group_entries(self, btree: int, heap: int) -> Dict[str, int]:
        h = heap + self.base
        if self.b[h:h + 4] != b"HEAP":
            raise ValueError("bad local heap")
        data = self.addr(h + 24) + self.base
        out: Dict[str, int] = {}

        def name(off: int) -> str:
            e = self.b.index(b"\0", data + off)
            return self.b[data + off:e].decode("utf-8")

        def walk(node: int):
            n = node + self.base
            if self.b[n:n + 4] == b"SNOD":
                cnt = self.u(n + 6, 2)
                p = n + 8
                for _ in range(cnt):
                    out[name(self.addr(p))] = self.addr(p + 8)
                    p += 40
                return
            if self.b[n:n + 4] != b"TREE":
                raise ValueError("bad group B-tree node")
            used = self.u(n + 6, 2)
            p = n + 24 + 8                                   # skip key 0
            for _ in range(used):
                walk(self.addr(p))
                p += 16
        if btree != UNDEF:
            walk(btree)
        return out


def _parse_dtype(r: _Reader, p: int) -> np.dtype:
    cls, bits0, size = r.b[p] & 0x0F, r.b[p + 1], r.u(p + 4, 4)
    order = ">" if (bits0 & 1) else "<"
    if cls == 0:
        return np.dtype(f"{order}{'i' if bits0 & 0x08 else 'u'}{size}")
    if cls == 1:
        if size not in (2, 4, 8):
            raise NotImplementedError(f"float of {size} bytes")
        return np.dtype(f"{order}f{size}")
    raise NotImplementedError(f"HDF5 datatype class {cls} (only integers and IEEE floats are read)")


def _parse_space(r: _Reader, p: int) -> Tuple[int, ...]:
    ver, rank = r.b[p], r.b[p + 1]
    q = p + (8 if ver == 1 else 4)
    return tuple(r.u(q + 8 * i, 8) for i in range(rank))


def _parse_filters(r: _Reader, p: int) -> List[Tuple[int, List[int]]]:
    ver, n = r.b[p], r.b[p + 1]
    q = p + (8 if ver == 1 else 2)
    out = []
    for _ in range(n):
        fid = r.u(q, 2); q += 2
        nlen = 0
        if ver == 1 or fid >= 256:
            nlen = r.u(q, 2); q += 2
        q += 2                                               # flags
        ncd = r.u(q, 2); q += 2
        q += (nlen + 7) // 8 * 8 if ver == 1 else nlen
        cd = [r.u(q + 4 * i, 4) for i in range(ncd)]
        q += 4 * ncd
        if ver == 1 and ncd % 2:
            q += 4
        out.append((fid, cd))
    return out


class Dataset:
    def __init__(self, r: _Reader, name: str, msgs):
        self._r, self.name = r, name
        self.filters: List[Tuple[int, List[int]]] = []
        self.attrs: Dict[str, object] = {}
        self.chunks: Optional[Tuple[int, ...]] = None
        self._layout = None
        for mt, mf, p, ms in msgs:
            if mt == 0x01:
                self.shape = _parse_space(r, p)
            elif mt == 0x03:
                self.dtype = _parse_dtype(r, p)
            elif mt == 0x0B:
                self.filters = _parse_filters(r, p)
            elif mt == 0x08:
                ver, cls = r.b[p], r.b[p + 1]
                if ver != 3:
                    raise NotImplementedError(f"data layout message version {ver}")
                if cls == 0:
                    n = r.u(p + 2, 2)
                    self._layout = ("compact", p + 4, n)
                elif cls == 1:
                    self._layout = ("contiguous", r.addr(p + 2), r.addr(p + 10))
                else:
                    nd = r.b[p + 2]
                    bt = r.addr(p + 3)
                    dims = tuple(r.u(p + 11 + 4 * i, 4) for i in range(nd))
                    self.chunks = dims[:-1]
                    self._layout = ("chunked", bt, dims)
            elif mt == 0x0C:
                _read_attr(r, p, self.attrs)
        self.compression = {LZF_FILTER: "lzf", GZIP_FILTER: "gzip"}.get(next((f for f, _ in self.filters if f in (LZF_FILTER, GZIP_FILTER)), None))

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        return self.shape[0]

    def _unfilter(self, raw: bytes, mask: int, nbytes: int) -> bytes:
        for i in range(len(self.filters) - 1, -1, -1):       # the pipeline is undone last filter first
            if mask & (1 << i):
                continue
            fid, cd = self.filters[i]
            if fid == LZF_FILTER:
                raw = lzf_decompress(raw, nbytes)
            elif fid == GZIP_FILTER:
                d = zlib.decompressobj()
                raw = d.decompress(raw, nbytes)                # never inflate beyond the chunk the dataset declares
                if d.unconsumed_tail or not d.eof:
                    raise ValueError("gzip chunk larger than the dataset's chunk size")
            elif fid == SHUFFLE_FILTER:
                es = cd[0] if cd else self.dtype.itemsize
                a = np.frombuffer(raw, dtype=np.uint8)
                n = len(a) // es
                raw = a[:n * es].reshape(es, n).T.tobytes() + a[n * es:].tobytes()
            else:
                raise NotImplementedError(f"HDF5 filter {fid}")
        return raw

    def _chunk_index(self) -> List[Tuple[Tuple[int, ...], int, int, int]]:
        r = self._r
        _, bt, dims = self._layout
        nd = len(dims)
        out = []

        def walk(node: int):
            n = node + r.base
            if r.b[n:n + 4] != b"TREE" or r.b[n + 4] != 1:
                raise ValueError("bad chunk B-tree node")
            level, used = r.b[n + 5], r.u(n + 6, 2)
            p = n + 24
            ksz = 8 + 8 * nd
            for _ in range(used):
                size, mask = r.u(p, 4), r.u(p + 4, 4)
                off = tuple(r.u(p + 8 + 8 * i, 8) for i in range(nd - 1))
                child = r.addr(p + ksz)
                if level == 0:
                    out.append((off, child, size, mask))
                else:
                    walk(child)
                p += ksz + 8
        if bt != UNDEF:
            walk(bt)
        return out

    def read(self) -> np.ndarray:
        r = self._r
        kind = self._layout[0]
        n = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        if kind == "compact":
            _, p, nb = self._layout
            return np.frombuffer(r.b[p:p + nb], dtype=self.dtype, count=n).reshape(self.shape).copy()
        if kind == "contiguous":
            _, a, nb = self._layout
            if a == UNDEF:
                return np.zeros(self.shape, dtype=self.dtype)
            return np.frombuffer(r.b, dtype=self.dtype, count=n, offset=a + r.base).reshape(self.shape).copy()
        _, _, dims = self._layout
        cshape = dims[:-1]
        cbytes = int(np.prod(cshape, dtype=np.int64)) * dims[-1]
        out = np.zeros(self.shape, dtype=self.dtype)
        for off, a, size, mask in self._chunk_index():
            raw = self._unfilter(r.b[a + r.base:a + r.base + size], mask, cbytes)
            c = np.frombuffer(raw, dtype=self.dtype, count=cbytes // dims[-1]).reshape(cshape)
            sl = tuple(slice(o, min(o + cs, s)) for o, cs, s in zip(off, cshape, self.shape))
            out[sl] = c[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out

    def __getitem__(self, key):
        if not hasattr(self, "_cache"):
            self._cache = self.read()
        if key is Ellipsis or (isinstance(key, tuple) and len(key) == 0):
            return self._cache
        return self._cache[key]

    def __array__(self, dtype=None, copy=None):
        a = self[...]
        return a.astype(dtype) if dtype is not None else a


def _read_attr(r: _Reader, p: int, into: Dict[str, object]) -> None:
    """attribute message v1 (name, datatype, dataspace padded to 8) — integers / floats only; others are skipped."""
    try:
        ver = r.b[p]
        if ver != 1:
            return
        nl, tl, sl = r.u(p + 2, 2), r.u(p + 4, 2), r.u(p + 6, 2)
        q = p + 8
        name = r.b[q:q + nl].split(b"\0")[0].decode()
        q += (nl + 7) // 8 * 8
        dt = _parse_dtype(r, q)
        q += (tl + 7) // 8 * 8
        shape = _parse_space(r, q)
        q += (sl + 7) // 8 * 8
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        v = np.frombuffer(r.b, dtype=dt, count=n, offset=q).reshape(shape)
        into[name] = v.copy() if shape else v.reshape(()).item()
    except NotImplementedError:
        pass


class Group:
    def __init__(self, r: _Reader, name: str, header: int):
        self._r, self.name = r, name
        self.attrs: Dict[str, object] = {}
        self._entries: Dict[str, int] = {}
        for mt, mf, p, ms in r.messages(header):
            if mt == 0x11:
                self._entries = r.group_entries(r.addr(p), r.addr(p + 8))
            elif mt == 0x0C:
                _read_attr(r, p, self.attrs)
            elif mt in (0x02, 0x06):
                raise NotImplementedError("new-style (link message / dense) groups: write the file with h5py's default libver")

    def keys(self):
        return self._entries.keys()

    def __iter__(self) -> Iterator[str]:
        return iter(self._entries)

    def __len__(self):
        return len(self._entries)

    def __contains__(self, key: str) -> bool:
        try:
            self[key]
            return True
        except KeyError:
            return False

    def __getitem__(self, key: str) -> Union["Group", Dataset]:
        node: Union[Group, Dataset] = self
        for part in [k for k in key.split("/") if k]:
            if not isinstance(node, Group) or part not in node._entries:
                raise KeyError(key)
            hdr = node._entries[part]
            msgs = node._r.messages(hdr)
            path = f"{node.name.rstrip('/')}/{part}"
            if any(mt == 0x11 for mt, *_ in msgs) or not any(mt == 0x08 for mt, *_ in msgs):
                node = Group(node._r, path, hdr)
            else:
                node = Dataset(node._r, path, msgs)
        return node

    def items(self):
        return [(k, self[k]) for k in self._entries]


class File(Group):
    """Read-only HDF5 file (whole file mapped into memory: episode files are tens to hundreds of MB)."""

    def __init__(self, path: str, mode: str = "r"):
        if mode != "r":
            raise ValueError("h5lite.File is read-only; write episodes with h5lite.write_file")
        with open(path, "rb") as f:
            buf = f.read()
        r = _Reader(buf)
        super().__init__(r, "/", r.root_header)
        self.filename = path

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def close(self):
        pass


# ===================================================================================== writing
def _guess_chunk(shape: Tuple[int, ...], itemsize: int) -> Tuple[int, ...]:
    """h5py's auto-chunk heuristic in spirit (h5py/_hl/filters.py guess_chunk): halve the axes round-robin until a chunk is
    between 8 KiB and 1 MiB, sized by the dataset.  Any chunk shape is valid HDF5: readers take it from the file."""
    chunks = np.array([max(int(s), 1) for s in shape], dtype=np.float64)
    dset = np.prod(chunks) * itemsize
    target = min(max(8 * 1024 * 2 ** np.log10(max(dset, 1) / (1024.0 * 1024)), 8 * 1024), 1024 * 1024)
    i = 0
    while True:
        nb = np.prod(chunks) * itemsize
        if (nb < target or abs(nb - target) / target < 0.5) and nb < 1024 * 1024:
            break
        if np.prod(chunks) == 1:
            break
        chunks[i % len(chunks)] = np.ceil(chunks[i % len(chunks)] / 2.0)
        i += 1
    return tuple(int(c) for c in chunks)


class _Writer:
    def __init__(self):
        self.buf = bytearray()

    def align(self, n=8):
        self.buf += b"\0" * (-len(self.buf) % n)

    def put(self, b: bytes) -> int:
        self.align()
        a = len(self.buf)
        self.buf += b
        return a


def _dtype_msg(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    big = 1 if dt.byteorder == ">" else 0
    if dt.kind in "iu":
        bits = big | (0x08 if dt.kind == "i" else 0)
        return struct.pack("<BBBBI", 0x10 | 0, bits, 0, 0, dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        if dt.itemsize == 4:
            props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
            b1 = 31
        else:
            props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            b1 = 63
        return struct.pack("<BBBBI", 0x10 | 1, 0x20 | big, b1, 0, dt.itemsize) + props
    raise NotImplementedError(f"h5lite.write: dtype {dt}")


def _msg(mtype: int, payload: bytes, flags: int = 0) -> bytes:
    payload += b"\0" * (-len(payload) % 8)
    return struct.pack("<HHB3x", mtype, len(payload), flags) + payload


def _object_header(msgs: Sequence[bytes]) -> bytes:
    body = b"".join(msgs)
    return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body


def _write_dataset(w: _Writer, arr: np.ndarray, compression: Optional[str]) -> int:
    arr = np.asarray(arr)
    arr = arr if arr.ndim == 0 else np.ascontiguousarray(arr)     # ascontiguousarray would promote a scalar to shape (1,)
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    shape = arr.shape
    space = struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", s) for s in shape)
    msgs = [_msg(0x01, space), _msg(0x03, _dtype_msg(arr.dtype), 1)]
    fill = _msg(0x05, struct.pack("<BBBB", 2, 2 if compression else 1, 0, 0))
    if not compression or arr.ndim == 0 or arr.size == 0:
        a = w.put(arr.tobytes()) if arr.size else UNDEF
        msgs += [fill, _msg(0x08, struct.pack("<BB", 3, 1) + struct.pack("<QQ", a, arr.nbytes))]
        return w.put(_object_header(msgs))
    if compression != "lzf":
        raise NotImplementedError("h5lite.write: compression must be 'lzf' or None")
    chunk = _guess_chunk(shape, arr.dtype.itemsize)
    cbytes = int(np.prod(chunk)) * arr.dtype.itemsize
    grid = [range(0, s, c) for s, c in zip(shape, chunk)]
    entries = []
    for off in np.ndindex(*[len(g) for g in grid]):
        o = tuple(g[i] for g, i in zip(grid, off))
        block = np.zeros(chunk, dtype=arr.dtype)
        sl = tuple(slice(a, min(a + c, s)) for a, c, s in zip(o, chunk, shape))
        block[tuple(slice(0, s.stop - s.start) for s in sl)] = arr[sl]
        raw = block.tobytes()
        comp = lzf_compress(raw)
        mask = 0 if comp is not None else 1
        data = comp if comp is not None else raw
        entries.append((o, w.put(data), len(data), mask))
    nd = len(shape) + 1
    ksz = 8 + 8 * nd

    def key(size, mask, off):
        return struct.pack("<II", size, mask) + b"".join(struct.pack("<Q", x) for x in off) + struct.pack("<Q", 0)

    # chunk B-tree: nodes of up to 2K = 64 entries (K = 32, the library default for a v0 superblock), levels above as needed.
    # libhdf5 reads a node at its FULL size (24 + (2K+1) keys + 2K children), so every node is padded to it.
    K2 = 64
    node_bytes = 24 + (K2 + 1) * ksz + K2 * 8
    level0 = []
    for i in range(0, len(entries), K2):
        part = entries[i:i + K2]
        body = b"".join(key(sz, m, o) + struct.pack("<Q", a) for o, a, sz, m in part)
        last = tuple(x + c for x, c in zip(part[-1][0], chunk))
        body += key(0, 0, last)
        level0.append((part[0], part, body))
    addrs = []
    for idx, (first, part, body) in enumerate(level0):
        addrs.append(w.put((b"TREE" + struct.pack("<BBH", 1, 0, len(part)) + struct.pack("<QQ", UNDEF, UNDEF) + body).ljust(node_bytes, b"\0")))
    for idx, a in enumerate(addrs):                              # sibling links
        left = addrs[idx - 1] if idx > 0 else UNDEF
        right = addrs[idx + 1] if idx + 1 < len(addrs) else UNDEF
        w.buf[a + 8:a + 24] = struct.pack("<QQ", left, right)
    nodes = [(lv[0], a) for lv, a in zip(level0, addrs)]
    level = 0
    while len(nodes) > 1:
        level += 1
        nxt = []
        for i in range(0, len(nodes), K2):
            part = nodes[i:i + K2]
            body = b"".join(key(f[2], f[3], f[0]) + struct.pack("<Q", a) for f, a in part)
            body += key(0, 0, tuple(s + c for s, c in zip(shape, chunk)))
            a = w.put((b"TREE" + struct.pack("<BBH", 1, level, len(part)) + struct.pack("<QQ", UNDEF, UNDEF) + body).ljust(node_bytes, b"\0"))
            nxt.append((part[0][0], a))
        nodes = nxt
    root = nodes[0][1]
    name = b"lzf\0"
    pipeline = struct.pack("<BB6x", 1, 1) + struct.pack("<HHHH", LZF_FILTER, 8, 1, 3) + name.ljust(8, b"\0") + struct.pack("<IIII", 4, 0x0105, cbytes, 0)
    layout = struct.pack("<BBB", 3, 2, nd) + struct.pack("<Q", root) + b"".join(struct.pack("<I", c) for c in chunk) + struct.pack("<I", arr.dtype.itemsize)
    msgs += [fill, _msg(0x0B, pipeline, 1), _msg(0x08, layout)]
    return w.put(_object_header(msgs))


def _write_group(w: _Writer, tree: Dict[str, object], compression: Optional[str]) -> Tuple[int, int, int]:
    """-> (object header address, B-tree address, local heap address).  One SNOD per <= 64 children is enough for episode files."""
    children: List[Tuple[str, int]] = []
    for name in sorted(tree):
        v = tree[name]
        if isinstance(v, dict):
            children.append((name, _write_group(w, v, compression)[0]))
        else:
            children.append((name, _write_dataset(w, np.asarray(v), compression)))
    if len(children) > 2 * 32:
        raise NotImplementedError("h5lite.write: more than 64 members in one group")
    heap_data = bytearray(b"\0" * 8)
    offs = []
    for name, _ in children:
        offs.append(len(heap_data))
        heap_data += name.encode("utf-8") + b"\0"
        heap_data += b"\0" * (-len(heap_data) % 8)
    free_off = len(heap_data)
    heap_data += struct.pack("<QQ", 1, 16)                       # one free block: next = 1 (none), size 16
    data_addr = w.put(bytes(heap_data))
    heap = w.put(b"HEAP" + struct.pack("<B3x", 0) + struct.pack("<QQQ", len(heap_data), free_off, data_addr))
    snod = b"SNOD" + struct.pack("<BxH", 1, len(children))
    for (name, hdr), off in zip(children, offs):
        snod += struct.pack("<QQII16x", off, hdr, 0, 0)
    snod += b"\0" * (40 * (64 - len(children)))
    snod_addr = w.put(snod)
    last_key = offs[-1] if offs else 0
    bt = w.put((b"TREE" + struct.pack("<BBH", 0, 0, 1 if children else 0) + struct.pack("<QQ", UNDEF, UNDEF) +
                struct.pack("<QQQ", 0, snod_addr, last_key)).ljust(24 + 33 * 8 + 32 * 8, b"\0"))      # full node: group internal K = 16
    hdr = w.put(_object_header([_msg(0x11, struct.pack("<QQ", bt, heap))]))
    return hdr, bt, heap


def write_file(path: str, tree: Dict[str, object], compression: Optional[str] = "lzf") -> None:
    """tree: {name: array | {name: array ...}} -> an HDF5 file in the reference's episode layout: every array is a chunked
    dataset with h5py's LZF filter, every nested dict a group (4_convert_to_hdf5.py:30-167).  '/' in a key nests groups."""
    nested: Dict[str, object] = {}
    for k, v in tree.items():
        parts = [p for p in k.split("/") if p]
        d = nested
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = v
    w = _Writer()
    w.buf += b"\0" * 96                                          # superblock v0 (56 bytes + root symbol table entry 40 bytes)
    hdr, bt, heap = _write_group(w, nested, compression)
    w.align()
    eof = len(w.buf)
    sb = SIG + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", 32, 16, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, hdr, 1, 0) + struct.pack("<QQ", bt, heap)      # root entry, cache type 1: scratch = B-tree + heap
    w.buf[0:len(sb)] = sb
    with open(path, "wb") as f:
        f.write(bytes(w.buf))
