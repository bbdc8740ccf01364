"""h5lite - a minimal, read-only HDF5 reader (pure Python + NumPy + zlib; no h5py, no libhdf5).

Why: the reference reads its two file formats through h5py - the Albacore fast5 reads
(nanorev_fast5_handeler.py:58-133: `Events` compound table, gzip-chunked int16 `Signal`,
`Fastq` scalar string, `version` attribute) and the Keras `save_weights` files
(NanoReviser_train.py:175-176: contiguous f32 datasets + `layer_names` / `weight_names`
string-array attributes).  h5py is not available where the engine runs, so this module
implements exactly the subset of the HDF5 file format those files use:

  * superblock version 0/1, 8-byte offsets/lengths
  * "old style" groups: symbol-table message -> v1 B-tree (TREE) -> SNOD nodes -> local HEAP
  * version-1 object headers with continuation blocks
  * dataspace v1/v2; datatypes: fixed-point, IEEE float, fixed strings, compound (v1-v3),
    variable-length strings (global heap) for attributes
  * data layout v3 (and v1/v2): compact, contiguous, chunked (v1 B-tree) with the deflate and
    shuffle filters
  * attribute messages v1-v3

It follows the published "HDF5 File Format Specification Version 2.0/3.0" (The HDF Group),
not h5py's or the reference's code.
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, List, Optional, Tuple

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(RuntimeError):
    pass


class _Buf:
    def __init__(self, data: bytes):
        self.d = data

    def u(self, off: int, n: int) -> int:
        return int.from_bytes(self.d[off:off + n], "little")

    def bytes(self, off: int, n: int) -> bytes:
        return self.d[off:off + n]


def _pad8(n: int) -> int:
    return (n + 7) & ~7


# ------------------------------------------------------------------------------------ datatypes
class _Type:
    """Parsed datatype message: numpy dtype + (optional) vlen-string flag."""

    def __init__(self, dtype: np.dtype, size: int, vlen_str: bool = False):
        self.dtype, self.size, self.vlen_str = dtype, size, vlen_str


def _parse_datatype(b: _Buf, off: int) -> Tuple[_Type, int]:
    """Returns (type, bytes consumed)."""
    cv = b.u(off, 1)
    cls, ver = cv & 0x0F, cv >> 4
    bits = b.u(off + 1, 3)
    size = b.u(off + 4, 4)
    p = off + 8
    if cls == 0:                                    # fixed-point
        signed = bool(bits & 0x08)
        order = ">" if bits & 1 else "<"
        p += 4
        return _Type(np.dtype(f"{order}{'i' if signed else 'u'}{size}"), size), p - off
    if cls == 1:                                    # floating point
        order = ">" if bits & 1 else "<"
        p += 12
        return _Type(np.dtype(f"{order}f{size}"), size), p - off
    if cls == 3:                                    # fixed-length string
        return _Type(np.dtype(f"S{size}"), size), p - off
    if cls == 6:                                    # compound
        nmemb = bits & 0xFFFF
        names, formats, offsets = [], [], []
        for _ in range(nmemb):
            e = b.d.index(b"\x00", p)
            name = b.d[p:e].decode("ascii")
            if ver < 3:
                p += _pad8(e - p + 1)
            else:
                p = e + 1
            if ver == 1:
                moff = b.u(p, 4)
                p += 4 + 1 + 3 + 4 + 4 + 16           # offset, rank, reserved, perm, reserved, dims
            elif ver == 2:
                moff = b.u(p, 4)
                p += 4
            else:
                nb = max(1, (max(size, 1).bit_length() + 7) // 8)
                moff = b.u(p, nb)
                p += nb
            mt, used = _parse_datatype(b, p)
            p += used
            names.append(name)
            formats.append(mt.dtype)
            offsets.append(moff)
        return _Type(np.dtype({"names": names, "formats": formats, "offsets": offsets,
                               "itemsize": size}), size), p - off
    if cls == 9:                                    # variable length
        is_str = (bits & 0x0F) == 1
        base, used = _parse_datatype(b, p)
        p += used
        if not is_str:
            raise H5Error("variable-length sequences are not supported")
        return _Type(np.dtype("O"), size, vlen_str=True), p - off
    raise H5Error(f"datatype class {cls} not supported")


def _parse_dataspace(b: _Buf, off: int) -> Tuple[int, ...]:
    ver, rank, flags = b.u(off, 1), b.u(off + 1, 1), b.u(off + 2, 1)
    if ver == 1:
        p = off + 8
    elif ver == 2:
        if b.u(off + 3, 1) == 2:                    # null dataspace
            return (0,)
        p = off + 4
    else:
        raise H5Error(f"dataspace version {ver}")
    return tuple(b.u(p + 8 * i, 8) for i in range(rank))


# ------------------------------------------------------------------------------------ objects
class _Obj:
    """An object header: its messages, lazily interpreted as a group or a dataset."""

    def __init__(self, f: "File", addr: int):
        self.f, self.addr = f, addr
        self.msgs: List[Tuple[int, int, int]] = []          # (type, data offset, size)
        self._read_header()
        self._attrs: Optional[Dict[str, object]] = None

    def _read_header(self):
        b = self.f.b
        if b.bytes(self.addr, 4) == b"OHDR":
            raise H5Error("version-2 object headers (new-style files) are not supported")
        ver = b.u(self.addr, 1)
        if ver != 1:
            raise H5Error(f"object header version {ver}")
        nmsg = b.u(self.addr + 2, 2)
        hsize = b.u(self.addr + 8, 4)
        blocks = [(self.addr + 16, hsize)]
        while blocks and len(self.msgs) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(self.msgs) < nmsg:
                mtype, msize = b.u(p, 2), b.u(p + 2, 2)
                data = p + 8
                if mtype == 0x10:                       # continuation
                    blocks.append((b.u(data, 8), b.u(data + 8, 8)))
                self.msgs.append((mtype, data, msize))
                p = data + msize

    def _find(self, mtype: int) -> List[Tuple[int, int]]:
        return [(o, s) for t, o, s in self.msgs if t == mtype]

    # ---- attributes ---------------------------------------------------------------------------
    @property
    def attrs(self) -> Dict[str, object]:
        if self._attrs is None:
            self._attrs = {}
            for off, size in self._find(0x0C):
                name, val = self._parse_attr(off)
                self._attrs[name] = val
        return self._attrs

    def _parse_attr(self, off: int):
        b = self.f.b
        ver = b.u(off, 1)
        nsz, tsz, ssz = b.u(off + 2, 2), b.u(off + 4, 2), b.u(off + 6, 2)
        p = off + 8
        if ver == 3:
            p += 1
        pad = _pad8 if ver == 1 else (lambda n: n)
        name = b.bytes(p, nsz).split(b"\x00")[0].decode("utf8")
        p += pad(nsz)
        typ, _ = _parse_datatype(b, p)
        p += pad(tsz)
        shape = _parse_dataspace(b, p) if ssz >= 4 and b.u(p + 1, 1) > 0 else ()
        p += pad(ssz)
        n = int(np.prod(shape)) if shape else 1
        return name, self.f._decode(typ, shape, b.bytes(p, n * typ.size), scalar=not shape)

    # ---- group -----------------------------------------------------------------------------------
    def is_group(self) -> bool:
        return bool(self._find(0x11))

    def links(self) -> Dict[str, int]:
        st = self._find(0x11)
        if not st:
            if self._find(0x02) or self._find(0x06):
                raise H5Error("new-style (link message) groups are not supported")
            raise H5Error("not a group")
        b = self.f.b
        btree, heap = b.u(st[0][0], 8), b.u(st[0][0] + 8, 8)
        if b.bytes(heap, 4) != b"HEAP":
            raise H5Error("bad local heap")
        hdata = b.u(heap + 24, 8)
        out: Dict[str, int] = {}

        def walk(addr):
            if b.bytes(addr, 4) == b"TREE":
                level, used = b.u(addr + 5, 1), b.u(addr + 6, 2)
                p = addr + 24
                for i in range(used):
                    child = b.u(p + 8, 8)                # key (8), child (8), ...
                    walk(child)
                    p += 16
            elif b.bytes(addr, 4) == b"SNOD":
                nsym = b.u(addr + 6, 2)
                p = addr + 8
                for i in range(nsym):
                    noff, oaddr = b.u(p, 8), b.u(p + 8, 8)
                    e = b.d.index(b"\x00", hdata + noff)
                    out[b.d[hdata + noff:e].decode("utf8")] = oaddr
                    p += 40
            else:
                raise H5Error("bad group B-tree node")
        walk(btree)
        return out

    # ---- dataset ---------------------------------------------------------------------------------
    def read(self) -> np.ndarray:
        b = self.f.b
        ds, dt, lay = self._find(0x01), self._find(0x03), self._find(0x08)
        if not (ds and dt and lay):
            raise H5Error("not a dataset")
        shape = _parse_dataspace(b, ds[0][0])
        scalar = b.u(ds[0][0] + 1, 1) == 0
        typ, _ = _parse_datatype(b, dt[0][0])
        n = int(np.prod(shape)) if not scalar else 1
        off = lay[0][0]
        ver = b.u(off, 1)
        if ver == 3:
            cls = b.u(off + 1, 1)
            if cls == 0:
                sz = b.u(off + 2, 2)
                raw = b.bytes(off + 4, sz)
            elif cls == 1:
                addr, sz = b.u(off + 2, 8), b.u(off + 10, 8)
                raw = b"" if addr == UNDEF else b.bytes(addr, sz)
            elif cls == 2:
                rank = b.u(off + 2, 1)
                btree = b.u(off + 3, 8)
                cdims = [b.u(off + 11 + 4 * i, 4) for i in range(rank)]
                raw = self._read_chunked(btree, shape, cdims[:-1], typ.size)
            else:
                raise H5Error(f"layout class {cls}")
        elif ver in (1, 2):
            rank, cls = b.u(off + 1, 1), b.u(off + 2, 1)
            p = off + 8
            addr = None
            if cls != 0:
                addr = b.u(p, 8)
                p += 8
            dims = [b.u(p + 4 * i, 4) for i in range(rank)]
            p += 4 * rank
            if cls == 1:
                raw = b.bytes(addr, n * typ.size)
            elif cls == 2:
                raw = self._read_chunked(addr, shape, dims[:-1], typ.size)
            else:
                sz = b.u(p, 4)
                raw = b.bytes(p + 4, sz)
        else:
            raise H5Error(f"data layout version {ver}")
        return self.f._decode(typ, shape, raw[:n * typ.size] if not typ.vlen_str else raw, scalar=scalar)

    def _filters(self) -> List[Tuple[int, List[int]]]:
        b = self.f.b
        out = []
        for off, _ in self._find(0x0B):
            ver, nf = b.u(off, 1), b.u(off + 1, 1)
            p = off + (8 if ver == 1 else 2)
            for _ in range(nf):
                fid = b.u(p, 2)
                if ver == 1 or fid >= 256:
                    nlen = b.u(p + 2, 2)
                    flags, ncv = b.u(p + 4, 2), b.u(p + 6, 2)
                    p += 8 + (_pad8(nlen) if ver == 1 else nlen)
                else:
                    flags, ncv = b.u(p + 2, 2), b.u(p + 4, 2)
                    p += 6
                cv = [b.u(p + 4 * i, 4) for i in range(ncv)]
                p += 4 * ncv
                if ver == 1 and ncv % 2:
                    p += 4
                out.append((fid, cv))
        return out

    def _read_chunked(self, btree: int, shape, cdims, esize: int) -> bytes:
        b = self.f.b
        rank = len(shape)
        filters = self._filters()
        arr = np.zeros(shape, dtype=f"V{esize}")
        chunk_elems = int(np.prod(cdims))

        def walk(addr):
            if addr == UNDEF:
                return
            if b.bytes(addr, 4) != b"TREE" or b.u(addr + 4, 1) != 1:
                raise H5Error("bad chunk B-tree node")
            level, used = b.u(addr + 5, 1), b.u(addr + 6, 2)
            ksz = 8 + 8 * (rank + 1)
            p = addr + 24
            for _ in range(used):
                csize, fmask = b.u(p, 4), b.u(p + 4, 4)
                offs = [b.u(p + 8 + 8 * i, 8) for i in range(rank)]
                child = b.u(p + ksz, 8)
                if level > 0:
                    walk(child)
                else:
                    data = b.bytes(child, csize)
                    for i, (fid, cv) in reversed(list(enumerate(filters))):
                        if fmask & (1 << i):
                            continue
                        if fid == 1:
                            data = zlib.decompress(data)
                        elif fid == 2:
                            k = cv[0] if cv else esize
                            data = np.frombuffer(data, np.uint8).reshape(k, -1).T.tobytes()
                        elif fid == 3:
                            data = data[:-4]                    # fletcher32 checksum trailer
                        else:
                            raise H5Error(f"filter {fid} not supported")
                    need = chunk_elems * esize
                    if len(data) < need:                        # writers may store a short edge chunk
                        data = data + b"\x00" * (need - len(data))
                    chunk = np.frombuffer(data[:need], dtype=f"V{esize}").reshape(cdims)
                    sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
                    arr[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
                p += ksz + 8
        walk(btree)
        return arr.tobytes()


class Group:
    def __init__(self, f: "File", obj: _Obj, name: str):
        self._f, self._o, self.name = f, obj, name
        self._links: Optional[Dict[str, int]] = None

    @property
    def attrs(self):
        return self._o.attrs

    def keys(self):
        if self._links is None:
            self._links = self._o.links()
        return list(self._links.keys())

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, path: str):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            if node._links is None:
                node._links = node._o.links()
            if part not in node._links:
                raise KeyError(path)
            obj = _Obj(self._f, node._links[part])
            child_name = node.name.rstrip("/") + "/" + part
            node = Group(self._f, obj, child_name) if obj.is_group() else Dataset(obj, child_name)
        return node


class Dataset:
    def __init__(self, obj: _Obj, name: str):
        self._o, self.name = obj, name

    @property
    def attrs(self):
        return self._o.attrs

    def read(self) -> np.ndarray:
        return self._o.read()

    def __getitem__(self, key):
        a = self.read()
        return a[()] if key == () else a[key]


class File(Group):
    """`File(path)[...]` mirrors the small part of h5py's API that the reference uses."""

    def __init__(self, path: str):
        with open(path, "rb") as fh:
            data = fh.read()
        self.b = _Buf(data)
        base = data.find(SIG)
        if base != 0:
            raise H5Error(f"{path}: not an HDF5 file (or it has a user block)")
        ver = self.b.u(8, 1)
        if ver not in (0, 1):
            raise H5Error(f"{path}: superblock version {ver} not supported")
        if self.b.u(13, 1) != 8 or self.b.u(14, 1) != 8:
            raise H5Error("only 8-byte offsets/lengths are supported")
        p = 24 + (4 if ver == 1 else 0)
        # base address, free-space address, end-of-file address, driver info address
        root_entry = p + 32
        root_addr = self.b.u(root_entry + 8, 8)
        super().__init__(self, _Obj(self, root_addr), "/")
        self._gheap: Dict[int, Dict[int, bytes]] = {}

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    # global heap (variable-length strings)
    def _gcol(self, addr: int) -> Dict[int, bytes]:
        if addr not in self._gheap:
            b = self.b
            if b.bytes(addr, 4) != b"GCOL":
                raise H5Error("bad global heap collection")
            size = b.u(addr + 8, 8)
            p, end = addr + 16, addr + size
            objs = {}
            while p + 16 <= end:
                idx, osz = b.u(p, 2), b.u(p + 8, 8)
                if idx == 0:
                    break
                objs[idx] = b.bytes(p + 16, osz)
                p += 16 + _pad8(osz)
            self._gheap[addr] = objs
        return self._gheap[addr]

    def _decode(self, typ: _Type, shape, raw: bytes, scalar: bool):
        if typ.vlen_str:
            n = max(1, int(np.prod(shape))) if shape else 1
            out = []
            for i in range(n):
                rec = raw[i * typ.size:(i + 1) * typ.size]
                ln = int.from_bytes(rec[0:4], "little")
                addr = int.from_bytes(rec[4:12], "little")
                idx = int.from_bytes(rec[12:16], "little")
                out.append(self._gcol(addr).get(idx, b"")[:ln].decode("utf8") if addr not in (0, UNDEF) else "")
            return out[0] if scalar else np.array(out, dtype=object).reshape(shape)
        a = np.frombuffer(raw, dtype=typ.dtype, count=(1 if scalar else int(np.prod(shape))))
        if scalar:
            v = a[0]
            return bytes(v) if typ.dtype.kind == "S" else v
        return a.reshape(shape).copy()


# ------------------------------------------------------------------------------------ users
def read_keras_weights(path: str) -> List[np.ndarray]:
    """The tensors of a Keras `save_weights` file in `load_weights` order: root attribute
    `layer_names`, then each layer group's `weight_names` (topology-free positional load)."""
    f = File(path)
    out = []
    for lname in f.attrs["layer_names"]:
        g = f[lname.decode() if isinstance(lname, bytes) else str(lname)]
        for wname in g.attrs.get("weight_names", []):
            out.append(np.asarray(g[wname.decode() if isinstance(wname, bytes) else str(wname)].read()))
    return out


def read_fast5(path: str, basecall_group: str = "Basecall_1D_000",
               basecall_subgroup: str = "BaseCalled_template") -> dict:
    """What nanorev_fast5_handeler.get_read_data / extract_fastq take from a single-read fast5:
    the Events table, the raw Signal, the Fastq record and the Albacore version / start_time."""
    f = File(path)
    try:
        grp = f["/Analyses/" + basecall_group]
    except KeyError:
        raise RuntimeError("No events or corrupted events in file. Likely a segmentation error .")
    version = grp.attrs.get("version", "0.0")
    if isinstance(version, bytes):
        version = version.decode()
    try:
        events = grp[basecall_subgroup + "/Events"].read()
    except KeyError:
        raise RuntimeError("No events or corrupted events in file. Likely a segmentation error .")
    try:
        reads = f["/Raw/Reads"]
        rname = reads.keys()[0]
        rg = reads[rname]
        signal = rg["Signal"].read()
        raw_attrs = dict(rg.attrs)
    except (KeyError, IndexError):
        raise RuntimeError("No signal stored in the file")
    fastq = None
    try:
        fastq = grp[basecall_subgroup + "/Fastq"].read()
    except KeyError:
        pass
    return {"events": events, "signal": signal, "fastq": fastq, "version": str(version),
            "raw_attrs": raw_attrs}
