"""HDF5 container of nnabla parameter files, read and written without h5py / libhdf5.

The reference stores and restores its networks with `nn.save_parameters(".../*.h5")` / `nn.load_parameters(path)`
(python/train.py:100-101, python/render_image.py:43, python/extract_by_mc.py:300).  nnabla (third party, not vendored in
the reference; 1.x `nnabla/parameter.py` save_parameters / load_parameters) writes, through h5py with default settings,

    for i, (k, v) in enumerate(params.items()):   hd[k] = v.d;  hd[k].attrs['need_grad'] = v.need_grad
                                                  hd[k].attrs['index'] = i

i.e. one contiguous little-endian float32 dataset per parameter under old-style (symbol-table) groups named by the
"/"-separated scope, each with a scalar enum{FALSE,TRUE}:int8 attribute `need_grad` and a scalar int64 `index`; loading
visits every dataset and installs them sorted by (index, name).

This module is the subset of the HDF5 1.8/1.10 file format (format specification version 2.0) those files use:
superblock v0/v1, version-1 object headers with continuation blocks, symbol-table groups (version-1 B-trees of symbol
nodes over a local heap), dataspace v1/v2, fixed-point / floating-point / enum datatypes, contiguous and compact
layouts, attribute messages v1-v3.  Anything else (chunked / filtered datasets, new-style groups, superblock v2+)
raises `H5FormatError` naming what was met.  The writer emits the same structures libhdf5 does for such a file (stock
K values, flags and message order) and is checked by libhdf5's own tools where they exist (tests/test_h5params_cpu.py).
"""
import struct
from collections import OrderedDict

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 4, 16              # libhdf5 defaults (H5Pset_sym_k): 8 symbols per node, 32 children per B-tree node
HEAP_FREE_NULL = 1                      # "no free block" marker of a local heap's free list

MSG_NIL, MSG_DATASPACE, MSG_DATATYPE, MSG_FILL_OLD, MSG_FILL, MSG_LAYOUT = 0x0, 0x1, 0x3, 0x4, 0x5, 0x8
MSG_FILTER, MSG_ATTRIBUTE, MSG_CONTINUATION, MSG_SYMBOL_TABLE, MSG_LINK_INFO = 0xB, 0xC, 0x10, 0x11, 0x2


class H5FormatError(ValueError):
    pass


def _pad8(n):
    return (n + 7) & ~7


# ------------------------------------------------------------------------------------------------------------ reader
class _Reader:
    def __init__(self, f):
        self.f = f
        head = self._at(0, 8)
        if head != SIGNATURE:
            raise H5FormatError("not an HDF5 file (signature at offset 0 missing; user blocks are not supported)")
        ver = self._at(8, 1)[0]
        if ver > 1:
            raise H5FormatError(f"superblock version {ver}: only the symbol-table formats (0, 1) nnabla/h5py write by default")
        so, sl = self._at(13, 2)
        if (so, sl) != (8, 8):
            raise H5FormatError(f"offset / length sizes {so}/{sl}: only 8/8")
        p = 24 if ver == 0 else 28
        self.base, _free, self.eof, _drv = struct.unpack("<4Q", self._at(p, 32))
        if self.base != 0:
            raise H5FormatError("non-zero base address")
        _name, self.root_header, _cache, _r = struct.unpack("<QQII", self._at(p + 32, 24))

    def _at(self, addr, n):
        self.f.seek(addr)
        b = self.f.read(n)
        if len(b) != n:
            raise H5FormatError(f"truncated file: {n} bytes wanted at {addr}")
        return b

    # --- object headers ---------------------------------------------------------------------------------------
    def messages(self, addr):
        """[(type, flags, body)] of a version-1 object header, continuation blocks followed."""
        ver, _, nmsg, _ref, size = struct.unpack("<BBHII", self._at(addr, 12))
        if ver != 1:
            raise H5FormatError(f"object header version {ver} at {addr} (signature {self._at(addr, 4)!r}): only version 1")
        blocks, out = [(addr + 16, size)], []
        while blocks and len(out) < nmsg:
            start, length = blocks.pop(0)
            raw, p = self._at(start, length), 0
            while p + 8 <= length and len(out) < nmsg:
                mtype, msize, flags = struct.unpack("<HHB", raw[p:p + 5])
                body = raw[p + 8:p + 8 + msize]
                if flags & 2:
                    raise H5FormatError(f"shared header message (type {mtype:#x}) at {start + p}")
                if mtype == MSG_CONTINUATION:
                    blocks.append(struct.unpack("<QQ", body[:16]))
                out.append((mtype, flags, body))
                p += 8 + msize
        return out

    # --- groups -----------------------------------------------------------------------------------------------
    def _heap_name(self, heap_data, offset):
        self.f.seek(heap_data + offset)
        out = b""
        while True:
            chunk = self.f.read(64)
            if not chunk:
                raise H5FormatError("unterminated link name")
            i = chunk.find(b"\0")
            if i >= 0:
                return (out + chunk[:i]).decode("utf-8")
            out += chunk

    def _tree(self, addr, heap_data, out):
        head = self._at(addr, 8)
        if head[:4] == b"SNOD":
            count = struct.unpack("<H", head[6:8])[0]
            raw = self._at(addr + 8, 40 * count)
            for i in range(count):
                name_off, header, cache = struct.unpack("<QQI", raw[40 * i:40 * i + 20])
                if cache == 2:
                    raise H5FormatError("symbolic link in a group")
                out.append((self._heap_name(heap_data, name_off), header))
            return
        if head[:4] != b"TREE" or head[4] != 0:
            raise H5FormatError(f"expected a group B-tree node or symbol node at {addr}, found {head!r}")
        used = struct.unpack("<H", head[6:8])[0]
        raw = self._at(addr + 24, 16 * used)
        for i in range(used):
            self._tree(struct.unpack("<Q", raw[16 * i + 8:16 * i + 16])[0], heap_data, out)

    def links(self, messages):
        """[(name, object header address)] of a group, None for a non-group."""
        for mtype, _flags, body in messages:
            if mtype == MSG_SYMBOL_TABLE:
                btree, heap = struct.unpack("<QQ", body[:16])
                hh = self._at(heap, 32)
                if hh[:4] != b"HEAP":
                    raise H5FormatError(f"local heap signature missing at {heap}")
                heap_data = struct.unpack("<Q", hh[24:32])[0]
                out = []
                self._tree(btree, heap_data, out)
                return out
            if mtype == MSG_LINK_INFO:
                raise H5FormatError("new-style (link message / fractal heap) group: write the file with h5py's default libver")
        return None

    # --- datasets ---------------------------------------------------------------------------------------------
    @staticmethod
    def dataspace(body):
        ver, rank, flags = body[0], body[1], body[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            p = 4
            if body[3] == 2:
                raise H5FormatError("null dataspace")
        else:
            raise H5FormatError(f"dataspace message version {ver}")
        return tuple(struct.unpack(f"<{rank}Q", body[p:p + 8 * rank]))

    @classmethod
    def datatype(cls, body):
        """-> (numpy dtype, enum names or None, bytes consumed)."""
        cls_ver, b0, _b1, _b2, size = struct.unpack("<BBBBI", body[:8])
        klass, ver = cls_ver & 15, cls_ver >> 4
        order = ">" if b0 & 1 else "<"
        if klass == 0:
            return np.dtype(f"{order}{'i' if b0 & 8 else 'u'}{size}"), None, 12
        if klass == 1:
            if size not in (2, 4, 8):
                raise H5FormatError(f"{size}-byte floating point")
            return np.dtype(f"{order}f{size}"), None, 20
        if klass == 8:
            count = b0 | (_b1 << 8)
            base, _, used = cls.datatype(body[8:])
            p, names = 8 + used, []
            for _ in range(count):
                end = body.index(b"\0", p)
                names.append(body[p:end].decode("ascii"))
                p = end + 1 if ver >= 3 else p + _pad8(end + 1 - p)
            values = np.frombuffer(body[p:p + count * base.itemsize], base)
            return base, dict(zip(values.tolist(), names)), p + count * base.itemsize
        raise H5FormatError(f"datatype class {klass} (only fixed-point, floating-point and enum)")

    def attribute(self, body):
        ver = body[0]
        if ver not in (1, 2, 3):
            raise H5FormatError(f"attribute message version {ver}")
        nsz, tsz, ssz = struct.unpack("<HHH", body[2:8])
        p = 8 + (1 if ver == 3 else 0)
        step = _pad8 if ver == 1 else (lambda n: n)
        name = body[p:p + nsz].split(b"\0")[0].decode("utf-8")
        p += step(nsz)
        dtype, enum, _ = self.datatype(body[p:p + tsz])
        p += step(tsz)
        shape = self.dataspace(body[p:p + ssz])
        p += step(ssz)
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        value = np.frombuffer(body[p:p + n * dtype.itemsize], dtype).reshape(shape)
        if enum is not None and set(enum.values()) == {"FALSE", "TRUE"}:     # h5py's mapping of numpy.bool_
            value = np.array([enum[v] == "TRUE" for v in value.reshape(-1).tolist()], bool).reshape(shape)
        return name, (value[()] if shape == () else value)

    def dataset(self, messages, name, mmap_path=None):
        shape = dtype = layout = None
        attrs = {}
        for mtype, _flags, body in messages:
            if mtype == MSG_DATASPACE:
                shape = self.dataspace(body)
            elif mtype == MSG_DATATYPE:
                dtype, enum, _ = self.datatype(body)
                if enum is not None:
                    raise H5FormatError(f"{name}: enum dataset")
            elif mtype == MSG_LAYOUT:
                layout = body
            elif mtype == MSG_FILTER:
                raise H5FormatError(f"{name}: filtered (compressed) dataset")
            elif mtype == MSG_ATTRIBUTE:
                k, v = self.attribute(body)
                attrs[k] = v
        if shape is None or dtype is None or layout is None:
            raise H5FormatError(f"{name}: object is neither a group nor a dataset")
        if layout[0] != 3:
            raise H5FormatError(f"{name}: data layout message version {layout[0]} (only 3)")
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if layout[1] == 1:
            addr, size = struct.unpack("<QQ", layout[2:18])
            if addr == UNDEF:                                         # never written: fill value (zeros)
                data = np.zeros(n, dtype)
            else:
                if size < n * dtype.itemsize:
                    raise H5FormatError(f"{name}: {size} bytes stored for {n} x {dtype}")
                self.f.seek(addr)
                data = np.fromfile(self.f, dtype, n)
                if data.size != n:
                    raise H5FormatError(f"{name}: raw data truncated")
        elif layout[1] == 0:
            size = struct.unpack("<H", layout[2:4])[0]
            data = np.frombuffer(layout[4:4 + size], dtype, n).copy()
        else:
            raise H5FormatError(f"{name}: chunked dataset (nnabla writes contiguous ones)")
        return data.reshape(shape), attrs

    def visit(self, addr, prefix, out, seen):
        if addr in seen:
            raise H5FormatError("hard-link cycle")
        msgs = self.messages(addr)
        links = self.links(msgs)
        if links is None:
            out[prefix] = self.dataset(msgs, prefix)
            return
        for name, child in links:
            self.visit(child, f"{prefix}/{name}" if prefix else name, out, seen | {addr})


def read_h5(path):
    """{full name: (array, {attribute: value})} of every dataset in the file, in the groups' (alphabetical) link order."""
    with open(path, "rb") as f:
        r = _Reader(f)
        out = OrderedDict()
        r.visit(r.root_header, "", out, frozenset())
    return out


def load_nnabla_h5(path):
    """[(name, float32 array, need_grad)] in nnabla's load order: sorted by (`index` attribute, name)
    (nnabla/parameter.py load_parameters: `hd.visit(_get_keys)`, `for _, key in sorted(keys)`)."""
    items = read_h5(path)
    keyed = []
    for name, (arr, attrs) in items.items():
        idx = attrs.get("index")
        keyed.append(((0, int(idx)) if idx is not None else (1, 0), name))
    out = []
    for _, name in sorted(keyed):
        arr, attrs = items[name]
        if "need_grad" not in attrs:
            raise H5FormatError(f"{name}: no `need_grad` attribute (not an nnabla parameter file)")
        out.append((name, np.asarray(arr, dtype=np.float32, order="C"), bool(attrs["need_grad"])))
    return out


# ------------------------------------------------------------------------------------------------------------ writer
_F32_TYPE = struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
_I64_TYPE = struct.pack("<BBBBI", 0x10, 0x08, 0x00, 0x00, 8) + struct.pack("<HH", 0, 64)
_I8_TYPE = struct.pack("<BBBBI", 0x10, 0x08, 0x00, 0x00, 1) + struct.pack("<HH", 0, 8)
_BOOL_TYPE = struct.pack("<BBBBI", 0x18, 0x02, 0x00, 0x00, 1) + _I8_TYPE + b"FALSE\0\0\0" + b"TRUE\0\0\0\0" + b"\x00\x01"
_SCALAR_SPACE = struct.pack("<BBBB4x", 1, 0, 0, 0)


def _message(mtype, body, flags=0):
    body = body + b"\0" * (_pad8(len(body)) - len(body))
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _attribute(name, dtype_bytes, value_bytes):
    nm = name.encode("ascii") + b"\0"
    body = struct.pack("<BxHHH", 1, len(nm), len(dtype_bytes), len(_SCALAR_SPACE))
    for part in (nm, dtype_bytes, _SCALAR_SPACE):
        body += part + b"\0" * (_pad8(len(part)) - len(part))
    return _message(MSG_ATTRIBUTE, body + value_bytes, flags=4)


def _object_header(messages):
    body = b"".join(messages)
    return struct.pack("<BxHII4x", 1, len(messages), 1, len(body)) + body


def _dataset_header(shape, data_addr, nbytes, need_grad, index):
    rank = len(shape)
    space = struct.pack("<BBBB4x", 1, rank, 1, 0) + struct.pack(f"<{2 * rank}Q", *shape, *shape)
    return _object_header([
        _message(MSG_DATASPACE, space),
        _message(MSG_DATATYPE, _F32_TYPE, flags=1),
        _message(MSG_FILL, struct.pack("<BBBBI", 2, 2, 2, 1, 0), flags=1),
        _message(MSG_LAYOUT, struct.pack("<BBQQ", 3, 1, data_addr if nbytes else UNDEF, nbytes)),
        _attribute("need_grad", _BOOL_TYPE, b"\x01" if need_grad else b"\x00"),
        _attribute("index", _I64_TYPE, struct.pack("<q", index)),
    ])


class _Group:
    def __init__(self):
        self.children = OrderedDict()          # name -> _Group | dataset record


class _Layout:
    """Assigns addresses front to back; every structure is a (address, bytes) pair produced once its children's
    addresses are known, so groups are laid out depth-first with the parent's pieces reserved before the children."""

    def __init__(self):
        self.pos = 0
        self.pieces = []

    def reserve(self, n):
        a = self.pos
        self.pos += _pad8(n)
        return a

    def put(self, addr, b):
        self.pieces.append((addr, b))


def _plan_group(lay, group):
    """Reserves and fills the object header, B-tree, heap and symbol nodes of `group`; returns (header, btree, heap)."""
    names = sorted(group.children, key=lambda s: s.encode("utf-8"))        # strcmp order
    # local heap: offset 0 = empty string (the B-tree's first key), then the names, then one free block
    offsets, seg = {}, bytearray(8)
    for n in names:
        offsets[n] = len(seg)
        e = n.encode("utf-8") + b"\0"
        seg += e + b"\0" * (_pad8(len(e)) - len(e))
    free_at = len(seg)
    seg += struct.pack("<QQ", HEAP_FREE_NULL, 32) + b"\0" * 16
    # symbol nodes of up to 2 * LEAF_K entries, B-tree levels of up to 2 * INTERNAL_K children
    per = 2 * LEAF_K
    leaves = [names[i:i + per] for i in range(0, len(names), per)]
    header = lay.reserve(16 + 24)
    counts, w = [], len(leaves)
    while True:
        w = max(1, -(-w // (2 * INTERNAL_K)))
        counts.append(w)
        if w == 1:
            break
    node_size = 24 + 8 * (2 * INTERNAL_K + 1) + 16 * INTERNAL_K
    node_addr = [[lay.reserve(node_size) for _ in range(c)] for c in counts]         # [level][i]
    heap = lay.reserve(32)
    heap_data = lay.reserve(len(seg))
    leaf_addr = [lay.reserve(8 + 40 * per) for _ in leaves]
    lay.put(header, _object_header([_message(MSG_SYMBOL_TABLE, struct.pack("<QQ", node_addr[-1][0], heap))]))
    lay.put(heap, b"HEAP" + struct.pack("<B3xQQQ", 0, len(seg), free_at, heap_data))
    lay.put(heap_data, bytes(seg))
    # children first need their addresses: plan them now (depth first), then write the symbol nodes
    entry = {}
    for n in names:
        c = group.children[n]
        if isinstance(c, _Group):
            h, bt, hp = _plan_group(lay, c)
            entry[n] = struct.pack("<QQII", offsets[n], h, 1, 0) + struct.pack("<QQ", bt, hp)
        else:
            c["header"] = lay.reserve(len(_dataset_header(c["shape"], 0, 0, True, 0)))
            entry[n] = struct.pack("<QQII", offsets[n], c["header"], 0, 0) + b"\0" * 16
    for a, leaf in zip(leaf_addr, leaves):
        body = b"".join(entry[n] for n in leaf)
        lay.put(a, b"SNOD" + struct.pack("<BxH", 1, len(leaf)) + body + b"\0" * (40 * (per - len(leaf))))
    # B-tree: (address, largest name offset) of the children of each level, bottom up
    below = [(a, offsets[leaf[-1]]) for a, leaf in zip(leaf_addr, leaves)]
    for level, addrs in enumerate(node_addr):
        above = []
        for i, a in enumerate(addrs):
            kids = below[i * 2 * INTERNAL_K:(i + 1) * 2 * INTERNAL_K]
            first_key = 0 if i == 0 else below[i * 2 * INTERNAL_K - 1][1]
            body = struct.pack("<Q", first_key) + b"".join(struct.pack("<QQ", child, key) for child, key in kids)
            left = addrs[i - 1] if i > 0 else UNDEF
            right = addrs[i + 1] if i + 1 < len(addrs) else UNDEF
            node = b"TREE" + struct.pack("<BBHQQ", 0, level, len(kids), left, right) + body
            lay.put(a, node + b"\0" * (node_size - len(node)))
            above.append((a, kids[-1][1] if kids else 0))
        below = above
    return header, node_addr[-1][0], heap


def write_nnabla_h5(path, params, data_align=8):
    """Writes `params` -- an ordered iterable of (name, array-like float32, need_grad) -- as an nnabla `.h5` parameter
    file: what `nn.save_parameters(path)` produces through h5py (module docstring).  Arrays may be numpy arrays or torch
    tensors (copied to the host one at a time while their bytes are streamed, so a 2 GiB grid is never held twice)."""
    root, records = _Group(), []
    for index, (name, arr, need_grad) in enumerate(params):
        parts = [p for p in name.split("/") if p]
        if not parts:
            raise ValueError("empty parameter name")
        shape = tuple(int(s) for s in arr.shape)
        g = root
        for p in parts[:-1]:
            nxt = g.children.setdefault(p, _Group())
            if not isinstance(nxt, _Group):
                raise ValueError(f"{name}: {p!r} is already a parameter, not a scope")
            g = nxt
        if parts[-1] in g.children:
            raise ValueError(f"{name}: duplicate name")
        rec = dict(name=name, shape=shape, source=arr, need_grad=bool(need_grad), index=index,
                   nbytes=4 * int(np.prod(shape, dtype=np.int64)) if shape else 4)
        g.children[parts[-1]] = rec
        records.append(rec)
    lay = _Layout()
    lay.reserve(96)
    header, btree, heap = _plan_group(lay, root)
    for rec in records:                                  # raw data behind all metadata
        lay.pos = -(-lay.pos // data_align) * data_align
        rec["data"] = lay.pos
        lay.pos += rec["nbytes"]
        lay.put(rec["header"], _dataset_header(rec["shape"], rec["data"], rec["nbytes"], rec["need_grad"], rec["index"]))
    eof = lay.pos
    superblock = (SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
                  + struct.pack("<4Q", 0, UNDEF, eof, UNDEF)
                  + struct.pack("<QQII", 0, header, 1, 0) + struct.pack("<QQ", btree, heap))
    assert len(superblock) == 96
    with open(path, "wb") as f:
        f.write(superblock)
        for addr, b in sorted(lay.pieces, key=lambda t: t[0]):
            f.seek(addr)
            f.write(b)
        for rec in records:
            src = rec["source"]
            if hasattr(src, "detach"):
                src = src.detach().cpu().numpy()
            a = np.asarray(src, dtype="<f4", order="C")
            if tuple(a.shape) != rec["shape"]:
                raise ValueError(f"{rec['name']}: shape changed while writing")
            f.seek(rec["data"])
            a.tofile(f)
        f.truncate(eof)
