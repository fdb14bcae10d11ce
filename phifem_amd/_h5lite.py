"""A minimal HDF5 writer and reader (no libhdf5, no h5py): groups + contiguous little-endian datasets, the subset
`dolfinx.io.XDMFFile` uses for heavy data (demo/weak-dirichlet/flower/main.py:193-195 writes `/Mesh/mesh/geometry`,
`/Mesh/mesh/topology` and `/Function/<name>/0`).

Format: HDF5 File Format Specification, version-0 superblock, version-1 object headers, version-1 group B-trees with
symbol-table nodes and a local heap per group -- the oldest layout, which every HDF5 reader (h5dump, h5py, ParaView,
VisIt) understands.  Datasets: f64 / f32 / i32 / i64 / u8 / i8, any rank, contiguous storage, no filters.
`write_h5(path, {"/Mesh/mesh/geometry": x, ...})`; tests/test_io_xdmf.py reads the files back with `h5dump`.
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 16, 16          # symbol-table nodes hold 2 K = 32 entries, a B-tree node 32 children


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(msgs):
    body = b"".join(msgs)
    return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body


def _datatype(dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        sign, esize, msize, bias = (63, 11, 52, 1023) if dt.itemsize == 8 else (31, 8, 23, 127)
        return struct.pack("<B3BI", 0x11, 0x20, sign, 0, dt.itemsize) + struct.pack(
            "<HHBBBBI", 0, 8 * dt.itemsize, msize, esize, 0, msize, bias)
    if dt.kind in "iu":
        return struct.pack("<B3BI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize) + struct.pack(
            "<HH", 0, 8 * dt.itemsize)
    raise NotImplementedError(f"HDF5 writer: dtype {dt}")


class _Node:
    def __init__(self):
        self.children = {}      # name -> _Node (group) or np.ndarray (dataset)


def write_h5(path, datasets):
    """datasets: {"/group/.../name": array}.  Arrays are written little-endian, C order, contiguous."""
    root = _Node()
    for full, arr in datasets.items():
        parts = [p for p in full.split("/") if p]
        if not parts:
            raise ValueError("empty dataset path")
        node = root
        for p in parts[:-1]:
            nxt = node.children.setdefault(p, _Node())
            if not isinstance(nxt, _Node):
                raise ValueError(f"{full}: {p} is a dataset")
            node = nxt
        arr = np.ascontiguousarray(arr)
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        if arr.dtype == np.bool_:
            arr = arr.astype(np.uint8)
        node.children[parts[-1]] = arr

    blob = bytearray(b"\0" * 96)        # the superblock is filled in last

    def alloc(data):
        data = _pad8(bytes(data))
        addr = len(blob)
        blob.extend(data)
        return addr

    def emit_dataset(arr):
        raw = arr.tobytes()
        data_addr = alloc(raw) if raw else UNDEF
        space = struct.pack("<BBB5x", 1, arr.ndim, 0) + b"".join(struct.pack("<Q", d) for d in arr.shape)
        fill = struct.pack("<BBBB", 2, 1, 0, 0)                        # early allocation, no fill value defined
        layout = struct.pack("<BBQQ", 3, 1, data_addr, len(raw))       # contiguous
        hdr = _object_header([_msg(0x0001, space), _msg(0x0003, _datatype(arr.dtype), flags=1),
                              _msg(0x0005, fill, flags=1), _msg(0x0008, layout)])
        return alloc(hdr)

    def emit_group(node):
        """-> (object header address, B-tree address, heap address)."""
        names = sorted(node.children)            # symbol-table entries are ordered by name (strcmp)
        names.sort(key=lambda s: s.encode())
        if len(names) > 4 * LEAF_K * INTERNAL_K:
            raise NotImplementedError("HDF5 writer: too many links in one group")
        entries = []
        for nm in names:
            ch = node.children[nm]
            if isinstance(ch, _Node):
                oh, bt, hp = emit_group(ch)
                entries.append((nm, oh, 1, struct.pack("<QQ", bt, hp)))
            else:
                entries.append((nm, emit_dataset(ch), 0, b"\0" * 16))
        # local heap: offset 0 holds the empty string, then the names, then one free block
        heap = bytearray(b"\0" * 8)
        offs = {}
        for nm, *_ in entries:
            offs[nm] = len(heap)
            heap.extend(_pad8(nm.encode() + b"\0"))
        free_off = len(heap)
        free_size = max(32, 256 - len(heap) % 256)
        heap.extend(struct.pack("<QQ", 1, free_size) + b"\0" * (free_size - 16))     # next = H5HL_FREE_NULL
        heap_data = alloc(heap)
        heap_addr = alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, heap_data))
        # symbol-table nodes of up to 2 K entries, one level-0 B-tree node over them
        snods, keys = [], [0]
        per = 2 * LEAF_K
        for i in range(0, max(len(entries), 1), per):
            chunk = entries[i:i + per]
            body = b"SNOD" + struct.pack("<BxH", 1, len(chunk))
            for nm, oh, cache, scratch in chunk:
                body += struct.pack("<QQI4x", offs[nm], oh, cache) + scratch
            body += b"\0" * (40 * (per - len(chunk)))
            snods.append(alloc(body))
            keys.append(offs[chunk[-1][0]] if chunk else 0)
        if not entries:
            snods, keys = [], [0]
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), UNDEF, UNDEF)
        for i, a in enumerate(snods):
            tree += struct.pack("<QQ", keys[i], a)
        tree += struct.pack("<Q", keys[len(snods)])
        tree += b"\0" * (24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8 - len(tree))
        bt_addr = alloc(tree)
        oh_addr = alloc(_object_header([_msg(0x0011, struct.pack("<QQ", bt_addr, heap_addr))]))
        return oh_addr, bt_addr, heap_addr

    oh, bt, hp = emit_group(root)
    sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0)
    sb += struct.pack("<HHI", LEAF_K, INTERNAL_K, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, len(blob), UNDEF)
    sb += struct.pack("<QQI4x", 0, oh, 1) + struct.pack("<QQ", bt, hp)
    assert len(sb) == 96
    blob[0:96] = sb
    with open(path, "wb") as f:
        f.write(blob)
    return path


# ------------------------------------------------------------------ reader (same subset)
class H5Unsupported(NotImplementedError):
    """The file uses a structure outside the subset read here (chunked / filtered data, new-style groups, ...)."""


class _Reader:
    def __init__(self, buf):
        self.b = buf
        if buf[:8] != b"\x89HDF\r\n\x1a\n":
            raise ValueError("not an HDF5 file")
        ver = buf[8]
        if ver not in (0, 1):
            raise H5Unsupported(f"superblock version {ver}")
        if buf[13] != 8 or buf[14] != 8:
            raise H5Unsupported("offsets / lengths that are not 8 bytes")
        p = 24 + (4 if ver == 1 else 0)
        self.base = struct.unpack_from("<Q", buf, p)[0]
        ste = p + 32
        self.root = struct.unpack_from("<Q", buf, ste + 8)[0]

    def messages(self, addr):
        """[(type, data bytes)] of a version-1 object header, continuation blocks followed."""
        b = self.b
        addr += self.base
        if b[addr] != 1:
            raise H5Unsupported(f"object header version {b[addr]}")
        nmsg, _, size = struct.unpack_from("<HII", b, addr + 2)
        blocks, out = [(addr + 16, size)], []
        while blocks and len(out) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize = struct.unpack_from("<HH", b, p)
                data = bytes(b[p + 8:p + 8 + msize])
                p += 8 + msize
                if mtype == 0x0010:
                    ca, cl = struct.unpack("<QQ", data[:16])
                    blocks.append((ca + self.base, cl))
                out.append((mtype, data))
        return out

    def children(self, addr):
        """{name: object header address} of an old-style group."""
        st = [d for t, d in self.messages(addr) if t == 0x0011]
        if not st:
            raise H5Unsupported("a group without a symbol table (new-style links)")
        bt, hp = struct.unpack("<QQ", st[0][:16])
        b = self.b
        hp += self.base
        if b[hp:hp + 4] != b"HEAP":
            raise ValueError("bad local heap")
        hdata = struct.unpack_from("<Q", b, hp + 24)[0] + self.base
        out = {}

        def walk(node):
            node += self.base
            if b[node:node + 4] == b"SNOD":
                n = struct.unpack_from("<H", b, node + 6)[0]
                for i in range(n):
                    off, oh = struct.unpack_from("<QQ", b, node + 8 + 40 * i)
                    s = hdata + off
                    e = s
                    while b[e]:
                        e += 1
                    out[bytes(b[s:e]).decode()] = oh
                return
            if b[node:node + 4] != b"TREE":
                raise ValueError("bad group B-tree node")
            n = struct.unpack_from("<H", b, node + 6)[0]
            for i in range(n):
                walk(struct.unpack_from("<Q", b, node + 24 + 8 + 16 * i)[0])
        walk(bt)
        return out

    def find(self, path):
        addr = self.root
        for p in [q for q in path.split("/") if q]:
            ch = self.children(addr)
            if p not in ch:
                raise KeyError(path)
            addr = ch[p]
        return addr

    def _filters(self, d):
        """Filter ids of a filter-pipeline message (versions 1 and 2)."""
        ver, nf = d[0], d[1]
        p = 8 if ver == 1 else 2
        ids = []
        for _ in range(nf):
            fid = struct.unpack_from("<H", d, p)[0]
            if ver == 1 or fid >= 256:
                nlen, _flags, ncv = struct.unpack_from("<HHH", d, p + 2)
                p += 8 + nlen + (-nlen % 8 if ver == 1 else 0)
            else:
                _flags, ncv = struct.unpack_from("<HH", d, p + 2)
                p += 6
            p += 4 * ncv + (4 if ver == 1 and ncv % 2 else 0)
            ids.append(fid)
        return ids

    def _chunked(self, d, shape, dtype, filters):
        """Chunked storage (layout class 2): a version-1 B-tree of type 1 over the chunks; deflate (1) and shuffle (2)
        are the filters understood (what meshio / h5py write by default for compressed data)."""
        import zlib
        rank1 = d[2]
        bt = struct.unpack_from("<Q", d, 3)[0]
        cdims = struct.unpack_from(f"<{rank1}I", d, 11)
        rank = rank1 - 1
        if rank != len(shape):
            raise H5Unsupported("chunk rank differs from the dataspace rank")
        if any(f not in (1, 2) for f in filters):
            raise H5Unsupported(f"filters {filters}")
        out = np.zeros(shape, dtype=dtype)
        b = self.b
        if bt == UNDEF:
            return out
        csh = cdims[:rank]

        def walk(node):
            node += self.base
            if b[node:node + 4] != b"TREE" or b[node + 4] != 1:
                raise ValueError("bad chunk B-tree node")
            level, n = b[node + 5], struct.unpack_from("<H", b, node + 6)[0]
            ksz = 8 + 8 * rank1
            p = node + 24
            for _ in range(n):
                nbytes, mask = struct.unpack_from("<II", b, p)
                offs = struct.unpack_from(f"<{rank1}Q", b, p + 8)
                child = struct.unpack_from("<Q", b, p + ksz)[0]
                p += ksz + 8
                if level > 0:
                    walk(child)
                    continue
                raw = bytes(b[child + self.base:child + self.base + nbytes])
                for k in range(len(filters) - 1, -1, -1):       # undo the pipeline back to front
                    if mask >> k & 1:
                        continue
                    if filters[k] == 1:
                        raw = zlib.decompress(raw)
                    else:
                        raw = np.frombuffer(raw, np.uint8).reshape(dtype.itemsize, -1).T.tobytes()
                chunk = np.frombuffer(raw, dtype=dtype, count=int(np.prod(csh))).reshape(csh)
                sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, csh, shape))
                out[sl] = chunk[tuple(slice(0, t.stop - t.start) for t in sl)]
        walk(bt)
        return out

    def dataset(self, path):
        shape = dtype = data = layout = None
        filters = []
        for t, d in self.messages(self.find(path)):
            if t == 0x0001:
                ver, rank = d[0], d[1]
                off = 8 if ver == 1 else 4
                shape = struct.unpack_from(f"<{rank}Q", d, off)
            elif t == 0x0003:
                cls, bits0, size = d[0] & 15, d[1], struct.unpack_from("<I", d, 4)[0]
                end = ">" if bits0 & 1 else "<"
                if cls == 0:
                    dtype = np.dtype(f"{end}{'i' if bits0 & 8 else 'u'}{size}")
                elif cls == 1:
                    dtype = np.dtype(f"{end}f{size}")
                else:
                    raise H5Unsupported(f"datatype class {cls}")
            elif t == 0x0008:
                if d[0] != 3:
                    raise H5Unsupported(f"data layout message version {d[0]}")
                layout = d
            elif t == 0x000B:
                filters = self._filters(d)
        if shape is None or dtype is None or layout is None:
            raise H5Unsupported(f"{path}: not a plain dataset")
        native = dtype.newbyteorder("=")
        if layout[1] == 2:
            return self._chunked(layout, shape, dtype, filters).astype(native)
        if layout[1] == 1:
            a, n = struct.unpack_from("<QQ", layout, 2)
            data = b"" if a == UNDEF else self.b[a + self.base:a + self.base + n]
        else:
            n = struct.unpack_from("<H", layout, 2)[0]
            data = layout[4:4 + n]
        n = int(np.prod(shape)) if shape else 1
        return np.frombuffer(data, dtype=dtype, count=n).reshape(shape).astype(native)


def read_h5(path, dataset):
    """One dataset of an HDF5 file in the subset `write_h5` writes (which is also what dolfinx's XDMFFile writes for
    unchunked data); raises H5Unsupported for anything else, so the caller can fall back to h5py / h5dump."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    return _Reader(buf).dataset(dataset)


def list_h5(path):
    """All dataset paths of a file (depth first, sorted by name)."""
    with open(path, "rb") as f:
        r = _Reader(memoryview(f.read()))
    out = []

    def rec(addr, prefix):
        for nm, a in sorted(r.children(addr).items()):
            if any(t == 0x0011 for t, _ in r.messages(a)):
                rec(a, f"{prefix}/{nm}")
            else:
                out.append(f"{prefix}/{nm}")
    rec(r.root, "")
    return out
