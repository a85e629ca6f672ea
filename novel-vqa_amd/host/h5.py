"""Minimal HDF5 reader / writer for the two data files either side of the training step.

The reference reads `data_prepro.h5` (uint32 question / length / img_pos / answer arrays written
by h5py, 002_train_vqa_arch1/000_prepro_vqa.py:273-300) and `data_img.h5` (float32 feature
matrices written by torch-hdf5, 001_prepro_img_vgg.lua:156-160) with `hdf5.open(...):read(name):all()`
(002_train_baseline.lua:89-111).  Neither h5py nor libhdf5 is available to this package, so this
is a from-the-specification reader of the subset those writers produce (HDF5 File Format
Specification 3.0): superblock 0-3, version-1 and version-2 object headers, old-style groups
(symbol table: v1 B-tree + local heap + SNOD) and compact new-style groups (link messages),
fixed-point and IEEE float datatypes of either byte order, compact / contiguous / chunked (v1
B-tree index; deflate, shuffle, fletcher32 filters) layouts.  Anything else raises H5Error.

`write()` emits what torch-hdf5's `file:write('/name', tensor)` emits in its default
configuration: superblock 0, one root group, contiguous little-endian datasets.

Pinned against files written by libhdf5 itself (tests/golden/h5/*.h5, made with h5py by
tests/golden/make_h5_fixtures.py using the reference's own create_dataset calls).
"""
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(Exception):
    pass


class _Dataset:
    __slots__ = ("shape", "dtype", "layout", "filters", "name")

    def __init__(self):
        self.shape = None
        self.dtype = None
        self.layout = None
        self.filters = []
        self.name = ""


class File:
    """Read-only view: `File(path).read('/ques_train')` -> numpy array; `keys()`; `shape(name)`."""

    def __init__(self, path):
        self.path = path
        self.f = open(path, "rb")
        self._find_superblock()
        self._root_links = None

    def close(self):
        if self.f:
            self.f.close()
            self.f = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- low level ----------------------------------------------------------------------------
    def _at(self, addr, n):
        self.f.seek(self.base + addr)
        b = self.f.read(n)
        if len(b) != n:
            raise H5Error(f"{self.path}: truncated file (wanted {n} bytes at {addr})")
        return b

    def _at_most(self, addr, n):
        self.f.seek(self.base + addr)
        return self.f.read(n)

    def _off(self, b, p):
        return int.from_bytes(b[p:p + self.O], "little")

    def _len(self, b, p):
        return int.from_bytes(b[p:p + self.L], "little")

    def _find_superblock(self):
        pos = 0
        self.base = 0
        while True:
            self.f.seek(pos)
            head = self.f.read(8)
            if head == SIGNATURE:
                break
            if len(head) < 8 or pos > (1 << 26):
                raise H5Error(f"{self.path}: not an HDF5 file (no superblock signature)")
            pos = 512 if pos == 0 else pos * 2
        self.f.seek(pos)
        sb = self.f.read(128)
        ver = sb[8]
        if ver in (0, 1):
            self.O, self.L = sb[13], sb[14]
            p = 24 + (4 if ver == 1 else 0)
            self.base = self._off(sb, p)
            p += 4 * self.O  # base, free-space, end-of-file, driver-info
            # root group symbol table entry: link name offset, object header address, cache type, ...
            self.root_header = self._off(sb, p + self.O)
        elif ver in (2, 3):
            self.O, self.L = sb[9], sb[10]
            self.base = self._off(sb, 12)
            self.root_header = self._off(sb, 12 + 3 * self.O)
        else:
            raise H5Error(f"{self.path}: unsupported superblock version {ver}")
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise H5Error(f"{self.path}: unsupported offset/length sizes {self.O}/{self.L}")
        if self.base != 0 and pos != 0:
            pass  # user block: addresses are relative to the base address field

    # -- object headers -----------------------------------------------------------------------
    def _messages(self, addr):
        """-> list of (type, flags, bytes) of the object header at addr (continuations followed)."""
        head = self._at(addr, 16)
        msgs = []
        if head[:4] == b"OHDR":
            if head[4] != 2:
                raise H5Error("unsupported object header version")
            flags = head[5]
            p = 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szb = 1 << (flags & 3)
            hdr = self._at(addr, p + szb)
            size0 = int.from_bytes(hdr[p:p + szb], "little")
            blocks = [(addr + p + szb, size0)]
            track = bool(flags & 0x04)
            while blocks:
                baddr, bsize = blocks.pop(0)
                blk = self._at(baddr, bsize)
                q = 0
                while q + 4 <= bsize:
                    mtype = blk[q]
                    msize = int.from_bytes(blk[q + 1:q + 3], "little")
                    mflags = blk[q + 3]
                    q += 4 + (2 if track else 0)
                    if q + msize > bsize:
                        break
                    body = blk[q:q + msize]
                    q += msize
                    if mtype == 0x10:
                        caddr, clen = self._off(body, 0), self._len(body, self.O)
                        if self._at(caddr, 4) != b"OCHK":
                            raise H5Error("bad object header continuation")
                        blocks.append((caddr + 4, clen - 8))  # minus signature and checksum
                    elif mtype != 0:
                        msgs.append((mtype, mflags, body))
            return msgs
        if head[0] != 1:
            raise H5Error(f"unsupported object header version {head[0]} at {addr}")
        nmsg = int.from_bytes(head[2:4], "little")
        size0 = int.from_bytes(head[8:12], "little")
        blocks = [(addr + 16, size0)]
        while blocks and len(msgs) < nmsg + 64:
            baddr, bsize = blocks.pop(0)
            blk = self._at(baddr, bsize)
            q = 0
            while q + 8 <= bsize:
                mtype = int.from_bytes(blk[q:q + 2], "little")
                msize = int.from_bytes(blk[q + 2:q + 4], "little")
                mflags = blk[q + 4]
                body = blk[q + 8:q + 8 + msize]
                q += 8 + msize
                if mtype == 0x10:
                    blocks.append((self._off(body, 0), self._len(body, self.O)))
                elif mtype != 0:
                    msgs.append((mtype, mflags, body))
        return msgs

    # -- groups ------------------------------------------------------------------------------
    def _heap_string(self, heap_data_addr, off):
        out = b""
        while True:
            chunk = self._at_most(heap_data_addr + off + len(out), 64)
            z = chunk.find(b"\0")
            if z >= 0:
                return (out + chunk[:z]).decode("utf-8")
            if not chunk:
                raise H5Error("unterminated name in the local heap")
            out += chunk

    def _group_btree(self, addr, heap_data_addr, links):
        node = self._at(addr, 8 + 2 * self.O)
        if node[:4] != b"TREE" or node[4] != 0:
            raise H5Error("bad group B-tree node")
        level = node[5]
        used = int.from_bytes(node[6:8], "little")
        body = self._at(addr + 8 + 2 * self.O, used * (self.O + self.L) + self.L)
        p = self.L  # skip key 0
        for _ in range(used):
            child = self._off(body, p)
            p += self.O + self.L
            if level > 0:
                self._group_btree(child, heap_data_addr, links)
            else:
                sn = self._at(child, 8)
                if sn[:4] != b"SNOD":
                    raise H5Error("bad symbol table node")
                nsym = int.from_bytes(sn[6:8], "little")
                esz = 2 * self.O + 24
                ents = self._at(child + 8, nsym * esz)
                for i in range(nsym):
                    e = ents[i * esz:(i + 1) * esz]
                    links[self._heap_string(heap_data_addr, self._off(e, 0))] = self._off(e, self.O)

    def _links(self, header_addr):
        links = {}
        for mtype, _, body in self._messages(header_addr):
            if mtype == 0x11:  # symbol table: B-tree + local heap
                btree, heap = self._off(body, 0), self._off(body, self.O)
                h = self._at(heap, 8 + 2 * self.L + self.O)
                if h[:4] != b"HEAP":
                    raise H5Error("bad local heap")
                self._group_btree(btree, self._off(h, 8 + 2 * self.L), links)
            elif mtype == 0x06:  # link message (compact new-style group)
                ver, fl = body[0], body[1]
                p = 2
                ltype = 0
                if fl & 0x08:
                    ltype = body[p]
                    p += 1
                if fl & 0x04:
                    p += 8
                if fl & 0x10:
                    p += 1
                lsz = 1 << (fl & 3)
                nlen = int.from_bytes(body[p:p + lsz], "little")
                p += lsz
                name = body[p:p + nlen].decode("utf-8")
                p += nlen
                if ltype == 0:
                    links[name] = self._off(body, p)
            elif mtype == 0x02:  # link info: dense storage needs the fractal heap
                if self._off(body, 2 + (8 if body[1] & 1 else 0)) != (UNDEF >> (64 - 8 * self.O)):
                    raise H5Error("dense (fractal-heap) groups are not supported")
        return links

    def _resolve(self, name):
        addr = self.root_header
        for part in [p for p in name.split("/") if p]:
            links = self._links(addr)
            if part not in links:
                raise KeyError(f"{self.path}: no object '{name}'")
            addr = links[part]
        return addr

    def keys(self, group="/"):
        return sorted(self._links(self._resolve(group)))

    # -- datasets ----------------------------------------------------------------------------
    def _dataset(self, name):
        ds = _Dataset()
        ds.name = name
        for mtype, _, body in self._messages(self._resolve(name)):
            if mtype == 0x01:
                ver, rank, fl = body[0], body[1], body[2]
                p = 8 if ver == 1 else 4
                ds.shape = tuple(self._len(body, p + i * self.L) for i in range(rank))
            elif mtype == 0x03:
                ds.dtype = _parse_dtype(body)
            elif mtype == 0x08:
                ds.layout = self._parse_layout(body)
            elif mtype == 0x0B:
                ds.filters = _parse_filters(body)
        if ds.shape is None or ds.dtype is None or ds.layout is None:
            raise H5Error(f"{self.path}: '{name}' is not a simple dataset")
        return ds

    def _parse_layout(self, b):
        ver = b[0]
        if ver in (3, 4):  # version 4 differs only in its chunk indexes
            cls = b[1]
            if cls == 0:
                n = int.from_bytes(b[2:4], "little")
                return ("compact", bytes(b[4:4 + n]))
            if cls == 1:
                return ("contiguous", self._off(b, 2), self._len(b, 2 + self.O))
            if cls == 2 and ver == 3:
                nd = b[2]
                bt = self._off(b, 3)
                dims = [int.from_bytes(b[3 + self.O + 4 * i:7 + self.O + 4 * i], "little") for i in range(nd)]
                return ("chunked", bt, dims)
        elif ver in (1, 2):
            nd, cls = b[1], b[2]
            p = 8
            addr = None
            if cls != 0:
                addr = self._off(b, p)
                p += self.O
            dims = [int.from_bytes(b[p + 4 * i:p + 4 * i + 4], "little") for i in range(nd)]
            p += 4 * nd
            if cls == 0:
                n = int.from_bytes(b[p:p + 4], "little")
                return ("compact", bytes(b[p + 4:p + 4 + n]))
            if cls == 1:
                return ("contiguous", addr, None)
            if cls == 2:
                return ("chunked", addr, dims)
        raise H5Error(f"unsupported data layout (version {ver})")

    def _chunks(self, addr, nd, out):
        node = self._at(addr, 8 + 2 * self.O)
        if node[:4] != b"TREE" or node[4] != 1:
            raise H5Error("bad chunk B-tree node")
        level = node[5]
        used = int.from_bytes(node[6:8], "little")
        ksz = 8 + 8 * nd
        body = self._at(addr + 8 + 2 * self.O, used * (ksz + self.O) + ksz)
        for i in range(used):
            k = body[i * (ksz + self.O):]
            csize = int.from_bytes(k[0:4], "little")
            mask = int.from_bytes(k[4:8], "little")
            offs = tuple(int.from_bytes(k[8 + 8 * j:16 + 8 * j], "little") for j in range(nd - 1))
            child = self._off(k, ksz)
            if level > 0:
                self._chunks(child, nd, out)
            else:
                out.append((offs, csize, mask, child))

    def shape(self, name):
        return self._dataset(name).shape

    def dtype(self, name):
        return self._dataset(name).dtype

    def read(self, name):
        """`h5_file:read(name):all()`: the whole dataset as a C-ordered numpy array (native byte order)."""
        ds = self._dataset(name)
        n = int(np.prod(ds.shape, dtype=np.int64)) if ds.shape else 1
        kind = ds.layout[0]
        if kind == "compact":
            arr = np.frombuffer(ds.layout[1], dtype=ds.dtype, count=n)
        elif kind == "contiguous":
            addr = ds.layout[1]
            if addr == (UNDEF >> (64 - 8 * self.O)):  # never written: fill value (zero)
                arr = np.zeros(n, ds.dtype)
            else:
                self.f.seek(self.base + addr)
                arr = np.fromfile(self.f, dtype=ds.dtype, count=n)
                if arr.size != n:
                    raise H5Error(f"{self.path}: '{name}' is truncated")
        else:
            _, bt, dims = ds.layout
            nd = len(dims)
            cshape = tuple(dims[:-1])
            if len(cshape) != len(ds.shape):
                raise H5Error("chunk rank does not match the dataspace")
            arr = np.zeros(ds.shape, ds.dtype)
            chunks = []
            if bt != (UNDEF >> (64 - 8 * self.O)):
                self._chunks(bt, nd, chunks)
            for offs, csize, mask, caddr in chunks:
                raw = self._at(caddr, csize)
                for idx in range(len(ds.filters) - 1, -1, -1):  # undo the pipeline back to front
                    if mask & (1 << idx):
                        continue
                    fid, cd = ds.filters[idx]
                    if fid == 1:
                        raw = zlib.decompress(raw)
                    elif fid == 2:
                        es = cd[0] if cd else ds.dtype.itemsize
                        raw = np.frombuffer(raw, np.uint8).reshape(es, -1).T.tobytes()
                    elif fid == 3:
                        raw = raw[:-4]
                    else:
                        raise H5Error(f"unsupported filter {fid}")
                c = np.frombuffer(raw, ds.dtype, count=int(np.prod(cshape))).reshape(cshape)
                sl = tuple(slice(o, min(o + s, e)) for o, s, e in zip(offs, cshape, ds.shape))
                arr[sl] = c[tuple(slice(0, s.stop - s.start) for s in sl)]
            return arr.astype(ds.dtype.newbyteorder("="), copy=False)
        return arr.reshape(ds.shape).astype(ds.dtype.newbyteorder("="), copy=False)


def _parse_dtype(b):
    cls = b[0] & 0x0F
    bits0 = b[1]
    size = int.from_bytes(b[4:8], "little")
    order = ">" if bits0 & 1 else "<"
    if cls == 0:
        signed = bool(bits0 & 0x08)
        if size not in (1, 2, 4, 8):
            raise H5Error(f"unsupported integer size {size}")
        return np.dtype(f"{order}{'i' if signed else 'u'}{size}")
    if cls == 1:
        if size not in (2, 4, 8):
            raise H5Error(f"unsupported float size {size}")
        return np.dtype(f"{order}f{size}")
    raise H5Error(f"unsupported datatype class {cls}")


def _parse_filters(b):
    ver, n = b[0], b[1]
    p = 8 if ver == 1 else 2
    out = []
    for _ in range(n):
        fid = int.from_bytes(b[p:p + 2], "little")
        p += 2
        nlen = 0
        if ver == 1 or fid >= 256:
            nlen = int.from_bytes(b[p:p + 2], "little")
            p += 2
        p += 2  # flags
        ncd = int.from_bytes(b[p:p + 2], "little")
        p += 2
        if ver == 1:
            nlen = (nlen + 7) // 8 * 8
        p += nlen
        cd = [int.from_bytes(b[p + 4 * i:p + 4 * i + 4], "little") for i in range(ncd)]
        p += 4 * ncd
        if ver == 1 and ncd % 2:
            p += 4
        out.append((fid, cd))
    return out


def read(path, name):
    with File(path) as f:
        return f.read(name)


# ---------------------------------------------------------------------------------------------
# writer: superblock 0, root group as a symbol table, contiguous little-endian datasets
# ---------------------------------------------------------------------------------------------
_LEAF_K = 16  # symbol-table node holds up to 2 * _LEAF_K entries


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind in "iu":
        bits = 0x08 if dt.kind == "i" else 0x00
        return struct.pack("<B3BI", 0x10, bits, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    if dt == np.float32:
        return struct.pack("<B3BI", 0x11, 0x20, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    if dt == np.float64:
        return struct.pack("<B3BI", 0x11, 0x20, 63, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    raise H5Error(f"cannot write dtype {dt}")


def write(path, datasets):
    """`hdf5.open(path, 'w'); file:write('/name', tensor); ...; file:close()` (001_prepro_img_vgg.lua:156-160).
    datasets: {name: array} of int / uint / float32 / float64 arrays, written contiguous, little-endian."""
    names = sorted(n.strip("/") for n in datasets)
    if not names or len(names) > 2 * _LEAF_K or any("/" in n or not n for n in names):
        raise H5Error("write(): between 1 and 32 datasets in the root group")
    arrays = {}
    for k, v in datasets.items():
        a = np.ascontiguousarray(v)
        if a.dtype.kind == "f" and a.dtype.itemsize not in (4, 8):
            a = a.astype(np.float32)
        arrays[k.strip("/")] = a.astype(a.dtype.newbyteorder("<"), copy=False)

    # local heap data segment: "" at offset 0, then the names
    heap = bytearray(8)
    name_off = {}
    for n in names:
        name_off[n] = len(heap)
        heap += _pad8(n.encode("utf-8") + b"\0")
    if len(heap) < 24:
        heap += b"\0" * (24 - len(heap))

    SB = 96                      # superblock 0 with 8-byte offsets (24 + 4*8 + 40)
    root_hdr = SB                # root object header: 16 + one symbol-table message (8 + 16)
    btree = root_hdr + 16 + 24
    btree_size = 8 + 16 + (2 * 2 * 16 + 1) * 8  # node sized for the superblock's internal K = 16
    heap_hdr = btree + btree_size
    heap_data = heap_hdr + 32
    snod = heap_data + len(heap)
    snod_size = 8 + 2 * _LEAF_K * 40
    pos = snod + snod_size
    hdr_addr, data_addr = {}, {}
    headers = {}
    for n in names:
        a = arrays[n]
        msgs = (_msg(0x01, struct.pack("<BBB5x", 1, a.ndim, 0) + b"".join(struct.pack("<Q", s) for s in a.shape))
                + _msg(0x03, _dtype_msg(a.dtype), flags=1)
                + _msg(0x05, struct.pack("<BBBB", 2, 2, 2, 0))
                + _msg(0x08, struct.pack("<BBQQ", 3, 1, 0, a.nbytes)))  # address patched below
        headers[n] = msgs
        hdr_addr[n] = pos
        pos += 16 + len(msgs)
    for n in names:
        pos = (pos + 7) // 8 * 8
        data_addr[n] = pos
        pos += arrays[n].nbytes
    eof = pos

    out = bytearray()
    out += SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, _LEAF_K, 16, 0)
    out += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    out += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", btree, heap_hdr)  # cached: B-tree, heap
    assert len(out) == SB
    out += struct.pack("<BBHII4x", 1, 0, 1, 1, 24) + _msg(0x11, struct.pack("<QQ", btree, heap_hdr))
    node = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod, name_off[names[-1]])
    out += node + b"\0" * (btree_size - len(node))
    out += b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), 1, heap_data)
    out += heap
    sn = b"SNOD" + struct.pack("<BBH", 1, 0, len(names))
    for n in names:
        sn += struct.pack("<QQII16x", name_off[n], hdr_addr[n], 0, 0)
    out += sn + b"\0" * (snod_size - len(sn))
    for n in names:
        m = bytearray(headers[n])
        lay = m.rfind(struct.pack("<BB", 3, 1) + struct.pack("<Q", 0))
        m[lay + 2:lay + 10] = struct.pack("<Q", data_addr[n])
        assert len(out) == hdr_addr[n]
        out += struct.pack("<BBHII4x", 1, 0, 4, 1, len(m)) + m
    with open(path, "wb") as f:
        f.write(out)
        for n in names:
            f.write(b"\0" * (data_addr[n] - f.tell()))
            arrays[n].tofile(f)
