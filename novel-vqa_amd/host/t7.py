"""Torch7 binary serialisation (torch.save / torch.load, torch/File.lua) for the checkpoint the
reference writes: a Lua table of flat FloatTensors

    torch.save(path, {encoder_w_q=..., embedding_w_q=..., multimodal_w=...})     -- arch1
    torch.save(path, {cnn_w=..., encoder_w_q=..., multimodal_w=...})             -- arch2
    (002_train_vqa_arch1/002_train_baseline.lua:401-402,419-420; 003_.../002_train_baseline.lua:391-399,420-421;
     loaded by 004_eval_model.lua:154-163)

Format (binary mode, little endian, 8-byte longs), object = int32 type tag followed by:
    0 nil | 1 number: float64 | 2 string: int32 n + n bytes | 5 boolean: int32
    3 table: int32 ref-index, then (first occurrence only) int32 n, n x (key object, value object)
    4 torch object: int32 ref-index, then (first occurrence only) string "V 1", string class name and
      Tensor : int32 ndim, int64 size[ndim], int64 stride[ndim], int64 storageOffset (1-based), storage object
      Storage: int64 n, n raw elements
PARITY UNPINNED: the reference ships no .t7 file; this follows Torch7's published File.lua format.
Note that the order of parameters INSIDE encoder_w_q is nngraph's.
"""
import struct

import numpy as np

_TENSOR = {"torch.FloatTensor": np.float32, "torch.DoubleTensor": np.float64, "torch.LongTensor": np.int64,
           "torch.IntTensor": np.int32, "torch.ByteTensor": np.uint8, "torch.CudaTensor": np.float32}
_STORAGE = {k.replace("Tensor", "Storage"): v for k, v in _TENSOR.items()}
_CLASS_OF = {np.dtype(np.float32): "torch.FloatTensor", np.dtype(np.float64): "torch.DoubleTensor",
             np.dtype(np.int64): "torch.LongTensor", np.dtype(np.int32): "torch.IntTensor",
             np.dtype(np.uint8): "torch.ByteTensor"}


class _Writer:
    def __init__(self, f):
        self.f, self.index = f, 0

    def i32(self, v):
        self.f.write(struct.pack("<i", v))

    def i64(self, v):
        self.f.write(struct.pack("<q", v))

    def string(self, s):
        b = s.encode() if isinstance(s, str) else bytes(s)
        self.i32(len(b))
        self.f.write(b)

    def new_index(self):
        self.index += 1
        return self.index

    def obj(self, o):
        if o is None:
            self.i32(0)
        elif isinstance(o, bool):
            self.i32(5)
            self.i32(1 if o else 0)
        elif isinstance(o, (int, float, np.integer, np.floating)):
            self.i32(1)
            self.f.write(struct.pack("<d", float(o)))
        elif isinstance(o, (str, bytes)):
            self.i32(2)
            self.string(o)
        elif isinstance(o, dict):
            self.i32(3)
            self.i32(self.new_index())
            self.i32(len(o))
            for k, v in o.items():
                self.obj(k)
                self.obj(v)
        elif isinstance(o, (list, tuple)):  # Lua array: keys 1..n
            self.obj({i + 1: v for i, v in enumerate(o)})
        elif isinstance(o, np.ndarray):
            a = np.ascontiguousarray(o)
            cls = _CLASS_OF[a.dtype]
            self.i32(4)
            self.i32(self.new_index())
            self.string("V 1")
            self.string(cls)
            self.i32(a.ndim)
            for s in a.shape:
                self.i64(s)
            for s in a.strides:
                self.i64(s // a.itemsize)
            self.i64(1)  # storageOffset, 1-based
            self.i32(4)  # the storage is a torch object of its own
            self.i32(self.new_index())
            self.string("V 1")
            self.string(cls.replace("Tensor", "Storage"))
            self.i64(a.size)
            self.f.write(a.tobytes())
        else:
            raise TypeError(f"cannot serialise {type(o)}")


class _Reader:
    def __init__(self, f):
        self.f, self.objects = f, {}

    def i32(self):
        return struct.unpack("<i", self.f.read(4))[0]

    def i64(self):
        return struct.unpack("<q", self.f.read(8))[0]

    def string(self):
        return self.f.read(self.i32()).decode()

    def obj(self):
        t = self.i32()
        if t == 0:
            return None
        if t == 1:
            return struct.unpack("<d", self.f.read(8))[0]
        if t == 2:
            return self.string()
        if t == 5:
            return self.i32() == 1
        if t == 3:
            idx = self.i32()
            if idx in self.objects:
                return self.objects[idx]
            out = self.objects[idx] = {}
            for _ in range(self.i32()):
                k = self.obj()
                out[int(k) if isinstance(k, float) and k == int(k) else k] = self.obj()
            return out
        if t == 4:
            idx = self.i32()
            if idx in self.objects:
                return self.objects[idx]
            version = self.string()
            cls = self.string() if version.startswith("V ") else version
            if cls in _TENSOR:
                nd = self.i32()
                size = [self.i64() for _ in range(nd)]
                stride = [self.i64() for _ in range(nd)]
                off = self.i64() - 1
                storage = self.obj()
                if storage is None or nd == 0:
                    arr = np.zeros(0, _TENSOR[cls])
                else:
                    arr = np.lib.stride_tricks.as_strided(
                        storage[off:], shape=size, strides=[s * storage.itemsize for s in stride]).copy()
                self.objects[idx] = arr
                return arr
            if cls in _STORAGE:
                n = self.i64()
                dt = np.dtype(_STORAGE[cls])
                arr = np.frombuffer(self.f.read(n * dt.itemsize), dt).copy()
                self.objects[idx] = arr
                return arr
            raise ValueError(f"unsupported torch class {cls!r} (only tensors and storages are read; nothing is executed)")
        raise ValueError(f"unsupported object type tag {t}")


def save(path, obj):
    with open(path, "wb") as f:
        _Writer(f).obj(obj)


def load(path):
    with open(path, "rb") as f:
        return _Reader(f).obj()


SEGMENT_KEYS = {1: ("encoder_w_q", "embedding_w_q", "multimodal_w"), 2: ("cnn_w", "encoder_w_q", "multimodal_w")}


LAYOUT_MARK = "nvqa"   # written under the key 'layout': the LSTM tensors inside encoder_w_q are in include/nvqa_layout.h order


def encoder_permutation(order, R, L, in0):
    """Index permutation that maps an encoder_w_q whose per-layer tensors come in another order onto this library's
    (W_i2h [4R x in], b_i2h [4R], W_h2h [4R x R], b_h2h [4R], layer after layer; in = in0 for layer 0, R above).
    `order`: the foreign order as a sequence of (layer, name) with name in {'w_i2h', 'b_i2h', 'w_h2h', 'b_h2h'},
    e.g. what nngraph's forward-node walk yields once it has been read off a Torch7 installation.  Returns perm with
    ours = foreign[perm] (trailing entries beyond the LSTM tensors, e.g. arch2's lookup table, keep their place).
    PARITY UNPINNED: no nngraph source or reference checkpoint is available offline to fix `order`."""
    sizes = {}
    for l in range(L):
        inn = in0 if l == 0 else R
        sizes[(l, "w_i2h")] = 4 * R * inn
        sizes[(l, "b_i2h")] = 4 * R
        sizes[(l, "w_h2h")] = 4 * R * R
        sizes[(l, "b_h2h")] = 4 * R
    order = [tuple(o) for o in order]
    if sorted(order) != sorted(sizes):
        raise ValueError("order must name every (layer, tensor) exactly once")
    start, off = {}, 0
    for key in order:
        start[key] = off
        off += sizes[key]
    perm = []
    for l in range(L):
        for name in ("w_i2h", "b_i2h", "w_h2h", "b_h2h"):
            perm.append(np.arange(start[(l, name)], start[(l, name)] + sizes[(l, name)], dtype=np.int64))
    return np.concatenate(perm)


def save_checkpoint(path, arch, params, segments, encoder_perm=None):
    """The reference's checkpoint table from the flat parameter vector (reference segment order).  Without
    encoder_perm the table is marked layout = 'nvqa' (this library's order inside encoder_w_q); with one
    (ours = foreign[perm]) the encoder segment is written in the foreign order and no marker is set."""
    params = np.asarray(params, np.float32)
    out, off = {}, 0
    for k, n in zip(SEGMENT_KEYS[arch], segments):
        out[k] = params[off:off + n].copy()
        off += n
    if encoder_perm is not None:
        enc = out["encoder_w_q"]
        foreign = enc.copy()
        foreign[np.asarray(encoder_perm)] = enc[:len(encoder_perm)]
        out["encoder_w_q"] = foreign
    else:
        out["layout"] = LAYOUT_MARK
    save(path, out)


def load_checkpoint(path, arch, segments, encoder_perm=None):
    """Flat parameter vector from a reference checkpoint table (004_eval_model.lua:154-163).  A table without the
    layout = 'nvqa' marker is a foreign (Torch7 / nngraph-ordered) file: loading it as is would put LSTM weights
    into the wrong tensors without any error, so it is refused unless encoder_perm (ours = foreign[perm], see
    encoder_permutation) says how its encoder segment is ordered."""
    t = load(path)
    if t.get("layout") != LAYOUT_MARK and encoder_perm is None:
        raise ValueError("checkpoint has no layout = 'nvqa' marker: the order of the LSTM tensors inside encoder_w_q is "
                         "unknown (nngraph forward-node order); pass encoder_perm = t7.encoder_permutation(...)")
    parts = []
    for k, n in zip(SEGMENT_KEYS[arch], segments):
        a = np.asarray(t[k], np.float32).ravel()
        if a.size != n:
            raise ValueError(f"{k}: checkpoint has {a.size} values, the model wants {n}")
        if k == "encoder_w_q" and encoder_perm is not None:
            a = a.copy()
            a[:len(encoder_perm)] = a[np.asarray(encoder_perm)]
        parts.append(a)
    return np.concatenate(parts)
