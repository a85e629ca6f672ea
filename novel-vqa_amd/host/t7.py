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


def save_checkpoint(path, arch, params, segments):
    """The reference's checkpoint table from the flat parameter vector (reference segment order)."""
    params = np.asarray(params, np.float32)
    out, off = {}, 0
    for k, n in zip(SEGMENT_KEYS[arch], segments):
        out[k] = params[off:off + n].copy()
        off += n
    save(path, out)


def load_checkpoint(path, arch, segments):
    """Flat parameter vector from a reference checkpoint table (004_eval_model.lua:154-163)."""
    t = load(path)
    parts = []
    for k, n in zip(SEGMENT_KEYS[arch], segments):
        a = np.asarray(t[k], np.float32).ravel()
        if a.size != n:
            raise ValueError(f"{k}: checkpoint has {a.size} values, the model wants {n}")
        parts.append(a)
    return np.concatenate(parts)
