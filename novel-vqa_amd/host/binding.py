"""ctypes binding of libnvqa.so (include/nvqa.h) -- the executable twin of
novel-vqa_amd/lua/nvqa_ffi.lua (LuaJIT is not available in the build image).

There is NO fallback: if the shared library or a HIP device is missing the
calls raise NvqaError (the product path never routes through the CPU oracle).
"""
import ctypes
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "libnvqa.so")

COMM_ID_BYTES = 128


class NvqaError(RuntimeError):
    pass


class Dims(ctypes.Structure):
    """nvqa_dims"""

    _fields_ = [(n, ctypes.c_int32) for n in ("arch", "B", "T", "V", "E", "R", "L", "I", "C", "A")]


class Dropout(ctypes.Structure):
    """nvqa_dropout"""

    _fields_ = [("mode", ctypes.c_int32), ("p", ctypes.c_float), ("seed", ctypes.c_uint64),
                ("step", ctypes.c_uint64)]


# every symbol include/nvqa.h declares: name -> (restype, argtypes)
_vp, _i32p, _f32p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
SYMBOLS = {
    "nvqa_create": (ctypes.c_int, [ctypes.POINTER(Dims), ctypes.c_int, ctypes.POINTER(_vp)]),
    "nvqa_destroy": (ctypes.c_int, [_vp]),
    "nvqa_last_error": (ctypes.c_char_p, []),
    "nvqa_sync": (ctypes.c_int, [_vp]),
    "nvqa_param_count": (ctypes.c_size_t, [_vp]),
    "nvqa_segments": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_size_t)]),
    "nvqa_init_params": (ctypes.c_int, [_vp, ctypes.c_uint64, ctypes.c_float, ctypes.c_float]),
    "nvqa_set_params": (ctypes.c_int, [_vp, _f32p]),
    "nvqa_get_params": (ctypes.c_int, [_vp, _f32p]),
    "nvqa_get_grads": (ctypes.c_int, [_vp, _f32p, ctypes.c_float]),
    "nvqa_step": (ctypes.c_int, [_vp, _i32p, _i32p, _f32p, _i32p, ctypes.POINTER(Dropout), _f32p]),
    "nvqa_get_loss": (ctypes.c_int, [_vp, _f32p]),
    "nvqa_forward": (ctypes.c_int, [_vp, ctypes.c_int32, _i32p, _i32p, _f32p, _f32p, _i32p]),
    "nvqa_evaluate": (ctypes.c_int, [_vp, ctypes.c_int32, _i32p, _i32p, _f32p, _i32p, _i32p, ctypes.c_int32, _f32p, _i32p,
                                     _i32p, _f32p]),
    "nvqa_rmsprop_update": (ctypes.c_int, [_vp] + [ctypes.c_float] * 5),
    "nvqa_set_fusion": (ctypes.c_int, [_vp, ctypes.c_int]),
    "nvqa_set_precision": (ctypes.c_int, [_vp, ctypes.c_int]),
    "nvqa_set_ref_quirks": (ctypes.c_int, [_vp, ctypes.c_int]),
    "nvqa_param_norms": (ctypes.c_int, [_vp, _f32p]),
    "nvqa_persistent_state": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int)]),
    "nvqa_set_grad_scales": (ctypes.c_int, [_vp, _f32p]),
    "nvqa_dataset_load": (ctypes.c_int, [_vp, ctypes.c_int64, _i32p, _i32p, _i32p, _i32p,
                                         ctypes.c_int64, _f32p, ctypes.c_int]),
    "nvqa_step_indices": (ctypes.c_int, [_vp, _i64p, ctypes.POINTER(Dropout), _f32p]),
    "nvqa_comm_unique_id": (ctypes.c_int, [_vp]),
    "nvqa_comm_init": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp]),
    "nvqa_comm_library": (ctypes.c_char_p, []),
    "nvqa_vgg16_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    "nvqa_vgg16_destroy": (ctypes.c_int, [_vp]),
    "nvqa_vgg16_weight_count": (ctypes.c_size_t, [_vp]),
    "nvqa_vgg16_feature_dim": (ctypes.c_int, [_vp]),
    "nvqa_vgg16_set_weights": (ctypes.c_int, [_vp, _f32p]),
    "nvqa_vgg16_fc7": (ctypes.c_int, [_vp, _f32p, ctypes.c_int, _f32p]),
    "nvqa_vgg16_set_precision": (ctypes.c_int, [_vp, ctypes.c_int]),
    "nvqa_vgg16_preprocess": (ctypes.c_int, [_vp, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p]),
    "nvqa_step_images": (ctypes.c_int, [_vp, _vp, _f32p, _i32p, _i32p, _i32p, ctypes.POINTER(Dropout), _f32p]),
    "nvqa_profile_enable": (ctypes.c_int, [_vp, ctypes.c_int]),
    "nvqa_profile_reset": (ctypes.c_int, [_vp]),
    "nvqa_profile_count": (ctypes.c_int, [_vp]),
    "nvqa_profile_name": (ctypes.c_char_p, [_vp, ctypes.c_int]),
    "nvqa_profile_get": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_int64),
                                        ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_double)]),
}

_lib = None


def load_library(path=None):
    """dlopen libnvqa.so and bind every declared symbol. Raises NvqaError if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise NvqaError(f"{p} not found: build it with __graft_entry__.build() "
                        f"(make -C novel-vqa_amd); there is no CPU fallback")
    lib = ctypes.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _f32(a):
    return None if a is None else a.ctypes.data_as(_f32p)


def _i32(a):
    return None if a is None else a.ctypes.data_as(_i32p)


class Context:
    """Owns one nvqa_ctx (one device)."""

    def __init__(self, dims, device=0):
        self.lib = load_library()
        self.dims = dims
        self.fusion = 0  # netdef.AxB until set_fusion says otherwise
        self._h = _vp()
        self._check(self.lib.nvqa_create(ctypes.byref(dims), device, ctypes.byref(self._h)))

    def _check(self, rc):
        if rc != 0:
            raise NvqaError(f"libnvqa error {rc}: {self.lib.nvqa_last_error().decode()}")

    def close(self):
        if self._h:
            self.lib.nvqa_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters -------------------------------------------------------
    @property
    def param_count(self):
        return int(self.lib.nvqa_param_count(self._h))

    def segments(self):
        out = (ctypes.c_size_t * 3)()
        self._check(self.lib.nvqa_segments(self._h, out))
        return tuple(int(x) for x in out)

    def init_params(self, seed=123, lo=-0.08, hi=0.08):
        self._check(self.lib.nvqa_init_params(self._h, seed, lo, hi))

    def set_params(self, params):
        p = np.ascontiguousarray(params, np.float32)
        assert p.size == self.param_count
        self._check(self.lib.nvqa_set_params(self._h, _f32(p)))

    def get_params(self):
        out = np.empty(self.param_count, np.float32)
        self._check(self.lib.nvqa_get_params(self._h, _f32(out)))
        return out

    def get_grads(self, clamp=0.0):
        out = np.empty(self.param_count, np.float32)
        self._check(self.lib.nvqa_get_grads(self._h, _f32(out), clamp))
        return out

    # ---- hot path ------------------------------------------------------------
    def step(self, tokens, lengths, img, labels, dropout=None, want_loss=True):
        d = self.dims
        tokens = np.ascontiguousarray(tokens, np.int32).reshape(d.B, d.T)
        lengths = None if lengths is None else np.ascontiguousarray(lengths, np.int32)
        img = np.ascontiguousarray(img, np.float32).reshape(d.B, d.I)
        labels = np.ascontiguousarray(labels, np.int32)
        loss = ctypes.c_float(0)
        dr = ctypes.byref(dropout) if dropout is not None else None
        self._check(self.lib.nvqa_step(self._h, _i32(tokens), _i32(lengths), _f32(img), _i32(labels),
                                       dr, ctypes.byref(loss) if want_loss else None))
        return float(loss.value) if want_loss else None

    def step_indices(self, qinds, dropout=None, want_loss=True):
        q = np.ascontiguousarray(qinds, np.int64)
        assert q.size == self.dims.B
        loss = ctypes.c_float(0)
        dr = ctypes.byref(dropout) if dropout is not None else None
        self._check(self.lib.nvqa_step_indices(self._h, q.ctypes.data_as(_i64p), dr,
                                               ctypes.byref(loss) if want_loss else None))
        return float(loss.value) if want_loss else None

    def step_images(self, vgg, images, tokens, lengths, labels, dropout=None):
        """Extractor + training step in one call (features stay on the device)."""
        d = self.dims
        images = np.ascontiguousarray(images, np.float32)
        tokens = np.ascontiguousarray(tokens, np.int32).reshape(d.B, d.T)
        lengths = None if lengths is None else np.ascontiguousarray(lengths, np.int32)
        labels = np.ascontiguousarray(labels, np.int32)
        loss = ctypes.c_float(0)
        dr = ctypes.byref(dropout) if dropout is not None else None
        self._check(self.lib.nvqa_step_images(self._h, vgg._h, _f32(images), _i32(tokens), _i32(lengths),
                                              _i32(labels), dr, ctypes.byref(loss)))
        return float(loss.value)

    def get_loss(self):
        loss = ctypes.c_float(0)
        self._check(self.lib.nvqa_get_loss(self._h, ctypes.byref(loss)))
        return float(loss.value)

    def forward(self, tokens, lengths, img):
        d = self.dims
        tokens = np.ascontiguousarray(tokens, np.int32)
        n = tokens.shape[0]
        lengths = None if lengths is None else np.ascontiguousarray(lengths, np.int32)
        img = np.ascontiguousarray(img, np.float32)
        scores = np.empty((n, d.A), np.float32)
        argmax = np.empty(n, np.int32)
        self._check(self.lib.nvqa_forward(self._h, n, _i32(tokens), _i32(lengths), _f32(img),
                                          _f32(scores), _i32(argmax)))
        return scores, argmax

    def evaluate(self, tokens, lengths, img, labels=None, mc_ans=None):
        """nvqa_evaluate: dict(scores, argmax, loss (if labels), mc_argmax (if mc_ans [n, n_mc], 0 = empty slot))."""
        d = self.dims
        tokens = np.ascontiguousarray(tokens, np.int32)
        n = tokens.shape[0]
        lengths = None if lengths is None else np.ascontiguousarray(lengths, np.int32)
        img = np.ascontiguousarray(img, np.float32)
        labels = None if labels is None else np.ascontiguousarray(labels, np.int32)
        mc = None if mc_ans is None else np.ascontiguousarray(mc_ans, np.int32)
        scores = np.empty((n, d.A), np.float32)
        argmax = np.empty(n, np.int32)
        mcout = np.empty(n, np.int32) if mc is not None else None
        loss = ctypes.c_float(0)
        self._check(self.lib.nvqa_evaluate(self._h, n, _i32(tokens), _i32(lengths), _f32(img), _i32(labels), _i32(mc),
                                           0 if mc is None else mc.shape[1], _f32(scores), _i32(argmax), _i32(mcout),
                                           ctypes.byref(loss) if labels is not None else None))
        out = {"scores": scores, "argmax": argmax}
        if labels is not None:
            out["loss"] = float(loss.value)
        if mc is not None:
            out["mc_argmax"] = mcout
        return out

    def rmsprop_update(self, lr, alpha=0.99, eps=1e-8, wd=0.0, clamp=10.0):
        self._check(self.lib.nvqa_rmsprop_update(self._h, lr, alpha, eps, wd, clamp))

    def set_fusion(self, mode):
        """0 netdef.AxB, 1 netdef.AskipB, 2 netdef.A_B (W_o becomes [A x 2C]: right after creation only)."""
        self._check(self.lib.nvqa_set_fusion(self._h, int(mode)))
        self.fusion = int(mode)

    QUIRK_H0, QUIRK_LOOKUP = 1, 2

    def set_ref_quirks(self, flags):
        """arch2: reproduce the reference's aliased-h0 / untrained-lookup artefacts (include/nvqa.h)."""
        self._check(self.lib.nvqa_set_ref_quirks(self._h, int(flags)))

    def set_precision(self, bf16):
        self._check(self.lib.nvqa_set_precision(self._h, int(bool(bf16))))

    def set_grad_scales(self, scales):
        a = np.ascontiguousarray(scales, np.float32)
        assert a.size == 3
        self._check(self.lib.nvqa_set_grad_scales(self._h, _f32(a)))

    def sync(self):
        self._check(self.lib.nvqa_sync(self._h))

    def param_norms(self):
        """torch.norm of the three parameter segments (arch2's training log line, 002_train_baseline.lua:400-407)"""
        out = np.empty(3, np.float32)
        self._check(self.lib.nvqa_param_norms(self._h, _f32(out)))
        return out

    def persistent_state(self):
        """{'fwd': bool, 'bwd': bool}: does the next step run the LSTM unroll / the BPTT as one persistent launch?"""
        out = (ctypes.c_int * 2)()
        self._check(self.lib.nvqa_persistent_state(self._h, out))
        return {"fwd": bool(out[0]), "bwd": bool(out[1])}

    # ---- dataset ------------------------------------------------------------------
    def dataset_load(self, questions, lengths, img_pos, answers, feats, l2_normalize=False):
        q = np.ascontiguousarray(questions, np.int32)
        l = None if lengths is None else np.ascontiguousarray(lengths, np.int32)
        ip = np.ascontiguousarray(img_pos, np.int32)
        an = np.ascontiguousarray(answers, np.int32)
        f = np.ascontiguousarray(feats, np.float32)
        self._check(self.lib.nvqa_dataset_load(self._h, q.shape[0], _i32(q), _i32(l), _i32(ip), _i32(an),
                                               f.shape[0], _f32(f), int(l2_normalize)))

    # ---- data parallel ------------------------------------------------------------
    def comm_unique_id(self):
        buf = ctypes.create_string_buffer(COMM_ID_BYTES)
        self._check(self.lib.nvqa_comm_unique_id(ctypes.cast(buf, _vp)))
        return buf.raw

    def comm_library(self):
        """path of the collective library behind nvqa_comm_* (NVQA_RCCL_LIB, or the librccl beside the mapped HIP runtime)"""
        p = self.lib.nvqa_comm_library()
        if p is None:
            raise NvqaError(f"libnvqa: {self.lib.nvqa_last_error().decode()}")
        return p.decode()

    def comm_init(self, rank, world, comm_id):
        buf = ctypes.create_string_buffer(bytes(comm_id), COMM_ID_BYTES)
        self._check(self.lib.nvqa_comm_init(self._h, rank, world, ctypes.cast(buf, _vp)))

    # ---- measurement ----------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self.lib.nvqa_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._check(self.lib.nvqa_profile_reset(self._h))

    def profile(self):
        out = {}
        for i in range(self.lib.nvqa_profile_count(self._h)):
            ms, n, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            self._check(self.lib.nvqa_profile_get(self._h, i, ctypes.byref(ms), ctypes.byref(n),
                                                  ctypes.byref(fl), ctypes.byref(by)))
            out[self.lib.nvqa_profile_name(self._h, i).decode()] = {
                "ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
        return out


class Vgg16:
    """VGG-16 fc7 extractor (001_prepro_img_vgg.lua): owns one nvqa_vgg."""

    def __init__(self, device=0, width_div=1, input_hw=224, max_batch=16):
        self.lib = load_library()
        self._h = _vp()
        self.hw = input_hw
        self._check(self.lib.nvqa_vgg16_create(device, width_div, input_hw, max_batch, ctypes.byref(self._h)))
        self.weight_count = int(self.lib.nvqa_vgg16_weight_count(self._h))
        self.feature_dim = int(self.lib.nvqa_vgg16_feature_dim(self._h))

    def _check(self, rc):
        if rc != 0:
            raise NvqaError(f"libnvqa error {rc}: {self.lib.nvqa_last_error().decode()}")

    def set_weights(self, flat):
        w = np.ascontiguousarray(flat, np.float32)
        assert w.size == self.weight_count
        self._check(self.lib.nvqa_vgg16_set_weights(self._h, _f32(w)))

    def set_precision(self, bf16):
        """1 = operands of the convolutions / fc products rounded to bf16 (f32 accumulate, f32 features)"""
        self._check(self.lib.nvqa_vgg16_set_precision(self._h, int(bool(bf16))))

    def fc7(self, images):
        x = np.ascontiguousarray(images, np.float32)
        n = x.shape[0]
        assert x.shape[1:] == (3, self.hw, self.hw)
        out = np.empty((n, self.feature_dim), np.float32)
        self._check(self.lib.nvqa_vgg16_fc7(self._h, _f32(x), n, _f32(out)))
        return out

    def preprocess(self, rgb):
        x = np.ascontiguousarray(rgb, np.float32)
        n, _, H, W = x.shape
        out = np.empty((n, 3, self.hw, self.hw), np.float32)
        self._check(self.lib.nvqa_vgg16_preprocess(self._h, _f32(x), n, H, W, _f32(out)))
        return out

    def close(self):
        if self._h:
            self.lib.nvqa_vgg16_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
