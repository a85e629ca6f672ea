"""VQATrainer.init_from_autoencoder: the initialisation of the AE-based training scripts
(002_train_vqa_arch1/003_train_ae_based.lua:65,175-186; _wp :153-160; 003_train_vqa_arch2/003_train_ae_based.lua:150-152,
186-194) from a synthetic .t7 table.  The slicing logic is checked on the CPU against a stand-in context (no GPU here);
tests/test_gpu_variants.py runs the same helper through the library."""
import numpy as np
import pytest


class _FakeCtx:
    """records what the trainer asks of the context: segment sizes of the real layout, a parameter vector"""
    def __init__(self, orc, d):
        self.d = d
        self.seg = [int(n) for n in orc.layout(d)["_segments"]]
        self.x = np.zeros(sum(self.seg), np.float32)

    def segments(self):
        return list(self.seg)

    def init_params(self, seed, lo, hi):
        self.x = np.random.default_rng(seed).uniform(lo, hi, self.x.size).astype(np.float32)
        self.uniform = self.x.copy()

    def get_params(self):
        return self.x.copy()

    def set_params(self, x):
        self.x = np.asarray(x, np.float32).copy()


def _trainer(pkg, orc, d):
    tr = object.__new__(pkg.trainer.VQATrainer)
    tr.dims, tr.seed = d, 7
    tr.ctx = _FakeCtx(orc, d)
    return tr


def _segments(orc, d):
    return [int(n) for n in orc.layout(d)["_segments"]]


@pytest.mark.parametrize("wp", [False, True])
def test_arch1_ae_init(pkg, orc, tmp_path, wp):
    d = orc.make_dims(arch=1, B=4, T=5, V=11, E=8, R=8, L=1, I=12, C=12, A=8)
    tr = _trainer(pkg, orc, d)
    tr.ctx.seg = _segments(orc, d)
    tr.ctx.x = np.zeros(sum(tr.ctx.seg), np.float32)
    rng = np.random.default_rng(0)
    lookup = rng.standard_normal((d.E, d.V + 1)).astype(np.float32)        # LookupTable weight [(V+1) x E], transposed
    enc = rng.standard_normal(tr.ctx.seg[0]).astype(np.float32)
    n_fuse = d.C * 2 * d.R * d.L + d.C + d.C * d.I + d.C
    mm = rng.standard_normal(n_fuse).astype(np.float32)
    path = str(tmp_path / "ae.t7")
    pkg.t7.save(path, {"lookup": lookup, "encoder": enc, "multimodal": mm, "layout": "nvqa"})
    if wp:   # the -variant wp path is netdef.AskipB: an AxB context must be refused
        tr.ctx.fusion = 0
        with pytest.raises(ValueError):
            tr.init_from_autoencoder(path, with_multimodal=True)
        tr.ctx.fusion = 1
    tr.init_from_autoencoder(path, with_multimodal=wp)
    x, (e, m, _) = tr.ctx.x, tr.ctx.seg
    assert np.array_equal(x[:e], enc)
    assert np.array_equal(x[e:e + d.E * d.V].reshape(d.E, d.V), lookup[:, :d.V])   # the AE's START column is dropped
    assert not x[e + d.E * d.V:e + m].any()                                        # embedding bias <- 0
    tail = x[e + m:]
    if wp:
        assert np.array_equal(tail[:n_fuse], mm)
        assert np.array_equal(tail[n_fuse:], tr.ctx.uniform[e + m + n_fuse:])      # classifier stays uniform
    else:
        assert np.array_equal(tail, tr.ctx.uniform[e + m:])
        assert np.abs(tail).max() <= 0.08
    # a table without the layout marker is refused unless the tensor order is given
    pkg.t7.save(path, {"lookup": lookup, "encoder": enc})
    with pytest.raises(ValueError):
        tr.init_from_autoencoder(path)
    perm = pkg.t7.encoder_permutation([(0, "w_h2h"), (0, "b_h2h"), (0, "w_i2h"), (0, "b_i2h")], d.R, d.L, d.E)
    tr.init_from_autoencoder(path, encoder_perm=perm)
    assert np.array_equal(tr.ctx.x[:e], enc[perm])
    with pytest.raises(ValueError):   # wrong vocabulary
        pkg.t7.save(path, {"lookup": lookup[:, :-2], "encoder": enc, "layout": "nvqa"})
        tr.init_from_autoencoder(path)


def test_arch2_ae_init(pkg, orc, tmp_path):
    d = orc.make_dims(arch=2, B=4, T=5, V=11, E=8, R=8, L=2, I=12, C=4, A=8)
    tr = _trainer(pkg, orc, d)
    tr.ctx.seg = _segments(orc, d)
    tr.ctx.x = np.zeros(sum(tr.ctx.seg), np.float32)
    rng = np.random.default_rng(1)
    lk = rng.standard_normal((d.V + 1, d.E)).astype(np.float32)
    n_lstm = tr.ctx.seg[1] - lk.size
    enc = rng.standard_normal(n_lstm).astype(np.float32)
    path = str(tmp_path / "ae2.t7")
    pkg.t7.save(path, {"lookup_table": lk, "encoder": enc, "layout": "nvqa"})
    tr.init_from_autoencoder(path)
    x, (c, e, m) = tr.ctx.x, tr.ctx.seg
    assert np.array_equal(x[:c], tr.ctx.uniform[:c])                 # cnn_w uniform
    assert np.array_equal(x[c:c + n_lstm], enc)
    assert np.array_equal(x[c + n_lstm:c + e].reshape(d.V + 1, d.E), lk)
    assert np.array_equal(x[c + e:], tr.ctx.uniform[c + e:])         # multimodal_w uniform
