"""Edge shapes through the C ABI (the GEMM tiles are 64-128 wide; everything here is smaller than a
tile, or degenerate): batch of one, one time step, one token per question, four LSTM layers,
all-null arch2 questions, a short eval batch."""
import numpy as np
import pytest

from util import gdims, gdrop, relmax, segment_errors

pytestmark = pytest.mark.gpu


def _check(pkg, orc, d, tok, lens, img, lab, mode=1):
    params = orc.synth_params(d)
    dr = orc.Dropout(mode, 0.5, 123, 2)
    ref = orc.Oracle(np.float64).step(d, params, tok, lens, img, lab, dr)
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    loss = ctx.step(tok, lens, img, lab, gdrop(pkg, dr))
    assert abs(loss - ref["loss"]) <= 1e-5 * abs(ref["loss"]), (loss, ref["loss"])
    bad = {k: e for k, e in segment_errors(orc, d, ctx.get_grads(), ref["grads"]).items()
           if e > 1e-3 and np.abs(ref["grads"][orc.layout(d)[k][0]:sum(orc.layout(d)[k])]).max() > 1e-12}
    assert not bad, bad
    ctx.close()


@pytest.mark.parametrize("kw", [
    dict(arch=1, B=1, T=5, V=9, E=8, R=8, L=2, I=8, C=8, A=4),
    dict(arch=1, B=3, T=1, V=9, E=8, R=8, L=2, I=8, C=8, A=4),
    dict(arch=1, B=70, T=4, V=9, E=4, R=4, L=4, I=4, C=4, A=4),
    dict(arch=2, B=1, T=3, V=9, E=8, R=8, L=1, I=8, C=4, A=4),
    dict(arch=2, B=5, T=4, V=9, E=8, R=12, L=4, I=8, C=4, A=8),
])
def test_small_and_deep(pkg, orc, kw):
    d = orc.make_dims(**kw)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    _check(pkg, orc, d, tok, lens if d.arch == 1 else None, img, lab)


def test_one_token_questions(pkg, orc):
    d = orc.make_dims(arch=1, B=9, T=6, V=20, E=8, R=8, L=2, I=8, C=8, A=4)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    lens[:] = 1
    tok[:, :-1] = 0
    tok[:, -1] = np.arange(1, d.B + 1)
    _check(pkg, orc, d, tok, lens, img, lab)


def test_arch2_all_null_questions(pkg, orc):
    # every sequence empty: the encoder sees the image and START only (tmax = 2)
    d = orc.make_dims(arch=2, B=4, T=5, V=9, E=8, R=8, L=2, I=8, C=4, A=4)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    tok[:] = 0
    _check(pkg, orc, d, tok, None, img, lab)


def test_short_eval_batch(pkg, orc):
    d = orc.make_dims(arch=1, B=16, T=6, V=20, E=8, R=8, L=2, I=8, C=8, A=4)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    ev = orc.Oracle(np.float64).step(d, params, tok, lens, img, lab, None, train=False)
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    scores, argmax = ctx.forward(tok[:5], lens[:5], img[:5])   # n < B rows (last batch of a split)
    assert scores.shape == (5, d.A) and relmax(scores, ev["scores"][:5]) < 1e-4
    ctx.close()


def test_create_failure_frees_the_context_and_bounds_are_checked(pkg, orc, monkeypatch):
    """nvqa_create must not leak when an allocation fails half-way (NVQA_FAIL_ALLOC injects the n-th hipMalloc
    failure), and dims that the single-workgroup assembly kernels cannot hold are rejected with a message."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")  # the runtime libnvqa itself is linked against (already loaded)

    def free_bytes():
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    d = orc.make_dims(arch=1, B=64, T=8, V=50, E=64, R=256, L=2, I=1024, C=256, A=64)
    ctx = pkg.binding.Context(gdims(pkg, d), 0)   # warm: allocator pools exist
    ctx.close()
    free0 = free_bytes()
    for n in (0, 3, 11, 19, 27):
        monkeypatch.setenv("NVQA_FAIL_ALLOC", str(n))
        with pytest.raises(pkg.binding.NvqaError, match="NVQA_FAIL_ALLOC"):
            pkg.binding.Context(gdims(pkg, d), 0)
    monkeypatch.delenv("NVQA_FAIL_ALLOC")
    assert free0 - free_bytes() < (8 << 20), "device memory leaked by the failed creates"
    ctx = pkg.binding.Context(gdims(pkg, d), 0)   # and the library still works
    ctx.close()
    with pytest.raises(pkg.binding.NvqaError, match="arch2: T="):
        pkg.binding.Context(gdims(pkg, orc.make_dims(arch=2, B=4, T=300, V=9, E=8, R=8, L=1, I=8, C=4, A=4)), 0)
    with pytest.raises(pkg.binding.NvqaError, match="k_sort_lengths"):
        pkg.binding.Context(gdims(pkg, orc.make_dims(arch=1, B=8, T=1000, V=9, E=8, R=8, L=1, I=8, C=8, A=4)), 0)


def test_length_sort_of_a_batch_larger_than_its_workgroup(pkg, orc):
    """k_sort_lengths is one workgroup of 1024 threads; a batch of 2100 rows takes three passes, and the stable rank of a row
    (rows of its length in earlier passes + in earlier waves of its pass + in lower lanes of its wave) must still give the
    order of sort_encoding_onehot_right_align (misc/RNNUtils.lua:84-103): a full step against the oracle, ragged lengths."""
    from util import assert_grads
    d = orc.make_dims(arch=1, B=2100, T=9, V=50, E=16, R=16, L=2, I=32, C=24, A=12)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, seed=8, full_length=False, min_len=1)
    dr = orc.Dropout(1, 0.5, 123, 5)
    ref = orc.Oracle(np.float64).step(d, params, tok, lens, img, lab, dr)
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    loss = ctx.step(tok, lens, img, lab, gdrop(pkg, dr))
    assert abs(loss - ref["loss"]) <= 2e-6 * abs(ref["loss"])
    assert_grads(orc, d, ctx.get_grads(), ref["grads"], 2e-5, "sort_three_passes")
    ctx.close()


@pytest.mark.parametrize("B", [40, 800])
def test_dataset_route_sample_ids_as_arguments_and_in_device_memory(pkg, orc, B):
    """nvqa_step_indices hands the sample ids to k_gather_batch as kernel arguments (B <= 768: no H2D blit in front of the
    step) or, for larger batches, through device memory: both must gather what the host-batch entry is given by hand
    (dataset:next_batch, 002_train_baseline.lua:202-219), repeated ids included."""
    d = orc.make_dims(arch=1, B=B, T=7, V=50, E=16, R=16, L=1, I=32, C=24, A=12)
    params = orc.synth_params(d)
    rng = np.random.default_rng(5)
    nq, nimg = 3 * B + 7, 61
    big = orc.make_dims(**{**{n: getattr(d, n) for n, _ in d._fields_}, "B": nq})
    tok, lens, _, lab = orc.synth_batch(big, seed=4, full_length=False, min_len=1)
    feats = np.abs(rng.standard_normal((nimg, d.I))).astype(np.float32)
    img_pos = rng.integers(1, nimg + 1, nq).astype(np.int32)
    tr = pkg.trainer.VQATrainer(gdims(pkg, d), 0, seed=123, dropout=True)
    tr.set_params(params)
    tr.load_dataset(tok, lens, img_pos, lab, feats, img_norm=True)
    qinds = rng.integers(0, nq, B).astype(np.int64)     # with replacement, like torch.random
    dr = tr._dropout()
    la = tr.ctx.step_indices(qinds, dr)
    ga = tr.ctx.get_grads()
    fn = feats / np.sqrt((feats * feats).sum(1, keepdims=True))
    lb = tr.ctx.step(tok[qinds], lens[qinds], fn[img_pos[qinds] - 1], lab[qinds], dr)
    gb = tr.ctx.get_grads()
    assert abs(la - lb) <= 1e-6 * abs(lb) and relmax(ga, gb) < 1e-5
    tr.close()
