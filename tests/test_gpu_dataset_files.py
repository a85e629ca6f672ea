"""Training and evaluation driven from the reference's file formats: data_prepro.h5 /
data_img.h5 / data_prepro.json -> VQAData -> HBM-resident dataset -> device next_batch -> step, checked
against the oracle fed with the same rows gathered on the host."""
import os

import numpy as np
import pytest

from util import gdims, relmax, segment_errors

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")


@pytest.mark.parametrize("arch", [1, 2])
def test_train_and_eval_from_files(pkg, orc, arch):
    data = pkg.dataset.VQAData(os.path.join(HERE, "data_prepro.h5"), os.path.join(HERE, "data_img.h5"),
                               os.path.join(HERE, "data_prepro.json"))
    tr_split, val = data.split("train", arch), data.split("val", arch)
    I = tr_split.fv_im.shape[1]
    kw = dict(arch=arch, B=8, T=26, V=data.vocabulary_size_q, E=16, R=16, L=2 if arch == 1 else 1, I=I, C=24,
              A=data.num_answers)
    if arch == 2:
        kw["E"] = kw["R"]
    d = orc.make_dims(**kw)
    params = orc.synth_params(d)
    tr = pkg.trainer.VQATrainer(gdims(pkg, d), 0, seed=123, dropout=True)
    tr.set_params(params)
    tr.load_dataset(tr_split.question, tr_split.lengths, tr_split.img_list, tr_split.answers, tr_split.fv_im, img_norm=True)
    qinds = tr.next_batch()
    dr = tr._dropout()
    loss = tr.ctx.step_indices(qinds, dr)
    grads = tr.ctx.get_grads()
    fn = tr_split.fv_im / np.sqrt((tr_split.fv_im ** 2).sum(1, keepdims=True))       # 002_train_baseline.lua:117-121
    tok = tr_split.question[qinds] if arch == 1 else tr_split.question[qinds]
    ref = orc.Oracle(np.float64).step(d, params, tok, tr_split.lengths[qinds] if arch == 1 else None,
                                      fn[tr_split.img_list[qinds] - 1].astype(np.float32), tr_split.answers[qinds],
                                      orc.Dropout(dr.mode, dr.p, dr.seed, dr.step))
    assert abs(loss - ref["loss"]) <= 1e-5 * abs(ref["loss"])
    assert max(segment_errors(orc, d, grads, ref["grads"]).values()) < 1e-3
    # validation pass in evaluate mode over the whole val split (ragged last batch), :337-381
    vn = val.fv_im / np.sqrt((val.fv_im ** 2).sum(1, keepdims=True))
    vimg = vn[val.img_list - 1].astype(np.float32)
    scores, pred = tr.predict(val.question, val.lengths if arch == 1 else None, vimg)
    assert scores.shape == (len(val), d.A)
    want = []
    for s in range(0, len(val), d.B):
        n = min(d.B, len(val) - s)
        pad = lambda a: np.concatenate([a[s:s + n], np.repeat(a[s:s + 1], d.B - n, 0)])  # noqa: E731
        ev = orc.Oracle(np.float64).step(d, params, pad(val.question), pad(val.lengths) if arch == 1 else None, pad(vimg),
                                         pad(val.answers), None, train=False)
        want.append(ev["scores"][:n])
    want = np.concatenate(want)
    assert relmax(scores, want) < 1e-4
    assert np.array_equal(pred, want.argmax(1) + 1)
    tr.close()
