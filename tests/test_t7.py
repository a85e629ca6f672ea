"""Torch7 .t7 serialisation of the reference's checkpoint table (host/t7.py).  The reference ships
no .t7 file (PARITY UNPINNED); the byte-level expectation below is written out from Torch7's
File.lua format by hand."""
import struct

import numpy as np


def test_bytes_of_a_small_table(pkg, tmp_path):
    p = tmp_path / "a.t7"
    pkg.t7.save(str(p), {"w": np.array([1.5, -2.0], np.float32)})
    exp = b"".join([
        struct.pack("<i", 3), struct.pack("<i", 1), struct.pack("<i", 1),          # table, ref 1, 1 pair
        struct.pack("<i", 2), struct.pack("<i", 1), b"w",                          # key: string "w"
        struct.pack("<i", 4), struct.pack("<i", 2),                                # torch object, ref 2
        struct.pack("<i", 3), b"V 1", struct.pack("<i", 17), b"torch.FloatTensor",
        struct.pack("<i", 1), struct.pack("<q", 2), struct.pack("<q", 1), struct.pack("<q", 1),  # ndim, size, stride, offset
        struct.pack("<i", 4), struct.pack("<i", 3),                                # its storage, ref 3
        struct.pack("<i", 3), b"V 1", struct.pack("<i", 18), b"torch.FloatStorage",
        struct.pack("<q", 2), struct.pack("<ff", 1.5, -2.0),
    ])
    assert p.read_bytes() == exp


def test_round_trip_and_checkpoint(pkg, tmp_path):
    rng = np.random.default_rng(0)
    obj = {"encoder_w_q": rng.standard_normal(37).astype(np.float32), "n": 3.0, "flag": True, "name": "lstm",
           "nested": {1: rng.standard_normal((2, 3)).astype(np.float64), 2: np.arange(5, dtype=np.int64)}}
    p = tmp_path / "b.t7"
    pkg.t7.save(str(p), obj)
    back = pkg.t7.load(str(p))
    assert np.array_equal(back["encoder_w_q"], obj["encoder_w_q"]) and back["n"] == 3.0 and back["flag"] is True
    assert back["name"] == "lstm" and np.array_equal(back["nested"][1], obj["nested"][1])
    assert np.array_equal(back["nested"][2], obj["nested"][2])
    seg = (5, 7, 11)
    x = rng.standard_normal(sum(seg)).astype(np.float32)
    for arch in (1, 2):
        q = tmp_path / f"ck{arch}.t7"
        pkg.t7.save_checkpoint(str(q), arch, x, seg)
        t = pkg.t7.load(str(q))
        assert sorted(t) == sorted(pkg.t7.SEGMENT_KEYS[arch] + ("layout",)) and t["layout"] == "nvqa"
        assert np.array_equal(pkg.t7.load_checkpoint(str(q), arch, seg), x)


def test_foreign_checkpoint_needs_a_permutation(pkg, orc, tmp_path):
    """A table without the layout marker (what Torch7 itself writes) is refused; with the order of its LSTM tensors
    given, encoder_w_q is permuted into this library's order -- and written back in the foreign order."""
    import pytest
    R, L, E = 4, 2, 8
    d = orc.make_dims(arch=1, B=2, T=3, V=5, E=E, R=R, L=L, I=4, C=4, A=4)
    lo = orc.layout(d)
    segs = lo["_segments"]
    ours = np.arange(lo["_total"], dtype=np.float32)
    # a foreign file: per layer h2h before i2h, weights before biases (a stand-in for nngraph's node order)
    order = [(l, n) for l in range(L) for n in ("w_h2h", "w_i2h", "b_h2h", "b_i2h")]
    enc = ours[:segs[0]]
    pieces = {(l, n): enc[lo[f"{n}{l}"][0]:lo[f"{n}{l}"][0] + lo[f"{n}{l}"][1]] for l in range(L) for n in ("w_i2h", "b_i2h", "w_h2h", "b_h2h")}
    foreign_enc = np.concatenate([pieces[k] for k in order])
    p = tmp_path / "torch7.t7"
    pkg.t7.save(str(p), {"encoder_w_q": foreign_enc, "embedding_w_q": ours[segs[0]:segs[0] + segs[1]],
                          "multimodal_w": ours[segs[0] + segs[1]:]})
    with pytest.raises(ValueError, match="layout"):
        pkg.t7.load_checkpoint(str(p), 1, segs)
    perm = pkg.t7.encoder_permutation(order, R, L, E)
    assert np.array_equal(pkg.t7.load_checkpoint(str(p), 1, segs, encoder_perm=perm), ours)
    q = tmp_path / "back.t7"
    pkg.t7.save_checkpoint(str(q), 1, ours, segs, encoder_perm=perm)
    t = pkg.t7.load(str(q))
    assert "layout" not in t and np.array_equal(t["encoder_w_q"], foreign_enc)
