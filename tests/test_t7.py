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
        assert sorted(t) == sorted(pkg.t7.SEGMENT_KEYS[arch])
        assert np.array_equal(pkg.t7.load_checkpoint(str(q), arch, seg), x)
