"""The reference's data files: data_prepro.h5 / data_img.h5 / data_prepro.json.

tests/golden/h5/*.h5 were written by libhdf5 through h5py with the reference's own create_dataset calls
(000_prepro_vqa.py:273-300) by tests/golden/make_h5_fixtures.py; expected.npz holds the arrays that went in.
The package's reader parses the format from the specification, so these files pin it."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")
H5PY_PYTHON = "/opt/conda/bin/python3.9"  # the one interpreter of the build image that has h5py


@pytest.fixture(scope="module")
def exp():
    return np.load(os.path.join(HERE, "expected.npz"))


def test_reads_h5py_written_question_file(pkg, exp):
    with pkg.h5.File(os.path.join(HERE, "data_prepro.h5")) as f:
        names = f.keys()
        assert len(names) == 15 and "ques_train" in names and "MC_ans_test" in names
        for k in names:
            a = f.read("/" + k)
            assert a.dtype == np.uint32 and np.array_equal(a, exp[k]), k
        assert f.shape("ques_train") == (37, 26)


def test_reads_feature_file(pkg, exp):
    for split in ("train", "val", "test"):
        a = pkg.h5.read(os.path.join(HERE, "data_img.h5"), "/images_" + split)
        assert a.dtype == np.float32 and np.array_equal(a, exp["images_" + split])


def test_layout_and_type_variants(pkg, exp):
    with pkg.h5.File(os.path.join(HERE, "variants.h5")) as f:
        for k in ("chunked", "gzip", "gzip_shuffle_f32"):          # chunk B-tree, deflate, shuffle, fletcher32
            assert np.array_equal(f.read(k), exp["v_chunked"]), k
        assert np.array_equal(f.read("be_i32"), exp["v_be_i32"]) and f.read("be_i32").dtype == np.int32
        assert np.array_equal(f.read("i64"), exp["v_be_i32"])
        assert np.array_equal(f.read("f64"), exp["v_chunked"].astype(np.float64))
        assert np.array_equal(f.read("u8"), exp["v_u8"])
        assert np.array_equal(f.read("/grp/inner"), exp["v_inner"]) and f.keys("/grp") == ["inner"]
        assert np.array_equal(f.read("never_written"), np.zeros((4, 3), np.float32))   # no storage allocated
        assert np.array_equal(f.read("many_chunks"), exp["v_many_chunks"])               # multi-level chunk B-tree, edge chunks
        with pytest.raises(KeyError):
            f.read("/nope")
    with pkg.h5.File(os.path.join(HERE, "latest.h5")) as f:                              # superblock 3, OHDR v2, link messages
        assert f.keys() == ["a", "b"]
        assert np.array_equal(f.read("a"), exp["v_chunked"]) and np.array_equal(f.read("b"), exp["v_be_i32"])


def test_rejects_non_hdf5(pkg, tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file" * 10)
    with pytest.raises(pkg.h5.H5Error):
        pkg.h5.File(str(p))


def test_writer_round_trip_and_libhdf5_reads_it(pkg, exp, tmp_path):
    # the extractor's output file (001_prepro_img_vgg.lua:156-160)
    p = str(tmp_path / "data_img_vgg.h5")
    pkg.dataset.write_features(p, exp["images_train"], exp["images_val"], exp["images_test"])
    with pkg.h5.File(p) as f:
        assert f.keys() == ["images_test", "images_train", "images_val"]
        for s in ("train", "val", "test"):
            assert np.array_equal(f.read("images_" + s), exp["images_" + s])
    if not os.path.exists(H5PY_PYTHON):
        pytest.skip("no interpreter with h5py on this machine: libhdf5 cross-check not run")
    code = ("import h5py, numpy as np, sys; f = h5py.File(sys.argv[1], 'r'); e = np.load(sys.argv[2]);"
            "assert sorted(f.keys()) == ['images_test', 'images_train', 'images_val'];"
            "assert all(np.array_equal(f['images_' + s][...], e['images_' + s]) and f['images_' + s].dtype == np.float32 "
            "for s in ('train', 'val', 'test')); print('ok')")
    r = subprocess.run([H5PY_PYTHON, "-c", code, p, os.path.join(HERE, "expected.npz")], capture_output=True, text=True)
    if "No module named" in r.stderr:
        pytest.skip("h5py not importable")
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


def test_right_align(pkg, orc):
    # misc/RNNUtils.lua:54-61; the vectorised host helper against the loop restatement
    seq = np.array([[3, 4, 5, 0, 0], [7, 0, 0, 0, 0], [1, 2, 3, 4, 5]], np.int32)
    assert pkg.dataset.right_align(seq, [3, 1, 5]).tolist() == [[0, 0, 3, 4, 5], [0, 0, 0, 0, 7], [1, 2, 3, 4, 5]]
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 27, 200)
    q = np.zeros((200, 26), np.int32)
    for i, l in enumerate(lens):
        q[i, :l] = rng.integers(1, 100, l)
    assert np.array_equal(pkg.dataset.right_align(q, lens), orc.right_align(q, lens))
    with pytest.raises(ValueError):
        pkg.dataset.right_align(q, lens + 26)


def test_vqadata_mirrors_the_loading_block(pkg, exp):
    d = pkg.dataset.VQAData(os.path.join(HERE, "data_prepro.h5"), os.path.join(HERE, "data_img.h5"),
                            os.path.join(HERE, "data_prepro.json"))
    assert d.vocabulary_size_q == 40 and d.num_answers == 12
    tr = d.split("train", arch=1)
    assert len(tr) == 37 and tr.question.dtype == np.int32 and tr.fv_im.shape == (7, 64)
    lens = exp["ques_length_train"]
    for i in range(37):                                   # right-aligned: tokens at the end, zeros in front
        assert np.array_equal(tr.question[i, 26 - lens[i]:], exp["ques_train"][i, :lens[i]])
        assert not tr.question[i, :26 - lens[i]].any()
    assert np.array_equal(tr.answers, exp["answers"]) and np.array_equal(tr.img_list, exp["img_pos_train"])
    a2 = d.split("val", arch=2)                           # arch2 keeps the stored (left-aligned) layout
    assert np.array_equal(a2.question, exp["ques_val"]) and np.array_equal(a2.answers, exp["answers_val"])
    te = d.split("test", arch=1)
    assert te.answers is None and np.array_equal(te.mc_ans, exp["MC_ans_test"]) and te.question_id is not None
    out = pkg.trainer.results_json(te.question_id[:3], [1, 2, 3], d.ix_to_ans)
    assert out[0]["answer"] == "a0" and json.loads(json.dumps(out))[2]["answer"] == "a2"
