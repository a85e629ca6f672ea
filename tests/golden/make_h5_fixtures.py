"""Writes tests/golden/h5/*: small HDF5 files produced by libhdf5 itself (through h5py), with the
reference's own writer calls, to pin novel-vqa_amd/host/h5.py (which parses the format from the
specification).  Run with an interpreter that has h5py -- in the build image:

    /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py

(h5py is not importable from /usr/bin/python3, so the test-suite only reads the committed files.)
Expected contents are stored beside them in expected.npz / data_prepro.json.
"""
import json
import os

import h5py
import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "h5")
os.makedirs(HERE, exist_ok=True)
rng = np.random.RandomState(123)

N = {"train": 37, "val": 11, "test": 9}
T, V, A, NIMG, I = 26, 40, 12, {"train": 7, "val": 5, "test": 4}, 64
exp = {}


def questions(n):
    lens = rng.randint(1, T + 1, n)
    q = np.zeros((n, T), np.int64)
    for i, l in enumerate(lens):
        q[i, :l] = rng.randint(1, V + 1, l)      # left-aligned, 0 = padding (000_prepro_vqa.py encode_question)
    return q, lens


# data_prepro.h5: exactly the create_dataset calls of 002_train_vqa_arch1/000_prepro_vqa.py:273-300
f = h5py.File(os.path.join(HERE, "data_prepro.h5"), "w")
for split in ("train", "val", "test"):
    q, lens = questions(N[split])
    qid = rng.randint(1, 10 ** 6, N[split])
    pos = rng.randint(1, NIMG[split] + 1, N[split])
    f.create_dataset("ques_" + split, dtype="uint32", data=q)
    f.create_dataset("ques_length_" + split, dtype="uint32", data=lens)
    f.create_dataset("question_id_" + split, dtype="uint32", data=qid)
    f.create_dataset("img_pos_" + split, dtype="uint32", data=pos)
    exp.update({"ques_" + split: q, "ques_length_" + split: lens, "question_id_" + split: qid, "img_pos_" + split: pos})
    if split == "train":
        a = rng.randint(1, A + 1, N[split])
        f.create_dataset("answers", dtype="uint32", data=a)
        exp["answers"] = a
    elif split == "val":
        a = rng.randint(1, A + 1, N[split])
        f.create_dataset("answers_val", dtype="uint32", data=a)
        exp["answers_val"] = a
    else:
        mc = rng.randint(0, A + 1, (N[split], 18))
        f.create_dataset("MC_ans_test", dtype="uint32", data=mc)
        exp["MC_ans_test"] = mc
f.close()

json.dump({"ix_to_word": {str(i + 1): "w%d" % i for i in range(V)},
           "ix_to_ans": {str(i + 1): "a%d" % i for i in range(A)},
           "unique_img_train": ["train/%d.jpg" % i for i in range(NIMG["train"])],
           "unique_img_val": ["val/%d.jpg" % i for i in range(NIMG["val"])],
           "unique_img_test": ["test/%d.jpg" % i for i in range(NIMG["test"])]},
          open(os.path.join(HERE, "data_prepro.json"), "w"))

# data_img.h5: float32 feature matrices (001_prepro_img_vgg.lua:156-160 writes them with torch-hdf5,
# which is libhdf5 with default properties = contiguous)
f = h5py.File(os.path.join(HERE, "data_img.h5"), "w")
for split in ("train", "val", "test"):
    x = np.abs(rng.randn(NIMG[split], I)).astype(np.float32)
    f.create_dataset("images_" + split, data=x)
    exp["images_" + split] = x
f.close()

# layouts / types the reader also claims: chunked (+gzip, +shuffle, +fletcher32), compact-ish tiny
# arrays, big-endian, int64 / float64, a nested group, an edge chunk, libver='latest' headers
f = h5py.File(os.path.join(HERE, "variants.h5"), "w")
x = rng.randn(23, 10).astype(np.float32)
f.create_dataset("chunked", data=x, chunks=(8, 4))
f.create_dataset("gzip", data=x, chunks=(8, 4), compression="gzip", compression_opts=4)
f.create_dataset("gzip_shuffle_f32", data=x, chunks=(5, 10), compression="gzip", shuffle=True, fletcher32=True)
y = rng.randint(-1000, 1000, (3, 4, 5))
f.create_dataset("be_i32", data=y, dtype=">i4")
f.create_dataset("i64", data=y.astype(np.int64))
f.create_dataset("f64", data=x.astype(np.float64))
f.create_dataset("u8", data=(y % 256).astype(np.uint8))
f.create_dataset("scalar_like", data=np.array([7], np.uint32))
g = f.create_group("grp")
g.create_dataset("inner", data=np.arange(12, dtype=np.uint16).reshape(3, 4))
f.create_dataset("never_written", shape=(4, 3), dtype="float32")
big = rng.randint(0, 2 ** 31, (300, 70)).astype(np.uint32)
f.create_dataset("many_chunks", data=big, chunks=(7, 9))       # > one B-tree node worth of chunks
f.close()
exp.update({"v_chunked": x, "v_be_i32": y, "v_u8": (y % 256).astype(np.uint8), "v_inner": np.arange(12).reshape(3, 4),
            "v_many_chunks": big})

f = h5py.File(os.path.join(HERE, "latest.h5"), "w", libver="latest")
f.create_dataset("a", data=x)
f.create_dataset("b", data=y.astype(np.int32))
f.close()

np.savez_compressed(os.path.join(HERE, "expected.npz"), **exp)
print("wrote", sorted(os.listdir(HERE)))
