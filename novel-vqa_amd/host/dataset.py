"""The data files either side of the training step, in the reference's own formats.

Mirror of the loading block of the training / evaluation scripts
(002_train_vqa_arch1/002_train_baseline.lua:84-126, 003_train_vqa_arch2/002_train_baseline.lua:86-130,
002_train_vqa_arch1/004_eval_model.lua:82-140):

    json_file           = cjson.decode(io.open(opt.input_json))          -- ix_to_word, ix_to_ans, unique_img_*
    dataset.question    = h5:read('/ques_train'):all()                   -- [N x T] uint32, 0 = padding, left-aligned
    dataset.lengths_q   = h5:read('/ques_length_train'):all()            -- [N]
    dataset.img_list    = h5:read('/img_pos_train'):all()                -- [N] 1-based row of fv_im
    dataset.answers     = h5:read('/answers'):all()                      -- [N] 1-based answer id
    dataset.fv_im       = h5img:read('/images_train'):all()              -- [N_img x I] float32
    dataset.question    = right_align(dataset.question, dataset.lengths_q)   (arch1 only)
    vocabulary_size_q   = #json_file.ix_to_word

The L2 normalisation of the image features (:117-121) is done on the device when the split is
handed to `VQATrainer.load_dataset(..., img_norm=True)` (k_l2norm_rows).
"""
import json

import numpy as np

from . import h5

# names of the per-split datasets in data_prepro.h5 (000_prepro_vqa.py:276-296)
_ANSWERS = {"train": "answers", "val": "answers_val", "test": None}


def right_align(seq, lengths):
    """misc/RNNUtils.lua:54-61: v[i][N-lengths[i]+1 .. N] = seq[i][1 .. lengths[i]], zeros on the left.
    A row of length 0 stays all-zero (Torch would raise on the empty range; the pre-processing never
    emits one)."""
    seq = np.asarray(seq)
    lengths = np.asarray(lengths).astype(np.int64)
    n, T = seq.shape
    if lengths.shape != (n,) or (lengths < 0).any() or (lengths > T).any():
        raise ValueError("right_align: lengths must be [N] with 0 <= length <= T")
    col = np.arange(T)[None, :]
    src = col - (T - lengths)[:, None]            # source column of every destination column
    out = np.take_along_axis(seq, np.clip(src, 0, T - 1), axis=1)
    out[src < 0] = 0
    return out


class VQASplit:
    """One split of the pre-processed data set, as the arrays `dataset:next_batch()` indexes."""

    def __init__(self, question, lengths, img_list, answers, fv_im, question_id=None, mc_ans=None):
        self.question = question      # int32 [N x T]
        self.lengths = lengths        # int32 [N]
        self.img_list = img_list      # int32 [N], 1-based
        self.answers = answers        # int32 [N], 1-based (None for the test split)
        self.fv_im = fv_im            # float32 [N_img x I] (not normalised)
        self.question_id = question_id
        self.mc_ans = mc_ans          # int32 [N x 18] candidate answer ids, 0 = empty (test split)

    def __len__(self):
        return int(self.question.shape[0])


class VQAData:
    """`VQAData(input_ques_h5, input_img_h5, input_json)`; `.split('train', arch=1)`."""

    def __init__(self, input_ques_h5, input_img_h5, input_json):
        self.input_ques_h5, self.input_img_h5 = input_ques_h5, input_img_h5
        with open(input_json, "r") as f:
            self.json_file = json.load(f)
        self.ix_to_word = self.json_file["ix_to_word"]
        self.ix_to_ans = self.json_file["ix_to_ans"]
        # `for i, w in pairs(json_file['ix_to_word']) do count = count + 1 end` (:125-127)
        self.vocabulary_size_q = len(self.ix_to_word)
        self.num_answers = len(self.ix_to_ans)

    def split(self, name="train", arch=1):
        """arch = 1: questions right-aligned (002_train_baseline.lua:113-114); arch = 2: left as stored
        (the arch2 scripts never call right_align)."""
        if name not in _ANSWERS:
            raise ValueError(f"unknown split '{name}'")
        with h5.File(self.input_ques_h5) as f:
            keys = set(f.keys())
            q = f.read("/ques_" + name)
            lens = f.read("/ques_length_" + name)
            pos = f.read("/img_pos_" + name)
            ans = f.read("/" + _ANSWERS[name]) if _ANSWERS[name] else None
            qid = f.read("/question_id_" + name) if "question_id_" + name in keys else None
            mc = f.read("/MC_ans_test") if name == "test" and "MC_ans_test" in keys else None
        with h5.File(self.input_img_h5) as f:
            fv = f.read("/images_" + name)
        if q.ndim != 2 or lens.shape != (q.shape[0],) or pos.shape != (q.shape[0],):
            raise ValueError(f"{self.input_ques_h5}: inconsistent shapes in split '{name}'")
        if fv.ndim != 2 or (pos.size and (pos.min() < 1 or pos.max() > fv.shape[0])):
            raise ValueError(f"img_pos_{name} does not index images_{name} ({fv.shape[0]} rows)")
        if q.size and q.max() > self.vocabulary_size_q:
            raise ValueError("token id beyond the vocabulary of the json file")
        if arch == 1:
            q = right_align(q, lens)
        i32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
        return VQASplit(i32(q), i32(lens), i32(pos), i32(ans), np.ascontiguousarray(fv, dtype=np.float32),
                        None if qid is None else qid.astype(np.int64), i32(mc))


def write_features(path, train, val, test):
    """001_prepro_img_vgg.lua:156-160: the extractor's output file."""
    h5.write(path, {"images_train": np.asarray(train, np.float32), "images_test": np.asarray(test, np.float32),
                    "images_val": np.asarray(val, np.float32)})
