"""Host-side logic that needs no GPU: data helpers and answer emission."""
import numpy as np


def test_right_align_matches_reference_semantics(orc):
    # misc/RNNUtils.lua:54-61
    seq = np.array([[3, 4, 5, 0, 0], [7, 0, 0, 0, 0], [1, 2, 3, 4, 5]], np.int32)
    out = orc.right_align(seq, [3, 1, 5])
    assert out.tolist() == [[0, 0, 3, 4, 5], [0, 0, 0, 0, 7], [1, 2, 3, 4, 5]]


def test_multiple_choice_argmax(pkg):
    # 004_eval_model.lua:259-271: argmax restricted to the non-zero candidate ids
    scores = np.array([[0.1, 0.9, 0.3, 0.5], [0.7, 0.2, 0.7, 0.1]])
    mc = np.array([[3, 4, 0], [3, 1, 4]])
    assert pkg.trainer.multiple_choice_argmax(scores, mc).tolist() == [4, 3]


def test_trainer_constants(pkg):
    assert abs(pkg.trainer.DECAY_FACTOR ** 28782 - 0.5) < 1e-3  # halves about every 28.8k iterations
