"""SURVEY 8(f) rank 3 / BASELINE configs[4], CPU side: the in-repo restatement of the reference's
network equals the reference's own module (`tests/golden/e2e_model.npz`, produced by
`make_e2e_golden.py` from `/root/reference/tools/models.py` executed in place), key for key and
output for output; plus the host-side rules of the driver (`test.py:25-41`)."""
import os

import numpy as np
import pytest
import torch

from fots_e2e.alphabet import ALPHABET
from fots_e2e.model import FOTSNet
from e2e_inputs import synthetic_boxes
from fots_e2e.pipeline import resize_rule, target_widths_host
from oracle.e2e_loop_oracle import host_roi
from fots_e2e.weights import deterministic_init

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e2e_model.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def net():
    torch.set_num_threads(1)
    return deterministic_init(FOTSNet(len(ALPHABET) + 1)).eval()


def test_state_dict_keys_are_the_references(gold, net):
    assert sorted(net.state_dict().keys()) == list(gold["keys"])
    assert len(ALPHABET) == 86  # data/alphabet.txt; class 0 is the CTC blank (src/utils.py:45-50)


def _close(got, want, what):
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert err <= 1e-4 * scale, (what, err, scale)  # fp32, same operations: observed <= 4e-6 of the scale


def test_forward_matches_the_reference_module(gold, net):
    with torch.no_grad():
        score, rbox, angle, feats = net(torch.from_numpy(gold["x"]))
    assert np.array_equal(feats[1].numpy(), gold["focr"])  # the map RoIRotate samples: bit for bit
    for got, key in ((score[0], "score4"), (score[1], "score8"), (rbox[0], "rbox4"), (rbox[1], "rbox8"),
                     (angle[0], "angle4"), (angle[1], "angle8"), (feats[0], "merged")):
        assert got.shape == gold[key].shape
        _close(got.numpy(), gold[key], key)


def test_forward_ocr_matches_the_reference_module(gold, net):
    with torch.no_grad():
        logp = net.forward_ocr(torch.from_numpy(gold["crops"]))
    assert logp.shape == gold["logp"].shape == (2, 87, 64)  # (N, nclass, T = pooled width)
    _close(logp.numpy(), gold["logp"], "logp")
    assert np.array_equal(logp.numpy().argmax(1), gold["logp"].argmax(1))


def test_resize_rule():
    assert resize_rule(720, 1280) == (704, 1280)          # the example images
    assert resize_rule(64, 96) == (64, 96)
    h, w = resize_rule(3000, 4000)                         # area cap 1585152 (test.py:25)
    assert h % 32 == 0 and w % 32 == 0 and h * w <= 1585152
    assert resize_rule(100, 100, scale_up=True) == (288, 288)


def test_host_roi_is_the_callers_rule():
    """tools/ocr_utils.py:133-150 on a hand-computed box (the demo quad of rroi_align/test2.py:43)."""
    import math
    box = np.asarray([206, 111, 199, 95, 349, 60, 355, 80, 0.9], np.float32)
    roi, gw = host_roi(box)
    w, h = math.sqrt(150 ** 2 + 35 ** 2), math.sqrt(7 ** 2 + 16 ** 2)
    assert roi[:3] == [0, 277, 86] and roi[3] == h and roi[4] == w
    assert roi[5] == -math.atan2(-35, 150) / 3.1415926535 * 180
    assert gw == max(2, (int(w * (11 / h)) + 11) // 32) * 32 == 96
    b = synthetic_boxes(50, 704, 1280, seed=1)
    # the product's vectorised width rule gives the per-word rule's widths
    assert target_widths_host(b) == [host_roi(x)[1] for x in b] and target_widths_host(box[None]) == [96]
    assert b.shape == (50, 9) and (b[:, :8] >= 0).all() and (b[:, 0:8:2] <= 1280).all() and (b[:, 1:8:2] <= 704).all()
