"""Greedy CTC decode: oracle pinned to the reference's own decode, HIP kernel
bit-exact against the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "fots.pytorch_amd")]
from oracle import ctc_decode_oracle as CO  # noqa: E402

GOLD = os.path.join(HERE, "golden", "ctc_decode.npz")


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLD)
    return {k: z[k] for k in z.files}


def test_oracle_matches_reference_decode(gold):
    """labels = torch.max(1) of the reference call, texts = the reference's strLabelConverter.decode."""
    lab = CO.argmax_labels(gold["logits"])
    assert np.array_equal(lab, gold["labels"])
    dec, dlen = CO.collapse(lab)
    alphabet = str(gold["alphabet"])
    texts = [CO.to_text(dec[n, :dlen[n]], alphabet) for n in range(len(lab))]
    assert texts == [str(t) for t in gold["texts"]]
    assert (dlen < lab.shape[1]).all() and (dlen > 0).all()  # the vectors do exercise collapsing


def test_host_converter_mirrors_reference(gold):
    from rroi_align.decode import CTCLabelConverter
    conv = CTCLabelConverter(str(gold["alphabet"]))
    T = gold["labels"].shape[1]
    for n in range(8):
        seq = torch.from_numpy(gold["labels"][n])
        assert conv.decode(seq, torch.IntTensor([T]), raw=False) == str(gold["texts"][n])
    for n in range(4):
        assert conv.decode(torch.from_numpy(gold["labels"][n]), torch.IntTensor([T]), raw=True) == str(gold["raw_texts"][n])
    with pytest.raises(AssertionError):
        conv.decode(torch.zeros(5, dtype=torch.int32), torch.IntTensor([4]))


def test_host_converter_batch_of_collapsed_labels(gold):
    """`to_texts` (round 6: one copy, plain lists) against the reference's texts and against `to_text` word by word, for
    tensors, arrays and lists of labels."""
    from rroi_align.decode import CTCLabelConverter
    conv = CTCLabelConverter(str(gold["alphabet"]))
    dec, dlen = CO.collapse(gold["labels"])
    want = [str(t) for t in gold["texts"]]
    assert conv.to_texts(torch.from_numpy(dec.astype(np.int32)), torch.from_numpy(dlen.astype(np.int32))) == want
    for n in range(len(want)):
        row = dec[n, :dlen[n]]
        assert conv.to_text(torch.from_numpy(row.astype(np.int32))) == conv.to_text(row) == conv.to_text(row.tolist()) == want[n]
    assert conv.to_texts(torch.zeros((0, 7), dtype=torch.int32), torch.zeros((0,), dtype=torch.int32)) == []
    assert conv.to_text([]) == ""


def test_oracle_rules():
    x = np.zeros((1, 4, 6), np.float32)
    x[0, 2, 0] = 1; x[0, 2, 1] = 1          # repeated label collapses
    x[0, 0, 2] = 1                          # blank separates
    x[0, 2, 3] = 1                          # same label again after a blank is kept
    x[0, 1, 4] = 1; x[0, 3, 4] = 1          # tie -> first index
    x[0, 3, 5] = np.nan                     # NaN is the largest
    lab = CO.argmax_labels(x)
    assert lab.tolist() == [[2, 2, 0, 2, 1, 3]]
    dec, dlen = CO.collapse(lab)
    assert dec[0, :dlen[0]].tolist() == [2, 2, 1, 3]
    dec, dlen = CO.collapse(lab, lengths=[3])
    assert dec[0, :dlen[0]].tolist() == [2] and dec[0, 1:].tolist() == [0] * 5


@pytest.mark.gpu
def test_kernel_on_reference_vectors(gold):
    from rroi_align.decode import CTCLabelConverter, ctc_greedy_decode
    logits = torch.from_numpy(gold["logits"]).cuda()
    dec, dlen, lab = ctc_greedy_decode(logits, return_labels=True)
    assert np.array_equal(lab.cpu().numpy(), gold["labels"])
    conv = CTCLabelConverter(str(gold["alphabet"]))
    assert conv.decode_logits(logits) == [str(t) for t in gold["texts"]]
    wdec, wlen = CO.collapse(gold["labels"])
    assert np.array_equal(dec.cpu().numpy(), wdec) and np.array_equal(dlen.cpu().numpy(), wlen)


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,T", [(1, 87, 16), (24, 87, 88), (5, 3, 64), (7, 40, 65), (3, 87, 300), (2, 1, 10)])
def test_kernel_vs_oracle(N, K, T):
    from rroi_align.decode import ctc_greedy_decode
    rng = np.random.default_rng(N * 1000 + T)
    x = (np.round(rng.standard_normal((N, K, T)) * 1.5) / 1.5).astype(np.float32)  # many exact ties
    x[rng.random(x.shape) < 0.002] = np.nan
    x[rng.random(x.shape) < 0.002] = np.inf
    x[:, 0, :] += (rng.random((N, T)) < 0.3) * 5.0  # blanks
    lengths = rng.integers(0, T + 1, N).astype(np.int32)
    lengths[0] = T
    wl = CO.argmax_labels(x)
    for ln in (None, lengths):
        dec, dlen, lab = ctc_greedy_decode(torch.from_numpy(x).cuda(), None if ln is None else torch.from_numpy(ln),
                                           return_labels=True)
        wdec, wlen = CO.collapse(wl, ln)
        assert np.array_equal(lab.cpu().numpy(), wl)
        assert np.array_equal(dec.cpu().numpy(), wdec)
        assert np.array_equal(dlen.cpu().numpy(), wlen)


@pytest.mark.gpu
def test_kernel_edge_cases():
    from rroi_align.decode import ctc_greedy_decode
    dec, dlen = ctc_greedy_decode(torch.zeros((0, 87, 20), device="cuda"))
    assert dec.shape == (0, 20) and dlen.numel() == 0
    dec, dlen = ctc_greedy_decode(torch.zeros((3, 87, 0), device="cuda"))
    assert dec.shape == (3, 0) and dlen.cpu().tolist() == [0, 0, 0]
    with pytest.raises(RuntimeError):
        ctc_greedy_decode(torch.zeros((1, 87, 4)))  # CPU tensor: no fallback
    with pytest.raises(ValueError):
        ctc_greedy_decode(torch.zeros((2, 87, 4), device="cuda"), lengths=[1])
