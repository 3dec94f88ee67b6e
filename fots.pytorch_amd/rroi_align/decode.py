"""rroi_align.decode -- the recognition side of the crops: greedy CTC decode of all boxes of an
image in one launch and one device->host copy.

The reference decodes one word at a time (`tools/ocr_utils.py:183-186`): `labels_pred.max(1)` on the
(1, nclass, T) output of `net.forward_ocr`, a transpose/view, and `strLabelConverter.decode(...,
raw=False)` -- a Python loop over the time steps of a tensor that has just been pulled to the host
(`src/utils.py:87-97`).  With the crops of an image batched (`rroi_align.batched`), the head runs
once on (N, C, 11, max_gw) and this module turns its (N, nclass, T) logits into N strings.

`CTCLabelConverter` mirrors `strLabelConverter` (same constructor argument, same `decode`
semantics: index 0 is the CTC blank, character i of the alphabet has index i + 1).
"""
import torch

from ._ext import rroi_align as _ext


def ctc_greedy_decode(logits, lengths=None, return_labels=False):
    """(N, nclass, T) fp32 -> (decoded (N, T) int32 zero-padded, decoded_len (N,) int32[, labels])."""
    return _ext.ctc_greedy_decode(logits, lengths, return_labels)


class CTCLabelConverter(object):
    def __init__(self, alphabet, ignore_case=False):
        self._ignore_case = ignore_case
        if self._ignore_case:
            alphabet = alphabet.lower()
        self.alphabet = alphabet + '-'  # index -1, as in the reference (src/utils.py:45)
        self.dict = {char: i + 1 for i, char in enumerate(alphabet)}

    def decode_logits(self, logits, lengths=None):
        """logits (N, nclass, T) on the GPU -> list of N strings (one kernel, one copy)."""
        decoded, dlen = ctc_greedy_decode(logits, lengths)
        return self.to_texts(decoded, dlen)

    def to_texts(self, decoded, decoded_len):
        """(N, T) collapsed labels + (N,) lengths (device or host tensors) -> N strings.  ONE copy each and plain lists from
        there on: indexing and iterating a tensor per word and per label costs 30 us a word (round 6: 0.7 of the 1.8 ms the
        recognition of an image's 24 words took)."""
        rows, lens = decoded.cpu().tolist(), decoded_len.cpu().tolist()
        return [self.to_text(row[:n]) for row, n in zip(rows, lens)]

    def to_text(self, kept_labels):
        """Already-collapsed labels -> string (the inner expression of src/utils.py:96)."""
        if isinstance(kept_labels, torch.Tensor):
            kept_labels = kept_labels.tolist()
        alphabet = self.alphabet
        return ''.join([alphabet[int(t) - 1] for t in kept_labels])

    def decode(self, t, length, raw=False):
        """Host-side decode of ONE label sequence, `strLabelConverter.decode` semantics
        (src/utils.py:87-97); kept for call-site compatibility."""
        t = torch.as_tensor(t).view(-1)
        n = int(torch.as_tensor(length).view(-1)[0])
        if t.numel() != n:
            raise AssertionError("text with length: {} does not match declared length: {}".format(t.numel(), n))
        if raw:
            return ''.join(self.alphabet[int(i) - 1] for i in t)
        keep = [int(t[i]) for i in range(n) if t[i] != 0 and not (i > 0 and t[i - 1] == t[i])]
        return self.to_text(keep)
