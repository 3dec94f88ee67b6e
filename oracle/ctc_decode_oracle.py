"""oracle/ctc_decode_oracle.py -- CPU restatement of the reference's per-word decode
(SURVEY.md section 8f rank 1).  TEST INFRASTRUCTURE ONLY (see rroi_align_oracle.py): imported by
tests/ only.  Citations relative to /root/reference.

tools/ocr_utils.py:183-186
    _, labels_pred = labels_pred.max(1)           arg max over the class axis of (N, nclass, T);
                                                  torch: first index of the largest value, NaN largest
    labels_pred = labels_pred.transpose(1, 0).contiguous().view(-1)
    sim_preds = converter.decode(labels_pred.data, preds_size.data, raw=False)
src/utils.py:93-97 (strLabelConverter.decode, one sequence, raw=False)
    keep t[i] iff t[i] != 0 and not (i > 0 and t[i-1] == t[i]);  char = alphabet[t[i] - 1]
src/utils.py:45,48-50: alphabet + '-', index 0 = blank.

Pinned by tests/golden/ctc_decode.npz (made by tests/golden/make_ctc_golden.py, which executes the
reference's own strLabelConverter class and torch.max on seeded inputs in the authoring container).
"""
import numpy as np


def argmax_labels(logits):
    """(N, K, T) -> (N, T) int32; plain loops so that the tie / NaN rule is explicit."""
    x = np.asarray(logits, np.float32)
    N, K, T = x.shape
    out = np.zeros((N, T), np.int32)
    for n in range(N):
        for t in range(T):
            best, bv = 0, x[n, 0, t]
            for k in range(1, K):
                v = x[n, k, t]
                if not np.isnan(bv) and (np.isnan(v) or v > bv):
                    best, bv = k, v
            out[n, t] = best
    return out


def collapse(labels, lengths=None):
    """(N, T) labels -> (decoded (N, T) zero-padded, decoded_len (N,)), src/utils.py:93-97."""
    lab = np.asarray(labels)
    N, T = lab.shape
    dec = np.zeros((N, T), np.int32)
    dlen = np.zeros(N, np.int32)
    for n in range(N):
        L = T if lengths is None else max(0, min(T, int(lengths[n])))
        k = 0
        for i in range(L):
            if lab[n, i] != 0 and not (i > 0 and lab[n, i - 1] == lab[n, i]):
                dec[n, k] = lab[n, i]
                k += 1
        dlen[n] = k
    return dec, dlen


def to_text(kept, alphabet):
    a = alphabet + '-'
    return ''.join(a[int(t) - 1] for t in kept)
