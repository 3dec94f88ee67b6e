#!/usr/bin/env python3
"""Golden vectors for the greedy CTC decode: runs the REFERENCE's own code in this container --
the `strLabelConverter` class, taken out of /root/reference/src/utils.py with `ast` (the module
itself does not import here: cv2 and torchvision are missing) and executed in place, and
`torch.max(dim=1)` exactly as tools/ocr_utils.py:183 calls it -- on seeded inputs, and stores only
inputs and expected outputs (tests/golden/ctc_decode.npz).  Nothing of the reference's text is
written anywhere.  Usage: python tests/golden/make_ctc_golden.py"""
import ast
import collections
import os

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

src = open(os.path.join(REF, "src/utils.py"), encoding="utf-8").read()
cls = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "strLabelConverter")
ns = {"torch": torch, "collections": collections}
exec(compile(ast.Module(body=[cls], type_ignores=[]), "ref:strLabelConverter", "exec"), ns)
alphabet = open(os.path.join(REF, "data/alphabet.txt"), encoding="utf-8").readlines()[0]
conv = ns["strLabelConverter"](alphabet)
K = len(alphabet) + 1  # test.py:59: nclass = len(alphabet) + 1

rng = np.random.default_rng(0)
N, T = 48, 75
# logits with runs, blanks and exact ties: quantised values make equal maxima common
logits = np.round(rng.standard_normal((N, K, T)).astype(np.float32) * 2) / 2
for n in range(N):
    t = 0
    while t < T:  # plant runs of one label (incl. the blank) so that collapsing matters
        run = int(rng.integers(1, 5))
        k = int(rng.integers(0, K)) if rng.random() < 0.7 else 0
        logits[n, k, t:t + run] += 6.0
        t += run
labels = torch.from_numpy(logits).max(1)[1]  # ocr_utils.py:183
texts = []
for n in range(N):
    seq = labels[n].contiguous().view(-1)  # :184 for one word
    texts.append(conv.decode(seq.data, torch.IntTensor([seq.size(0)]), raw=False))  # :186
raw = [conv.decode(labels[n].view(-1), torch.IntTensor([T]), raw=True) for n in range(4)]
np.savez_compressed(os.path.join(HERE, "ctc_decode.npz"), logits=logits, labels=labels.numpy().astype(np.int32),
                    texts=np.array(texts), raw_texts=np.array(raw), alphabet=np.array(alphabet))
print("wrote ctc_decode.npz:", N, "sequences, e.g.", repr(texts[0]))
