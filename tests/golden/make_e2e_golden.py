#!/usr/bin/env python3
"""Pins fots_e2e.model.FOTSNet to the reference's OWN network class.

Runs in the authoring container only (needs /root/reference): imports the reference's
`tools/models.py` in place, builds `ModelResNetSep2(attention=True, nclass=87)`, gives it the
name-seeded stand-in weights of `fots_e2e.weights.deterministic_init` (the same function the tests
apply to the restatement), and stores a seeded input with the outputs the reference's module
computes for it: the two score / rbox / angle maps, the merged 256-channel map, `focr`, and
`forward_ocr` of seeded crops.  Only arrays are stored; no reference source travels.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "fots.pytorch_amd"))
from fots_e2e.weights import deterministic_init  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_models", "/root/reference/tools/models.py")
ref_models = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_models)

torch.manual_seed(0)
torch.set_num_threads(1)
net = deterministic_init(ref_models.ModelResNetSep2(attention=True, nclass=87)).eval()
x = torch.randn(1, 3, 64, 96)
crops = torch.randn(2, 64, 11, 64)
with torch.no_grad():
    score, rbox, angle, feats = net(x)
    logp = net.forward_ocr(crops)
np.savez_compressed(
    os.path.join(HERE, "e2e_model.npz"),
    x=x.numpy(), crops=crops.numpy(),
    score4=score[0].numpy(), score8=score[1].numpy(), rbox4=rbox[0].numpy(), rbox8=rbox[1].numpy(),
    angle4=angle[0].numpy(), angle8=angle[1].numpy(), merged=feats[0].numpy(), focr=feats[1].numpy(),
    logp=logp.numpy(), keys=np.array(sorted(net.state_dict().keys())))
print("wrote e2e_model.npz:", {k: tuple(v.shape) for k, v in (("merged", feats[0]), ("focr", feats[1]), ("logp", logp))})
