"""Deterministic stand-in weights.  The reference's checkpoint (`weights/FOTS_280000.h5`) is not
in its repository (`.MISSING_LARGE_BLOBS`), so throughput runs use random weights -- drawn per
tensor from a generator seeded by the tensor's NAME, so that the reference's module and this
package's restatement receive identical values whatever order they create their parameters in
(`tests/golden/make_e2e_golden.py` applies the same function to the reference's own class)."""
import zlib

import torch


def deterministic_init(module, seed=0):
    state = module.state_dict()
    with torch.no_grad():
        for name, t in state.items():
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
            if name.endswith("num_batches_tracked"):
                continue
            if name.endswith("running_mean"):
                v = 0.05 * torch.randn(t.shape, generator=g)
            elif name.endswith("running_var"):
                v = 1.0 + 0.1 * torch.rand(t.shape, generator=g)
            elif t.dim() >= 2:  # convolution kernels: variance-preserving for leaky/ReLU stacks
                fan_in = t[0].numel()
                v = torch.randn(t.shape, generator=g) * (2.0 / fan_in) ** 0.5
            elif name.endswith("weight"):  # norm scales
                v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
            else:  # biases, norm shifts
                v = 0.1 * torch.randn(t.shape, generator=g)
            t.copy_(v.to(t.dtype))
    return module
