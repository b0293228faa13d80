"""Deterministic synthetic weights for parity runs (test infrastructure).

There is no network for pretrained checkpoints, and a 51 M-parameter state_dict is too large to
commit.  Both the golden generator (which loads these weights into the REAL reference model) and the
tests (which load them into the oracle and into the HIP drop-in) call `synth_state_dict` with the
same manifest and seed, so only the manifest (key names + shapes, taken from the reference model)
and the outputs need to be stored.
"""
import math

import torch


def manifest_of(state_dict):
    return [(k, tuple(v.shape)) for k, v in state_dict.items()]


def synth_state_dict(manifest, seed=0):
    g = torch.Generator().manual_seed(seed)
    keys = {k for k, _ in manifest}
    sd = {}
    for key, shape in manifest:
        stem, _, leaf = key.rpartition(".")
        is_bn = (stem + ".running_mean") in keys
        if leaf == "num_batches_tracked":
            t = torch.zeros(shape, dtype=torch.int64)
        elif leaf == "running_mean":
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == "running_var":
            t = torch.rand(shape, generator=g) + 0.5
        elif is_bn and leaf == "weight":
            t = torch.rand(shape, generator=g) * 0.4 + 0.4   # keeps frozen-BN activations O(1) through 16 residual blocks
        elif is_bn and leaf == "bias":
            t = torch.randn(shape, generator=g) * 0.1
        elif len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / max(fan_in, 1))
        else:
            t = torch.randn(shape, generator=g) * 0.1
        sd[key] = t
    return sd


def synth_batch(N, C, H, W, num_classes, ignore_index=255, seed=1234):
    """Synthetic (image, target) recipe of SURVEY.md §8(d)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=g)
    t = torch.randint(0, num_classes, (N, H, W), generator=g)
    t[:, : max(1, H // 20), :] = ignore_index
    return x, t
