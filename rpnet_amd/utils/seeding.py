"""Name-seeded parameter fill.

Weights are never shipped: every parameter/buffer of the RP-Net hot path is
filled from a generator seeded by crc32(<state_dict key>), so the reference
model (imported only by tests/golden/gen_golden.py), the oracle and the HIP
path on the GPU box all see bit-identical values.
"""
import zlib

import torch


def _gen(name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return g


def seeded_tensor(name: str, like: torch.Tensor) -> torch.Tensor:
    """Deterministic CPU value for the state_dict entry `name` shaped like `like`."""
    shape = tuple(like.shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    u = torch.rand(shape, generator=_gen(name), dtype=torch.float32)
    if leaf == "running_var":
        return 0.5 + u
    if leaf == "running_mean":
        return (u - 0.5) * 0.2
    if like.dim() == 4:  # conv weight [Cout, Cin, kh, kw]: fan-in scaled uniform
        fan_in = shape[1] * shape[2] * shape[3]
        s = (3.0 / fan_in) ** 0.5
        return (u * 2 - 1) * s
    # 1-D weight/bias: conv bias, BN weight, BN bias
    if leaf == "weight":  # BN gamma
        return 0.9 + 0.2 * u
    return (u - 0.5) * 0.2


def seed_state_dict(state_dict) -> dict:
    """Return a new {key: CPU tensor} with every entry replaced by its seeded value."""
    return {k: seeded_tensor(k, v) for k, v in state_dict.items()}


def seed_module_(module) -> None:
    """In-place name-seeded fill of an nn.Module (on whatever device it lives)."""
    sd = module.state_dict()
    new = seed_state_dict(sd)
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(new[k].to(v.device, v.dtype))
