"""Deterministic parameter values by NAME, shared by tests/golden/make_golden_sam_full.py (which feeds them to the reference's
SAM encoder) and the GPU test (which feeds them to the product's): a 90 M-parameter state dict cannot be stored as a fixture,
and seeding two differently constructed modules does not give them the same numbers."""
import zlib

import torch


def det_tensor(name: str, shape, seed: int = 2026) -> torch.Tensor:
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
    shape = tuple(shape)
    if name.endswith("norm1.weight") or name.endswith("norm2.weight") or (name.startswith("neck.") and name.endswith(".weight") and len(shape) == 1):
        return 1.0 + 0.1 * torch.randn(shape, generator=g)                 # LayerNorm / LayerNorm2d gains
    if len(shape) == 1:
        return 0.05 * torch.randn(shape, generator=g)                      # biases
    if "rel_pos" in name:
        return 0.2 * torch.randn(shape, generator=g)                       # zero in a fresh SAM: make the bias path live
    if "pos_embed" in name:
        return 0.1 * torch.randn(shape, generator=g)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return torch.randn(shape, generator=g) * (fan_in ** -0.5)              # linear / conv weights: unit-gain


def det_state(shapes: dict, seed: int = 2026) -> dict:
    return {k: det_tensor(k, s, seed) for k, s in shapes.items()}


def det_image(shape, seed: int = 7) -> torch.Tensor:
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))
