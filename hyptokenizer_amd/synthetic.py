"""Deterministic synthetic inputs for the merge engine.

Everything here is plain torch on the CPU generator, so the same seed gives the same table in the
build container and on the GPU box.  No file of the reference is read.
"""
from __future__ import annotations

from typing import List

import torch


def lorentz_table(vocab_size: int, dim: int, seed: int = 42, scale: float = 0.05,
                  curvature: float = 1.0) -> torch.Tensor:
    """Random points on the hyperboloid, ``[vocab_size, dim + 1]`` fp32, column 0 = time.

    ``S = randn(V, d) * scale`` and ``x0 = sqrt(1 + c * ||S||^2)`` (the arithmetic of the
    reference's ``project_to_hyperboloid``, ``embedding/lorentz_model.py:41-56``).
    """
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    spatial = torch.randn(vocab_size, dim, generator=g, dtype=torch.float32) * scale
    r = torch.norm(spatial, dim=-1, keepdim=True)
    x0 = torch.sqrt(1.0 + curvature * r * r)
    return torch.cat([x0, spatial], dim=-1).contiguous()


def cjk_vocab(vocab_size: int) -> List[str]:
    """Single-character tokens ``chr(0x4e00 + i)``: every merge weight starts at 0.5."""
    return [chr(0x4E00 + i) for i in range(vocab_size)]
