"""Lorentz-model primitives with the reference's function surface, served by the gfx950 kernels.

Mirrors ``embedding/lorentz_model.py`` of the reference (function names, argument meaning, result
shapes).  The hot-path functions -- ``minkowski_dot``, ``distance``, ``batch_distance``,
``batch_distance_optimized``, ``log_map``, ``exp_map``, ``project_to_hyperboloid`` -- run as HIP
kernels through the C ABI and need tensors on a HIP device; on a CPU tensor they raise
``HypMergeUnavailable`` (there is no CPU fallback).  The remaining helpers are not on the hot
path and are kept as short torch expressions on top of ``minkowski_dot``.

Sign convention: the reference's ``minkowski_dot`` is ``x0*y0 - sum`` and its
``distance`` feeds ``-minkowski_dot`` to acosh, which makes every distance 0.0.  The module-level
default ``"reference"`` reproduces exactly that; ``"lorentz"`` flips the sign of the form (the
behaviour the reference's own ``test_distance`` expects).  Every function takes an optional
``sign_convention=`` keyword; ``set_sign_convention`` changes the default.
"""
from __future__ import annotations

from typing import Optional

import torch

from ..engine import device_batch_distance, device_rows_op, sign_mode_id

_DEFAULT_SIGN = "reference"


def set_sign_convention(sign_convention: str) -> None:
    """Select the module default: ``"reference"`` (as shipped) or ``"lorentz"`` (sign-corrected)."""
    global _DEFAULT_SIGN
    sign_mode_id(sign_convention)
    _DEFAULT_SIGN = sign_convention


def get_sign_convention() -> str:
    return _DEFAULT_SIGN


def _sign(sign_convention) -> int:
    return sign_mode_id(_DEFAULT_SIGN if sign_convention is None else sign_convention)


def minkowski_dot(x: torch.Tensor, y: torch.Tensor, *, sign_convention: Optional[str] = None) -> torch.Tensor:
    """Reference ``lorentz_model.py:14-25``: ``x0*y0 - sum_k xk*yk`` (negated under "lorentz")."""
    return device_rows_op("minkowski", x, y, 1.0, _sign(sign_convention))


def minkowski_norm(x: torch.Tensor, *, sign_convention: Optional[str] = None) -> torch.Tensor:
    """Reference ``:28-38`` (not on the hot path)."""
    return torch.sqrt(torch.clamp(minkowski_dot(x, x, sign_convention=sign_convention), min=1e-8))


def project_to_hyperboloid(x: torch.Tensor, c: float = 1.0) -> torch.Tensor:
    """Reference ``:41-56``: keep the spatial part, ``x0 = sqrt(1 + c*||x_1:||^2)``."""
    return device_rows_op("project", x, None, float(c), 0)


def lorentz_to_klein(x: torch.Tensor, c: float = 1.0) -> torch.Tensor:
    """Reference ``:59-70`` (only fed FAISS in the reference; not on the hot path)."""
    return x[..., 1:] / x[..., 0:1]


def exp_map(x: torch.Tensor, v: torch.Tensor, c: float = 1.0) -> torch.Tensor:
    """Reference ``:73-93``: Euclidean norm of the spatial part of v, ``cosh(n) x + sinh(n) v/n``."""
    return device_rows_op("exp_map", x, v, 1.0, 0)


def log_map(x: torch.Tensor, y: torch.Tensor, c: float = 1.0, *, sign_convention: Optional[str] = None) -> torch.Tensor:
    """Reference ``:96-119`` (the curvature argument is ignored there as well)."""
    return device_rows_op("log_map", x, y, 1.0, _sign(sign_convention))


def distance(x: torch.Tensor, y: torch.Tensor, c: float = 1.0, *, sign_convention: Optional[str] = None) -> torch.Tensor:
    """Reference ``:122-138``: ``acosh(clamp(u, 1)) / sqrt(c)`` on broadcast operands."""
    return device_rows_op("distance", x, y, float(c), _sign(sign_convention))


def batch_distance(x: torch.Tensor, y: torch.Tensor, c: float = 1.0, *,
                   sign_convention: Optional[str] = None) -> torch.Tensor:
    """Reference ``:141-178``: all-pairs distances ``[B1, B2]`` without the ``(B1, B2, d+1)`` temporary."""
    return device_batch_distance(x, y, float(c), _sign(sign_convention))


def batch_distance_optimized(x: torch.Tensor, y: torch.Tensor, c: float = 1.0, *,
                             sign_convention: Optional[str] = None) -> torch.Tensor:
    """Reference ``:181-210`` (einsum form; same values as ``batch_distance`` here)."""
    return device_batch_distance(x, y, float(c), _sign(sign_convention))


def parallel_transport(v: torch.Tensor, x: torch.Tensor, y: torch.Tensor, c: float = 1.0, *,
                       sign_convention: Optional[str] = None) -> torch.Tensor:
    """Reference ``:213-228`` (not on the hot path)."""
    xy = -minkowski_dot(x, y, sign_convention=sign_convention).unsqueeze(-1)
    coef = minkowski_dot(y, v, sign_convention=sign_convention).unsqueeze(-1) / (1 - xy)
    return v + coef * (x + y)


def riemannian_gradient(euclidean_grad: torch.Tensor, x: torch.Tensor, c: float = 1.0, *,
                        sign_convention: Optional[str] = None) -> torch.Tensor:
    """Reference ``:231-244`` (dead code in the reference; not on the hot path)."""
    return euclidean_grad + minkowski_dot(x, euclidean_grad, sign_convention=sign_convention).unsqueeze(-1) * x
