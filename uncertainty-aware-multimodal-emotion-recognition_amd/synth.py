"""Counter-based synthetic data and closed-form parameter fills.

Everything here is a pure function of (stream id, element index): no RNG state,
no dependence on generation order, so this container (where the golden vectors
are captured from the imported reference) and the GPU box (where they are
replayed) produce bit-identical inputs and weights.  SURVEY 8c "golden-vector
plan", 8d "synthetic inputs".

Input recipe mirrors the reference's synthetic loader
(experiments/run_multimodal_deer.py:329-338): audio/video/text ~ N(0,1),
targets = tanh(z + 0.1 n).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

from .spec import (
    BIAS_DEFAULT,
    DEFAULT_DIMS,
    KAIMING,
    ONE,
    XAVIER,
    ZERO,
    Dims,
    gate_param_table,
    param_table,
)

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """SplitMix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


def uniform01(stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n doubles in [0,1): element i of stream s = splitmix64(key(s) + offset + i) >> 11, where
    key(s) = splitmix64(s * golden) is a full 64-bit mix of the (arbitrary non-negative) stream id."""
    key = _splitmix64(np.array([(int(stream) * 0x9E3779B97F4A7C15 + 0x1234567) & 0xFFFFFFFFFFFFFFFF],
                               dtype=np.uint64))[0]
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64((idx + key) & _M64)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n N(0,1) doubles by Box-Muller over two decorrelated counter streams."""
    u1 = uniform01(stream * 2 + 0, n, offset)
    u2 = uniform01(stream * 2 + 1, n, offset)
    r = np.sqrt(-2.0 * np.log1p(-u1))  # log1p(-u) is finite for u in [0,1)
    return r * np.cos(2.0 * math.pi * u2)


def _stream_of(name: str) -> int:
    """Stable 30-bit stream id from a parameter name (FNV-1a)."""
    h = 0x811C9DC5
    for c in name.encode():
        h = ((h ^ c) * 0x01000193) & 0xFFFFFFFF
    return h & 0x3FFFFFFF


def module_fill(tag: str, shapes) -> "dict[str, np.ndarray]":
    """Closed-form fp32 parameters for an arbitrary module, keyed by state_dict name (``shapes``: name -> shape).

    Matrices: Xavier-range uniforms; ``*.weight`` vectors (LayerNorm gamma) and ``*temperature``: 1 + 0.1 u; other
    vectors (biases): 0.05 u.  ``tests/golden/make_golden.py`` fills the reference modules with exactly this."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(int(d) for d in shape)
        n = int(np.prod(shape))
        u = uniform01(_stream_of(tag + "." + name), n) * 2.0 - 1.0
        if len(shape) >= 2:
            w = u * np.sqrt(6.0 / (shape[0] + shape[1]))
        elif name.endswith("weight") or name.endswith("temperature"):
            w = 1.0 + 0.1 * u
        else:
            w = 0.05 * u
        out[name] = w.reshape(shape).astype(np.float32)
    return out


def make_batch(batch: int, seed: int = 42, dims: Dims = DEFAULT_DIMS,
               row_offset: int = 0) -> Dict[str, np.ndarray]:
    """Synthetic (audio, video, text, targets) float32 batch.

    ``row_offset`` lets a data-parallel rank draw rows [row_offset, row_offset+batch)
    of the same global stream, so N ranks x B rows == one N*B batch.
    """
    out = {}
    for k, (name, width) in enumerate(
        (("audio", dims.audio), ("video", dims.video), ("text", dims.text))
    ):
        x = normal(seed * 16 + k, batch * width, row_offset * width)
        out[name] = x.reshape(batch, width).astype(np.float32)
    z = normal(seed * 16 + 8, batch * dims.ndim, row_offset * dims.ndim)
    n = normal(seed * 16 + 9, batch * dims.ndim, row_offset * dims.ndim)
    out["targets"] = np.tanh(z + 0.1 * n).reshape(batch, dims.ndim).astype(np.float32)
    return out


def _fans(shape: Tuple[int, ...]) -> Tuple[int, int]:
    if len(shape) == 2:
        return shape[1], shape[0]
    return shape[0], shape[0]


def closed_form_state(dims: Dims = DEFAULT_DIMS, variant: int = 0,
                      include_gate: bool = False) -> Dict[str, np.ndarray]:
    """Closed-form deterministic parameters (float32), keyed by canonical name.

    Unlike the reference's init (zero biases, unit LayerNorm) every tensor is
    non-trivial so the parity fixtures exercise bias, gamma and beta paths:
      weights  ~ U(-b, b), b = Xavier bound sqrt(6/(fan_in+fan_out))
      biases   ~ U(-0.05, 0.05)
      LN gamma ~ 1 + U(-0.1, 0.1);  LN beta ~ U(-0.05, 0.05)
    """
    table = list(param_table(dims))
    if include_gate:
        table += gate_param_table(dims)
    sd: Dict[str, np.ndarray] = {}
    for name, shape, kind in table:
        n = int(np.prod(shape))
        u = uniform01(_stream_of(name) + (variant << 32), n) * 2.0 - 1.0
        if kind in (XAVIER, KAIMING):
            fan_in, fan_out = _fans(shape)
            w = u * math.sqrt(6.0 / (fan_in + fan_out))
        elif kind == ONE:
            w = 1.0 + 0.1 * u
        elif kind in (ZERO, BIAS_DEFAULT):
            w = 0.05 * u
        else:  # pragma: no cover
            raise ValueError(kind)
        sd[name] = w.reshape(shape).astype(np.float32)
    return sd


def reference_init_state(dims: Dims = DEFAULT_DIMS, seed: int = 0,
                         include_gate: bool = True) -> Dict[str, np.ndarray]:
    """Parameters distributed as the reference initialises them (fusion.py:108-117:
    Xavier-uniform Linear weights, zero biases, LayerNorm (1, 0); deer.py:61-66 the
    same for DEERLayer; ``feature_processor`` keeps torch's default nn.Linear init:
    kaiming_uniform(a=sqrt 5) == U(+-1/sqrt(fan_in)) for weight and bias), drawn
    from the counter-based generator instead of torch's global RNG."""
    table = list(param_table(dims))
    if include_gate:
        table += gate_param_table(dims)
    sd: Dict[str, np.ndarray] = {}
    for name, shape, kind in table:
        n = int(np.prod(shape))
        u = uniform01(_stream_of(name) + ((seed + 7) << 32), n) * 2.0 - 1.0
        fan_in, fan_out = _fans(shape)
        if kind == XAVIER:
            w = u * math.sqrt(6.0 / (fan_in + fan_out))
        elif kind == KAIMING:
            w = u / math.sqrt(fan_in)
        elif kind == BIAS_DEFAULT:
            # fan_in of the owning Linear: read it from the weight registered just before
            w = u / math.sqrt(sd[name[: -len("bias")] + "weight"].shape[1])
        elif kind == ONE:
            w = np.ones(n)
        else:
            w = np.zeros(n)
        sd[name] = np.asarray(w, dtype=np.float64).reshape(shape).astype(np.float32)
    return sd


FUSION_ALT_DIMS = [84, 256, 768]


def fusion_alt_inputs(tag: str, out_shape=None, B: int = 9):
    """Closed-form inputs (and, given the output shape, the loss weights c) of the a5 golden cases
    (tests/golden/make_golden.py: capture_fusion_alt)."""
    dims = [256, 256, 256] if tag == "factory_attention" else (FUSION_ALT_DIMS[:2] if tag == "bilinear2" else FUSION_ALT_DIMS)
    xs = [normal(700 + 10 * len(tag) + i, B * d).reshape(B, d).astype(np.float32) for i, d in enumerate(dims)]
    c = None
    if out_shape is not None:
        c = normal(990 + len(tag), int(np.prod(out_shape))).reshape(out_shape).astype(np.float32)
    return xs, c
