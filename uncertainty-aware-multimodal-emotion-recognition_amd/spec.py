"""Static description of the fusion + DEER hot path: dimensions and the
canonical parameter table.

The table is the single source of truth shared by the host module
(``model.py``), the C-ABI (``include/mmdeer.h`` -- the ``MMDEER_P_*`` enum is
generated from the same order, see ``csrc/params.inc``) and the oracle.

Names are the reference's ``state_dict`` keys:
  fusion.*  <- HierarchicalMultimodalFusion   (reference src/models/fusion.py:47-106)
  head.*    <- MultiDimensionalDEER           (reference src/models/deer.py:201-231)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

DIM_NAMES = ("valence", "arousal", "dominance")


@dataclass(frozen=True)
class Dims:
    """Layer widths of the path (reference fusion.py:47-50, deer.py:201-202)."""

    audio: int = 84
    video: int = 256
    text: int = 768
    inter: int = 256  # intermediate_dim: AV stage width
    fusion: int = 512
    heads: int = 8
    hidden: int = 256  # MultiDimensionalDEER hidden_dim
    evid1: int = 128  # DEERLayer hidden = hidden // 2
    evid2: int = 64  # DEERLayer hidden // 2
    ndim: int = 3  # valence, arousal, dominance
    dropout: float = 0.3

    @property
    def audio_pad(self) -> int:
        """Audio width padded to the MFMA K granule (bf16 16x16x32)."""
        return (self.audio + 31) // 32 * 32


DEFAULT_DIMS = Dims()

# init kinds (reference fusion.py:108-117, deer.py:61-66 and torch defaults)
XAVIER = "xavier"  # xavier_uniform_ weight
ZERO = "zero"  # zero bias
ONE = "one"  # LayerNorm weight
KAIMING = "kaiming"  # torch default nn.Linear weight init (kaiming_uniform a=sqrt5)
BIAS_DEFAULT = "bias_default"  # torch default nn.Linear bias U(+-1/sqrt(fan_in))


def param_table(d: Dims = DEFAULT_DIMS) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Canonical (name, shape, init) list of the LIVE parameters, in ABI order."""
    t: List[Tuple[str, Tuple[int, ...], str]] = []
    av = "fusion.audio_visual_fusion."
    tri = "fusion.trimodal_fusion."
    t += [
        (av + "audio_projection.weight", (d.inter, d.audio), XAVIER),
        (av + "audio_projection.bias", (d.inter,), ZERO),
        (av + "video_projection.weight", (d.inter, d.video), XAVIER),
        (av + "video_projection.bias", (d.inter,), ZERO),
        (av + "cross_attention.in_proj_weight", (3 * d.inter, d.inter), XAVIER),
        (av + "cross_attention.in_proj_bias", (3 * d.inter,), ZERO),
        (av + "cross_attention.out_proj.weight", (d.inter, d.inter), XAVIER),
        (av + "cross_attention.out_proj.bias", (d.inter,), ZERO),
        (av + "fusion_layers.0.weight", (d.inter, 2 * d.inter), XAVIER),
        (av + "fusion_layers.0.bias", (d.inter,), ZERO),
        (av + "fusion_layers.3.weight", (d.inter,), ONE),
        (av + "fusion_layers.3.bias", (d.inter,), ZERO),
        (tri + "audiovisual_projection.weight", (d.fusion, d.inter), XAVIER),
        (tri + "audiovisual_projection.bias", (d.fusion,), ZERO),
        (tri + "text_projection.weight", (d.fusion, d.text), XAVIER),
        (tri + "text_projection.bias", (d.fusion,), ZERO),
        (tri + "modality_attention.in_proj_weight", (3 * d.fusion, d.fusion), XAVIER),
        (tri + "modality_attention.in_proj_bias", (3 * d.fusion,), ZERO),
        (tri + "modality_attention.out_proj.weight", (d.fusion, d.fusion), XAVIER),
        (tri + "modality_attention.out_proj.bias", (d.fusion,), ZERO),
        (tri + "final_fusion.0.weight", (d.fusion, d.fusion), XAVIER),
        (tri + "final_fusion.0.bias", (d.fusion,), ZERO),
        (tri + "final_fusion.3.weight", (d.fusion,), ONE),
        (tri + "final_fusion.3.bias", (d.fusion,), ZERO),
        ("fusion.output_projection.0.weight", (d.fusion, d.fusion), XAVIER),
        ("fusion.output_projection.0.bias", (d.fusion,), ZERO),
        ("fusion.output_projection.3.weight", (d.fusion,), ONE),
        ("fusion.output_projection.3.bias", (d.fusion,), ZERO),
        ("head.feature_processor.0.weight", (d.hidden, d.fusion), KAIMING),
        ("head.feature_processor.0.bias", (d.hidden,), BIAS_DEFAULT),
        ("head.feature_processor.3.weight", (d.hidden, d.hidden), KAIMING),
        ("head.feature_processor.3.bias", (d.hidden,), BIAS_DEFAULT),
    ]
    # The three DEERLayer heads are listed layer-major (all heads' layer 0, then
    # layer 3, then layer 6) so that the packed weight / flat gradient buffers
    # hold [3*128,256], [3,64,128] and [3,4,64] blocks contiguously: the head
    # GEMMs run as one stacked / strided-batched launch.
    for layer, shp_w, shp_b in (
        ("0", (d.evid1, d.hidden), (d.evid1,)),
        ("3", (d.evid2, d.evid1), (d.evid2,)),
        ("6", (4, d.evid2), (4,)),
    ):
        for h in range(d.ndim):
            t.append((f"head.deer_heads.{h}.evidence_net.{layer}.weight", shp_w, XAVIER))
        for h in range(d.ndim):
            t.append((f"head.deer_heads.{h}.evidence_net.{layer}.bias", shp_b, ZERO))
    return t


def gate_param_table(d: Dims = DEFAULT_DIMS) -> List[Tuple[str, Tuple[int, ...], str]]:
    """``uncertainty_gate.*`` -- kept for state_dict compatibility only.

    Unreachable in the reference (fusion.py:148-150 vs :384 raises TypeError), never
    receives a gradient; not part of the ABI table.  SURVEY 8a row a4.
    """
    g = "fusion.uncertainty_gate."
    t: List[Tuple[str, Tuple[int, ...], str]] = []
    for i, dim in enumerate((d.audio, d.video, d.text)):
        t += [
            (g + f"modality_encoders.{i}.0.weight", (128, dim), XAVIER),
            (g + f"modality_encoders.{i}.0.bias", (128,), ZERO),
            (g + f"modality_encoders.{i}.2.weight", (64, 128), XAVIER),
            (g + f"modality_encoders.{i}.2.bias", (64,), ZERO),
        ]
    t += [
        (g + "uncertainty_encoder.0.weight", (64, 3), XAVIER),
        (g + "uncertainty_encoder.0.bias", (64,), ZERO),
        (g + "uncertainty_encoder.2.weight", (32, 64), XAVIER),
        (g + "uncertainty_encoder.2.bias", (32,), ZERO),
        (g + "gating_network.0.weight", (128, 224), XAVIER),
        (g + "gating_network.0.bias", (128,), ZERO),
        (g + "gating_network.2.weight", (3, 128), XAVIER),
        (g + "gating_network.2.bias", (3,), ZERO),
    ]
    return t


def param_offsets(d: Dims = DEFAULT_DIMS):
    """Element offsets of each live parameter inside the flat (packed / gradient)
    buffers.  Every tensor starts on a 64-element boundary so 16-byte vector
    access is aligned for both fp32 and bf16 storage."""
    offs = []
    cur = 0
    for name, shape, _ in param_table(d):
        n = 1
        for s in shape:
            n *= s
        offs.append(cur)
        cur += (n + 63) // 64 * 64
    return offs, cur


def n_live_params(d: Dims = DEFAULT_DIMS) -> int:
    n = 0
    for _, shape, _ in param_table(d):
        c = 1
        for s in shape:
            c *= s
        n += c
    return n
