"""Side rows of the hot-path table (a8, a9, a14), forward / inference only.

Each class keeps its parameters in the same ``nn.Module`` containers as the reference class it mirrors, so
``state_dict()`` keys and shapes interchange with reference checkpoints; the arithmetic runs on the HIP library
(``ops.py``).  Backward is not built for these rows: a8 has no caller in the reference, a9 cannot execute there
(shape bug, parity unpinned beyond its three encoders) and a14 is the evaluation-time feature branch.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import nn

from . import ops


def _xavier_(module: nn.Module) -> None:
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            nn.init.zeros_(m.bias)


class CrossModalAttention(nn.Module):
    """Mirror of ``deer.CrossModalAttention`` (reference src/models/deer.py:353-425).

    ``forward(audio, video, text) -> (weighted_audio, weighted_video)``, each (B, feature_dim // num_heads): the
    reference's softmax runs over the head axis and its weighted sum collapses the heads.  ``output_proj`` exists
    (and is checkpointed) but is never applied, as in the reference."""

    def __init__(self, feature_dim: int = 256, num_heads: int = 8, compute_dtype: str = "fp32"):
        super().__init__()
        if feature_dim != 256 or num_heads != 8:
            raise NotImplementedError("the HIP kernel is specialised for feature_dim=256, num_heads=8")
        self.feature_dim, self.num_heads, self.head_dim = feature_dim, num_heads, feature_dim // num_heads
        self.compute_dtype = compute_dtype
        self.query_proj = nn.Linear(feature_dim, feature_dim)
        self.key_proj = nn.Linear(feature_dim, feature_dim)
        self.value_proj = nn.Linear(feature_dim, feature_dim)
        self.output_proj = nn.Linear(feature_dim, feature_dim)
        self.uncertainty_gate = nn.Sequential(nn.Linear(feature_dim * 3, feature_dim), nn.ReLU(),
                                              nn.Linear(feature_dim, 2), nn.Softmax(dim=1))

    @torch.no_grad()
    def forward(self, audio: torch.Tensor, video: torch.Tensor, text: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        for t in (audio, video, text):
            if t.dim() != 2 or t.shape[1] != self.feature_dim:
                raise ValueError(f"expected (B, {self.feature_dim}) features, got {tuple(t.shape)}")
        c = self.compute_dtype
        B = audio.shape[0]
        q = ops.linear(text, self.query_proj.weight, self.query_proj.bias, compute=c)
        # key / value projections of both modalities in one GEMM each: rows [0,B) audio, [B,2B) video
        av = torch.cat([audio, video], dim=0)
        k = ops.linear(av, self.key_proj.weight, self.key_proj.bias, compute=c)
        v = ops.linear(av, self.value_proj.weight, self.value_proj.bias, compute=c)
        ctx = torch.cat([audio, video, text], dim=1)
        g = ops.linear(ctx, self.uncertainty_gate[0].weight, self.uncertainty_gate[0].bias, relu=True, compute=c)
        # 256 -> 2 logits: N = 2 is below the GEMM's column granule; pad the weight to 4 rows
        w2 = torch.zeros(4, self.feature_dim, dtype=torch.float32, device=audio.device)
        b2 = torch.zeros(4, dtype=torch.float32, device=audio.device)
        w2[:2] = self.uncertainty_gate[2].weight
        b2[:2] = self.uncertainty_gate[2].bias
        logits = ops.linear(g, w2, b2, compute=c)[:, :2].float()
        return ops.cross_modal_attention_core(q, k[:B], v[:B], k[B:], v[B:], logits)


class ModalityEncoders(nn.Module):
    """The three ``ReLU(Linear)`` encoders of ``deer.HierarchicalDEERFusion`` (reference src/models/deer.py:287-289,
    330-332).  The rest of that class cannot execute in the reference (its ``av_fusion`` expects 512 features and
    receives 64), so only this part is built; parameter names match the reference's."""

    def __init__(self, audio_dim: int = 84, video_dim: int = 256, text_dim: int = 768, hidden_dim: int = 256,
                 compute_dtype: str = "fp32"):
        super().__init__()
        self.compute_dtype = compute_dtype
        self.audio_encoder = nn.Linear(audio_dim, hidden_dim)
        self.video_encoder = nn.Linear(video_dim, hidden_dim)
        self.text_encoder = nn.Linear(text_dim, hidden_dim)

    @torch.no_grad()
    def forward(self, audio, video, text):
        c = self.compute_dtype
        enc = lambda x, m: ops.linear(x, m.weight, m.bias, relu=True, compute=c)   # noqa: E731
        return enc(audio, self.audio_encoder), enc(video, self.video_encoder), enc(text, self.text_encoder)


class EnhancedAudioEncoder(nn.Module):
    """Feature branch of ``encoders.EnhancedAudioEncoder`` (reference src/models/encoders.py:65-126, 356-389) in
    evaluation mode: (B, 84) or (B, 1, 84) pre-extracted features -> (B, 512).

    With one time step and zero initial state the recurrent weights multiply zeros and the attention pool's
    softmax over time is 1, so the path is: per layer one GEMM against the stacked ``[forward; reverse]`` input
    weights (bias ``b_ih + b_hh``) + the gate kernel, then ``output_projection``.  Raw-waveform input (librosa
    feature extraction on the host) and T > 1 are outside this path."""

    def __init__(self, config: Optional[Dict] = None, compute_dtype: str = "fp32"):
        super().__init__()
        config = config or {}
        self.hidden_dim = config.get("hidden_dim", 512)
        self.num_layers = config.get("num_layers", 2)
        self.dropout = config.get("dropout", 0.3)
        self.bidirectional = config.get("bidirectional", True)
        self.enhanced_features_dim = 84
        self.compute_dtype = compute_dtype
        if not self.bidirectional or self.hidden_dim % 8:
            raise NotImplementedError("built for the reference default: bidirectional, hidden_dim % 8 == 0")
        self.lstm = nn.LSTM(input_size=84, hidden_size=self.hidden_dim // 2, num_layers=self.num_layers, batch_first=True,
                            dropout=self.dropout if self.num_layers > 1 else 0, bidirectional=True)
        self.attention = nn.Sequential(nn.Linear(self.hidden_dim, self.hidden_dim // 2), nn.Tanh(),
                                       nn.Linear(self.hidden_dim // 2, 1), nn.Softmax(dim=1))
        self.output_projection = nn.Sequential(nn.Linear(self.hidden_dim, self.hidden_dim), nn.ReLU(), nn.Dropout(self.dropout),
                                               nn.Linear(self.hidden_dim, self.hidden_dim), nn.LayerNorm(self.hidden_dim))
        _xavier_(self)
        for name, p in self.lstm.named_parameters():   # encoders.py:120-126
            if "weight" in name:
                nn.init.xavier_uniform_(p.data)
            else:
                nn.init.zeros_(p.data)

    @torch.no_grad()
    def forward(self, audio_input: torch.Tensor) -> torch.Tensor:
        x = audio_input
        if x.shape[-1] != self.enhanced_features_dim:
            raise NotImplementedError("raw-waveform input (host-side librosa feature extraction) is outside the hot path")
        if x.dim() == 3:
            if x.shape[1] != 1:
                raise NotImplementedError("the HIP path covers the single-time-step feature branch (T = 1)")
            x = x[:, 0]
        if x.dim() != 2:
            raise ValueError(f"expected (B, 84) or (B, 1, 84), got {tuple(audio_input.shape)}")
        if self.training:
            raise NotImplementedError("side rows are inference-only: call .eval() first")
        c = self.compute_dtype
        H = self.hidden_dim // 2
        h = x
        for layer in range(self.num_layers):
            w = torch.cat([getattr(self.lstm, f"weight_ih_l{layer}"), getattr(self.lstm, f"weight_ih_l{layer}_reverse")], dim=0)
            b = torch.cat([getattr(self.lstm, f"bias_ih_l{layer}") + getattr(self.lstm, f"bias_hh_l{layer}"),
                           getattr(self.lstm, f"bias_ih_l{layer}_reverse") + getattr(self.lstm, f"bias_hh_l{layer}_reverse")])
            gates = ops.linear(h, w, b, compute=c)                 # (B, 2 * 4H)
            h = ops.lstm_cell_t1(gates, H, 2)                      # (B, 2H) = [forward | reverse]
        op = self.output_projection
        y = ops.linear(h, op[0].weight, op[0].bias, relu=True, compute=c)
        y = ops.linear(y, op[3].weight, op[3].bias, compute=c)
        return ops.layer_norm(y, op[4].weight, op[4].bias)
