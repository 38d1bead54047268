"""ctypes binding of libmmdeer_hip.so (include/mmdeer.h).

There is NO fallback: if the library cannot be loaded (or built from the in-tree
sources with hipcc) every entry point raises.  The product path never touches
``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

# torch must be imported BEFORE libmmdeer_hip.so is dlopen'ed: the PyTorch-ROCm wheel ships its own
# libamdhip64.so, and the library has to bind to that already-loaded runtime (one HIP runtime per process,
# shared streams and device pointers) rather than pull in a second copy from /opt/rocm/lib.
# kernel arguments in device memory: a kernarg fetch from host memory costs a PCIe round trip per dependent
# scalar load (measured 8.7k vs 2.4k cycles of kernel setup, 0.69 vs 0.53 ms per train step).  Read at HIP init.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: F401,E402

from . import build as _build

_LIB: Optional[C.CDLL] = None
_LOCK = threading.Lock()

c_void_p, c_int, c_float, c_u64, c_size_t, c_ll = C.c_void_p, C.c_int32, C.c_float, C.c_uint64, C.c_size_t, C.c_longlong
c_char_p = C.c_char_p


class LossCfg(C.Structure):
    _fields_ = [("reg_weight", c_float), ("kl_weight", c_float), ("ece_weight", c_float),
                ("cross_weight", c_float), ("task_weight", c_float * 3)]


class ForwardArgs(C.Structure):
    _fields_ = [
        ("batch", c_int), ("compute_f32", c_int), ("training", c_int), ("inputs_bf16", c_int), ("repack", c_int),
        ("dropout_p", c_float), ("seed", c_u64), ("offset", c_u64), ("offset_dev", c_void_p), ("bump_offset_dev", c_int),
        ("audio", c_void_p), ("video", c_void_p), ("text", c_void_p),
        ("params", C.POINTER(c_void_p)),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t), ("weights", c_void_p), ("weights_bytes", c_size_t),
        ("nig_out", c_void_p), ("fused_features", c_void_p), ("audiovisual_features", c_void_p),
        ("trimodal_features", c_void_p), ("av_attention", c_void_p), ("trimodal_attention", c_void_p),
        ("targets", c_void_p), ("prof_events", c_void_p * 2), ("stream", c_void_p),
    ]


class BackwardArgs(C.Structure):
    _fields_ = [
        ("batch", c_int), ("compute_f32", c_int), ("training", c_int), ("inputs_bf16", c_int),
        ("dropout_p", c_float), ("seed", c_u64), ("offset", c_u64), ("offset_dev", c_void_p),
        ("audio", c_void_p), ("video", c_void_p), ("text", c_void_p),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t), ("weights", c_void_p), ("weights_bytes", c_size_t),
        ("targets", c_void_p), ("g_mu", c_void_p), ("g_nu", c_void_p), ("g_alpha", c_void_p), ("g_beta", c_void_p),
        ("loss", LossCfg),
        ("grads", c_void_p), ("loss_out", c_void_p), ("bin_counts", c_void_p),
        ("bucket_events", c_void_p * 3), ("phase", c_int), ("bump_offset_dev", c_int), ("global_stats", c_void_p), ("g_fused", c_void_p), ("stream", c_void_p),
    ]


class AdamWArgs(C.Structure):
    _fields_ = [
        ("compute_f32", c_int), ("pack_transposed", c_int), ("step", c_int),
        ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("weight_decay", c_float),
        ("max_grad_norm", c_float), ("grad_scale", c_float),
        ("lr", C.POINTER(c_float)), ("params", C.POINTER(c_void_p)), ("grads", c_void_p),
        ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("grad_norm", c_void_p),
        ("weights", c_void_p), ("weights_bytes", c_size_t), ("stream", c_void_p),
    ]


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("C", c_void_p), ("bias", c_void_p), ("bias_grad", c_void_p), ("Y", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int), ("lda", c_int), ("ldw", c_int), ("ldc", c_int), ("ldy", c_int),
        ("a_f32", c_int), ("w_f32", c_int), ("c_f32", c_int), ("y_f32", c_int), ("trans_a", c_int), ("trans_w", c_int),
        ("relu", c_int), ("accumulate", c_int), ("compute_f32", c_int), ("tile", c_int),
        ("drop_site", c_int), ("drop_shift", c_int), ("regen_site", c_int),
        ("dropout_p", c_float), ("mask_scale", c_float), ("seed", c_u64), ("offset", c_u64), ("offset_dev", c_void_p),
        ("splitk", c_int), ("slab", c_void_p), ("debug", c_void_p), ("stream", c_void_p),
    ]


class AdamWFlatArgs(C.Structure):
    _fields_ = [
        ("params", c_void_p), ("grads", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p),
        ("packed", c_void_p), ("packed_f32", c_int), ("flat_elems", c_ll),
        ("nseg", c_int), ("seg_begin", C.POINTER(c_ll)), ("seg_elems", C.POINTER(c_ll)), ("seg_lr", C.POINTER(c_float)),
        ("scratch", c_void_p), ("grad_norm", c_void_p), ("step", c_int),
        ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("weight_decay", c_float), ("max_grad_norm", c_float),
        ("grad_scale", c_float), ("stream", c_void_p),
    ]


class StackBAttnArgs(C.Structure):
    _fields_ = [
        ("h2", c_void_p), ("pre", c_void_p), ("self_out", c_void_p), ("cross_out", c_void_p),
        ("est_w3", c_void_p), ("est_b3", c_void_p), ("wn_w1_unc", c_void_p), ("wn_w2", c_void_p), ("wn_b2", c_void_p),
        ("out_av", c_void_p), ("out_text", c_void_p), ("weights", c_void_p), ("uncertainties", c_void_p),
        ("ld_w1_unc", c_int), ("ld_av", c_int), ("ld_text", c_int), ("B", c_int), ("act_f32", c_int),
        ("stream", c_void_p),
    ]


class SoftmaxMixArgs(C.Structure):
    _fields_ = [
        ("P", c_void_p), ("ldp", C.c_int64), ("sp", C.c_int64), ("S", c_int), ("D", c_int), ("B", c_int), ("act_f32", c_int),
        ("w_att", c_void_p), ("b_att", c_void_p), ("logits", c_void_p), ("ld_logits", c_int), ("weights8", c_void_p),
        ("out", c_void_p), ("ld_out", c_int), ("dout", c_void_p), ("ld_dout", c_int), ("dP", c_void_p), ("dlogits8", c_void_p),
        ("stream", c_void_p),
    ]


class StackBAttnTrainArgs(C.Structure):
    _fields_ = [
        ("h2", c_void_p), ("pre", c_void_p), ("self_out", c_void_p), ("cross_out", c_void_p),
        ("est_w3", c_void_p), ("est_b3", c_void_p), ("wn_w1_unc", c_void_p), ("wn_w2", c_void_p), ("wn_b2", c_void_p),
        ("out_av", c_void_p), ("out_text", c_void_p), ("r", c_void_p), ("weights4", c_void_p), ("unc4", c_void_p), ("unc8", c_void_p),
        ("d_av", c_void_p), ("d_text", c_void_p), ("d_self", c_void_p), ("d_cross", c_void_p), ("d_pre", c_void_p),
        ("d_logits8", c_void_p), ("d_z8", c_void_p), ("d_h2", c_void_p),
        ("ld_w1_unc", c_int), ("ld_av", c_int), ("ld_text", c_int), ("B", c_int), ("act_f32", c_int), ("ld_dcross", c_int),
        ("training", c_int), ("drop_site", c_int), ("dropout_p", c_float), ("seed", c_u64), ("offset", c_u64),
        ("offset_dev", c_void_p), ("stream", c_void_p),
    ]


class StackBWeights(C.Structure):
    _fields_ = [
        ("audio_dim", c_int), ("video_dim", c_int), ("text_dim", c_int), ("encoder_layers", c_int), ("audio_ld", c_int),
        ("enc_in_w", c_void_p * 3), ("enc_in_vec", c_void_p), ("enc_res_w", c_void_p), ("enc_res_vec", c_void_p),
        ("enc_out_w", c_void_p), ("enc_out_b", c_void_p), ("value_w", c_void_p), ("value_b", c_void_p),
        ("attn_out_w", c_void_p), ("attn_out_b", c_void_p),
        ("est_w1", c_void_p), ("est_b1", c_void_p), ("est_w2", c_void_p), ("est_b2", c_void_p), ("est_w3", c_void_p), ("est_b3", c_void_p),
        ("wn_w1", c_void_p), ("wn_b1", c_void_p), ("wn_w1_unc", c_void_p), ("wn_w2", c_void_p), ("wn_b2", c_void_p),
        ("av_w0", c_void_p), ("av_w4", c_void_p), ("av_vec", c_void_p),
        ("tri_w0", c_void_p), ("tri_w4", c_void_p), ("tri_vec", c_void_p),
        ("gate_w", c_void_p), ("gate_b", c_void_p),
        ("head_w0", c_void_p), ("head_b0", c_void_p), ("head_w3", c_void_p), ("head_b3", c_void_p), ("head_w6", c_void_p), ("head_b6", c_void_p),
        ("calibration", c_void_p * 7),
    ]


class StackBForwardArgs(C.Structure):
    _fields_ = [
        ("batch", c_int), ("compute_f32", c_int),
        ("audio", c_void_p), ("video", c_void_p), ("text", c_void_p),
        ("weights", C.POINTER(StackBWeights)),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("planes", c_void_p), ("attention_weights", c_void_p), ("modality_uncertainties", c_void_p), ("fused_features", c_void_p),
        ("stream", c_void_p),
    ]


class ChainSeg(C.Structure):
    """mmdeer_chain_seg (include/mmdeer.h)."""
    _fields_ = [
        ("W", c_void_p), ("bias", c_void_p), ("N", c_int), ("K", c_int), ("kin_off", c_int), ("nout_off", c_int), ("relu", c_int),
        ("drop_site", c_int), ("drop_shift", c_int), ("dcol_off", c_int),
        ("mask_y", c_void_p), ("ld_mask", c_int), ("mask_col0", c_int), ("mask_scale", c_float),
        ("res_add", c_int), ("res_dup", c_int), ("end_layer", c_int), ("nout", c_int),
        ("stash", c_void_p), ("ld_stash", c_int), ("stash2", c_void_p), ("stash_split", c_int),
        ("gamma", c_void_p), ("beta", c_void_p), ("xln", c_void_p), ("mean", c_void_p), ("rstd", c_void_p), ("residual", c_int),
        ("lnb_gamma", c_void_p), ("lnb_y", c_void_p), ("lnb_mean", c_void_p), ("lnb_rstd", c_void_p), ("lnb_dz", c_void_p),
        ("lnb_partial", c_void_p), ("lnb_mask_scale", c_float),
    ]


CHAIN_MAX_SEGS = 12


class ChainArgs(C.Structure):
    """mmdeer_chain_args (include/mmdeer.h)."""
    _fields_ = [
        ("X", c_void_p), ("ldx", c_int), ("K0", c_int), ("rows", c_int), ("samples_per_workgroup", c_int), ("nseg", c_int),
        ("dropout_p", c_float), ("seed", c_u64), ("offset", c_u64), ("offset_dev", c_void_p),
        ("seg", ChainSeg * CHAIN_MAX_SEGS), ("debug", c_void_p), ("stream", c_void_p),
    ]


class RepackJob(C.Structure):
    """mmdeer_repack_job (include/mmdeer.h)."""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("ld_src", c_int), ("rows", c_int), ("cols", c_int), ("cols_valid", c_int),
                ("transpose", c_int), ("layout", c_int), ("ld_dst", c_int), ("dst_col", c_int)]


# ctypes mirror of every argument struct, by the name mmdeer_sizeof() knows it under
STRUCTS = {"gemm_args": GemmArgs, "chain_args": ChainArgs, "chain_seg": ChainSeg, "repack_job": RepackJob, "forward_args": ForwardArgs,
           "backward_args": BackwardArgs, "adamw_args": AdamWArgs, "adamw_flat_args": AdamWFlatArgs, "stackb_attn_train_args": StackBAttnTrainArgs,
           "stackb_attn_args": StackBAttnArgs, "stackb_forward_args": StackBForwardArgs, "stackb_weights": StackBWeights,
           "softmax_mix_args": SoftmaxMixArgs}

# every symbol include/mmdeer.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("mmdeer_version", C.c_char_p, []),
    ("mmdeer_abi_version", c_int, []),
    ("mmdeer_last_error", C.c_char_p, []),
    ("mmdeer_num_params", c_int, []),
    ("mmdeer_param_name", C.c_char_p, [c_int]),
    ("mmdeer_param_rows", c_int, [c_int]),
    ("mmdeer_param_cols", c_int, [c_int]),
    ("mmdeer_param_offset", c_ll, [c_int]),
    ("mmdeer_flat_elems", c_ll, []),
    ("mmdeer_workspace_bytes", c_size_t, [c_int, c_int]),
    ("mmdeer_weights_bytes", c_size_t, [c_int]),
    ("mmdeer_forward", c_int, [C.POINTER(ForwardArgs)]),
    ("mmdeer_backward", c_int, [C.POINTER(BackwardArgs)]),
    ("mmdeer_loss_stats", c_int, [c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p]),
    ("mmdeer_bucket_begin", c_ll, [c_int]),
    ("mmdeer_bucket_end", c_ll, [c_int]),
    ("mmdeer_gemm", c_int, [C.POINTER(GemmArgs)]),
    ("mmdeer_sizeof", c_ll, [c_char_p]),
    ("mmdeer_chain", c_int, [C.POINTER(ChainArgs)]),
    ("mmdeer_chain_workgroups", c_int, [c_int, c_int]),
    ("mmdeer_repack", c_int, [C.POINTER(RepackJob), c_int, c_void_p]),
    ("mmdeer_gemm_batch_slab_elems", c_ll, [C.POINTER(GemmArgs), c_int]),
    ("mmdeer_gemm_batch", c_int, [C.POINTER(GemmArgs), c_int, c_void_p, c_ll, c_void_p]),
    ("mmdeer_adamw_flat", c_int, [C.POINTER(AdamWFlatArgs)]),
    ("mmdeer_reduce_batch", c_int, [c_int, C.POINTER(c_void_p), C.POINTER(c_void_p), C.POINTER(c_int), C.POINTER(c_int), C.POINTER(c_ll), c_void_p]),
    ("mmdeer_pack_transposed_batch", c_int, [c_int, C.POINTER(c_void_p), C.POINTER(c_int), C.POINTER(c_int), c_void_p, C.POINTER(c_ll),
                                             C.POINTER(c_int), C.POINTER(c_int), c_int, c_void_p]),
    ("mmdeer_layernorm_fwd", c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p]),
    ("mmdeer_layernorm_bwd_nparts", c_int, [c_int]),
    ("mmdeer_layernorm_bwd", c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_float, c_void_p]),
    ("mmdeer_trimodal_attn_fwd", c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_float, c_u64, c_u64, c_void_p]),
    ("mmdeer_trimodal_attn_bwd", c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_float, c_u64, c_u64, c_void_p]),
    ("mmdeer_pack_qkv_headmajor", c_int, [c_void_p] * 3),
    ("mmdeer_trimodal_fused_fwd", c_int, [c_void_p] * 8 + [c_int, c_int, c_float, c_u64, c_u64, c_void_p]),
    ("mmdeer_trimodal_fused_bwd", c_int, [c_void_p] * 6 + [c_int, c_int, c_float, c_u64, c_u64, c_void_p]),
    ("mmdeer_nig_stats_elems", c_ll, [c_int]),
    ("mmdeer_nig_loss", c_int, [c_void_p] * 12 + [c_int, C.POINTER(LossCfg), c_void_p]),
    ("mmdeer_deer_loss_v1_scratch", c_ll, [c_ll]),
    ("mmdeer_deer_loss_v1", c_int, [c_void_p] * 5 + [c_ll, c_float, c_float] + [c_void_p] * 6 + [c_void_p]),
    ("mmdeer_uncertainty_reg_loss", c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("mmdeer_calibration_loss", c_int, [c_void_p] * 4 + [c_ll] + [c_void_p] * 5 + [c_void_p]),
    ("mmdeer_calibration_loss_bins", c_int, [c_void_p] * 4 + [c_ll, C.POINTER(c_float), c_int] + [c_void_p] * 5 + [c_void_p]),
    ("mmdeer_dropout_mask", c_int, [c_int, c_int, c_int, c_float, c_u64, c_u64, c_void_p, c_void_p]),
    ("mmdeer_adamw_step", c_int, [C.POINTER(AdamWArgs)]),
    ("mmdeer_pack_weights", c_int, [C.POINTER(c_void_p), c_void_p, c_size_t, c_int, c_void_p]),
    ("mmdeer_cross_modal_attn_fwd", c_int, [c_void_p] * 5 + [c_int] + [c_void_p] * 3 + [c_int, c_int, c_void_p]),
    ("mmdeer_lstm_cell_t1", c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    ("mmdeer_eval_accumulate", c_int, [c_void_p] * 6 + [c_int, c_void_p]),
    ("mmdeer_eval_quantile_select", c_int, [c_void_p, c_void_p, c_ll, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("mmdeer_eval_ece_bins", c_int, [c_void_p, c_void_p, c_ll, c_void_p, c_int, c_void_p, c_void_p]),
    ("mmdeer_stackb_residual_ln", c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    ("mmdeer_stackb_attn_mix", c_int, [C.POINTER(StackBAttnArgs)]),
    ("mmdeer_stackb_gate_mix", c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    ("mmdeer_stackb_head", c_int, [c_void_p, c_int] + [c_void_p] * 8 + [c_int, c_void_p]),
    ("mmdeer_stackb_attn_mix_train_fwd", c_int, [C.POINTER(StackBAttnTrainArgs)]),
    ("mmdeer_stackb_attn_mix_bwd", c_int, [C.POINTER(StackBAttnTrainArgs)]),
    ("mmdeer_stackb_gate_mix_bwd", c_int, [c_void_p, c_int] * 7 + [c_int, c_int, c_int, c_void_p]),
    ("mmdeer_cross_modal_attn_bwd", c_int, [c_void_p] * 5 + [c_int] + [c_void_p] * 9 + [c_int, c_int, c_void_p]),
    ("mmdeer_lstm_cell_t1_bwd", c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    ("mmdeer_softmax_mix_fwd", c_int, [C.POINTER(SoftmaxMixArgs)]),
    ("mmdeer_softmax_mix_bwd", c_int, [C.POINTER(SoftmaxMixArgs)]),
    ("mmdeer_outer_fwd", c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    ("mmdeer_outer_bwd", c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    ("mmdeer_stackb_head_bwd", c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    ("mmdeer_add_masked", c_int, [c_void_p, c_int] * 4 + [c_float, c_int, c_int, c_int, c_void_p]),
    ("mmdeer_stackb_workspace_bytes", c_size_t, [c_int, c_int, c_int]),
    ("mmdeer_stackb_forward", c_int, [C.POINTER(StackBForwardArgs)]),
    ("mmdeer_comm_unique_id", c_int, [c_void_p]),
    ("mmdeer_comm_init", c_int, [C.POINTER(c_void_p), c_int, c_int, c_void_p]),
    ("mmdeer_comm_destroy", c_int, [c_void_p]),
    ("mmdeer_allreduce", c_int, [c_void_p, c_ll, c_int, c_int, c_void_p, c_void_p]),
    ("mmdeer_comm_rank", c_int, [c_void_p]),
    ("mmdeer_comm_world", c_int, [c_void_p]),
    ("mmdeer_reduce_scatter", c_int, [c_void_p, c_void_p, c_ll, c_int, c_int, c_void_p, c_void_p]),
    ("mmdeer_allgather", c_int, [c_void_p, c_void_p, c_ll, c_int, c_void_p, c_void_p]),
    ("mmdeer_convert", c_int, [c_void_p, c_int, c_void_p, c_int, c_ll, c_void_p]),
    ("mmdeer_trace_begin", c_int, [C.POINTER(c_void_p), c_int]),
    ("mmdeer_trace_end", c_int, []),
    ("mmdeer_trace_label", c_char_p, [c_int]),
    ("mmdeer_set_option", c_int, [c_char_p, c_int]),
    ("mmdeer_get_option", c_int, [c_char_p, C.POINTER(c_int)]),
    ("mmdeer_option_name", c_char_p, [c_int]),
    ("mmdeer_workspace_offset", c_ll, [c_int, c_int, c_char_p]),
    ("mmdeer_weights_offset", c_ll, [c_int, c_char_p]),
]


def lib_path() -> str:
    return _build.LIB_PATH


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load (building first when the in-tree .so is missing or stale)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    with _LOCK:
        if _LIB is not None:
            return _LIB
        path = _build.LIB_PATH
        if build_if_missing and _build.needs_build():
            try:
                _build.build()
            except Exception as e:  # noqa: BLE001
                raise RuntimeError(
                    "libmmdeer_hip.so is missing or older than its sources and could not be (re)built with "
                    "hipcc --offload-arch=gfx950.  There is no CPU fallback for the mmdeer hot path and a stale "
                    f"library is never loaded.  Build error: {e}") from e
        if not os.path.exists(path):
            raise RuntimeError(f"libmmdeer_hip.so not found at {path}; run `python -m mmdeer.build`. "
                               "There is no CPU fallback for the mmdeer hot path.")
        lib = C.CDLL(path)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError here == ABI drift between header and library
            fn.restype = res
            fn.argtypes = args
        if lib.mmdeer_abi_version() != 15:   # MMDEER_ABI_VERSION of include/mmdeer.h
            raise RuntimeError("libmmdeer_hip.so ABI version mismatch")
        for cname, cls in STRUCTS.items():      # the ctypes mirrors against the library's own sizeof (a field added on one side only)
            if lib.mmdeer_sizeof(cname.encode()) != C.sizeof(cls):
                raise RuntimeError(f"mmdeer: ctypes layout of mmdeer_{cname} ({C.sizeof(cls)} bytes) differs from the library's "
                                   f"({lib.mmdeer_sizeof(cname.encode())})")
        # The library itself reads no environment variable (include/mmdeer.h).  For the A/B tools the host forwards
        # MMDEER_<OPTION> (e.g. MMDEER_FUSED_ATTN=0) to mmdeer_set_option once, here.
        i = 0
        while True:
            nm = lib.mmdeer_option_name(i)
            if nm is None:
                break
            env = os.environ.get("MMDEER_" + nm.decode().upper())
            if env is not None:
                if lib.mmdeer_set_option(nm, int(env)) != 0:
                    raise RuntimeError("mmdeer: " + lib.mmdeer_last_error().decode())
            i += 1
        _LIB = lib
    return _LIB


def set_option(name: str, value: int) -> None:
    """mmdeer_set_option: choose a launch plan (include/mmdeer.h lists the names); takes effect from the next call."""
    check(load().mmdeer_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = c_int(0)
    check(load().mmdeer_get_option(name.encode(), C.byref(v)))
    return v.value


class options:
    """``with _lib.options(fused_attn=0): ...`` -- set launch-plan options for a block and restore them afterwards."""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("mmdeer: " + load().mmdeer_last_error().decode())


def ptr(t) -> Optional[int]:
    """data_ptr of a tensor or None."""
    return None if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
