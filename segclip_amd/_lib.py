"""ctypes binding of libsegclip_hip.so (include/segclip_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or a kernel returns an
error, a RuntimeError is raised.  Device pointers come from torch tensors (torch is used for memory
and streams only); nothing of torch crosses the C ABI.
"""
import ctypes as C
import os

import torch

F32, BF16 = 0, 1
ACT_NONE, ACT_QUICK_GELU, ACT_GELU_ERF = 0, 1, 2
GEMM_DEFER_SPLITK, GEMM_DEFER_COLSUM = 1, 2
REDUCE_SLABS, REDUCE_ROWS, REDUCE_MAX = 0, 1, 16
ATTN_FP8 = 1

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsegclip_hip.so")
_lib = None

i64, i32, f32, vp = C.c_int64, C.c_int32, C.c_float, C.c_void_p


class GemmDesc(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("bias", vp), ("residual", vp), ("aux", vp),
                ("M", i64), ("N", i64), ("K", i64), ("sam", i64), ("sak", i64), ("sbn", i64), ("sbk", i64),
                ("ldc", i64), ("ldr", i64), ("ldaux", i64), ("nb1", i64), ("nb2", i64),
                ("bsA1", i64), ("bsA2", i64), ("bsB1", i64), ("bsB2", i64), ("bsC1", i64), ("bsC2", i64),
                ("bsR1", i64), ("bsR2", i64),
                ("a_dtype", i32), ("b_dtype", i32), ("c_dtype", i32), ("r_dtype", i32),
                ("act", i32), ("mul_dact", i32), ("alpha", f32), ("aux_kind", i32),
                ("ws", vp), ("ws_bytes", i64), ("colsum", vp), ("colsum_ws", vp), ("flags", i32), ("res_row_mod", i32)]


class ReduceEntry(C.Structure):
    _fields_ = [("src", vp), ("out0", vp), ("out1", vp), ("out2", vp), ("rows", i64), ("width", i64), ("ld", i64), ("seg", i64),
                ("scale", f32), ("out_dtype", i32)]


class WgradItem(C.Structure):
    _fields_ = [("dy", vp), ("x", vp), ("dw", vp), ("M", i64), ("N", i64), ("ld_dy", i64), ("ld_x", i64), ("ld_dw", i64)]


class TrainCtrl(C.Structure):
    _fields_ = [("grad_sqnorm", f32), ("clip_coef", f32), ("nan_skips", i32), ("steps", i32),
                ("loss_sum", f32), ("last_loss", f32), ("reserved", i32 * 2)]


class AdamWGroup(C.Structure):
    _fields_ = [("lr", C.c_double), ("weight_decay", C.c_double), ("b1", C.c_double), ("b2", C.c_double),
                ("eps", C.c_double), ("warmup", C.c_double), ("lr_start", C.c_double), ("lr_end", C.c_double),
                ("t_total", i64), ("schedule", i32), ("reserved", i32)]


class AdamWTensor(C.Structure):
    _fields_ = [("param", vp), ("grad", vp), ("exp_avg", vp), ("exp_avg_sq", vp), ("shadow_bf16", vp),
                ("n", i64), ("step", i32), ("group", i32)]


class AttnDesc(C.Structure):
    _fields_ = [("Q", vp), ("K", vp), ("V", vp), ("O", vp), ("stats", vp), ("dO", vp), ("dQ", vp), ("dK", vp),
                ("dV", vp), ("ws", vp),
                ("B", i64), ("H", i64), ("Tq", i64), ("Tk", i64), ("hd", i64),
                ("q_sb", i64), ("q_st", i64), ("k_sb", i64), ("k_st", i64), ("v_sb", i64), ("v_st", i64),
                ("o_sb", i64), ("o_st", i64),
                ("dq_sb", i64), ("dq_st", i64), ("dk_sb", i64), ("dk_st", i64), ("dv_sb", i64), ("dv_st", i64),
                ("do_sb", i64), ("do_st", i64),
                ("scale", f32), ("causal", i32), ("dtype", i32), ("flags", i32), ("colsum_part", vp), ("klen", vp)]


class ResBlockFwdDesc(C.Structure):
    _fields_ = [("x", vp), ("ln1w", vp), ("ln1b", vp), ("wqkv", vp), ("bqkv", vp), ("wo", vp), ("bo", vp),
                ("ln2w", vp), ("ln2b", vp), ("wfc", vp), ("bfc", vp), ("wpr", vp), ("bpr", vp),
                ("y1", vp), ("mean1", vp), ("rstd1", vp), ("qkv", vp), ("o", vp), ("stats", vp), ("x1", vp),
                ("y2", vp), ("mean2", vp), ("rstd2", vp), ("h", vp), ("ld_h", i64), ("u", vp), ("ld_u", i64), ("xo", vp),
                ("klen", vp), ("M", i64), ("B", i64), ("T", i64), ("D", i64), ("F", i64), ("H", i64),
                ("eps", f32), ("attn_scale", f32), ("causal", i32), ("act", i32), ("aux_kind", i32), ("x_dtype", i32)]


# name -> (restype, argtypes); every symbol declared in include/segclip_hip.h
SIGNATURES = {
    "segclip_version": (C.c_int, []),
    "segclip_last_error_string": (C.c_char_p, []),
    "segclip_gemm_ws_bytes": (C.c_size_t, [C.POINTER(GemmDesc)]),
    "segclip_gemm_splits": (C.c_int, [C.POINTER(GemmDesc)]),
    "segclip_gemm": (C.c_int, [C.POINTER(GemmDesc), vp]),
    "segclip_reduce_multi": (C.c_int, [C.POINTER(ReduceEntry), C.c_int, C.c_int, vp]),
    "segclip_layernorm_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, i64, f32, C.c_int, C.c_int, vp]),
    "segclip_layernorm_bwd_ws_bytes": (C.c_size_t, [i64, i64]),
    "segclip_layernorm_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, C.c_int, C.c_int, C.c_int, vp]),
    "segclip_wgrad_group_splits": (C.c_int, [i64, i64]),
    "segclip_wgrad_group_model_us": (C.c_double, [i64, i64, C.c_int]),
    "segclip_gemm_pq_half_tail": (C.c_int, [i64]),
    "segclip_wgrad_group_ws_bytes": (C.c_size_t, [vp, C.c_int, C.c_int]),
    "segclip_wgrad_group": (C.c_int, [vp, C.c_int, i64, C.c_int, vp, C.c_size_t, vp]),
    "segclip_layernorm_fwd_multi": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, i64, i64, f32, C.c_int, C.c_int, vp]),
    "segclip_layernorm_bwd_multi_ws_bytes": (C.c_size_t, [i64, i64, C.c_int]),
    "segclip_layernorm_bwd_multi": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, i64, i64, C.c_int, C.c_int, vp]),
    "segclip_layernorm_fwd_seg": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, i64, f32, C.c_int, C.c_int, i64, i64, i64, vp]),
    "segclip_layernorm_bwd_seg": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, C.c_int, C.c_int, C.c_int,
                                            i64, i64, i64, vp]),
    "segclip_attn_stats_bytes": (C.c_size_t, [C.POINTER(AttnDesc)]),
    "segclip_attn_bwd_ws_bytes": (C.c_size_t, [C.POINTER(AttnDesc)]),
    "segclip_attn_fwd": (C.c_int, [C.POINTER(AttnDesc), vp]),
    "segclip_attn_bwd": (C.c_int, [C.POINTER(AttnDesc), vp]),
    "segclip_resblock_fwd": (C.c_int, [C.POINTER(ResBlockFwdDesc), vp]),
    "segclip_cast": (C.c_int, [vp, vp, i64, C.c_int, C.c_int, vp]),
    "segclip_split3_bf16": (C.c_int, [vp, vp, i64, i64, i64, C.c_int, C.c_int, vp]),
    "segclip_colsum_ws_bytes": (C.c_size_t, [i64, i64]),
    "segclip_colsum": (C.c_int, [vp, vp, vp, i64, i64, i64, C.c_int, vp]),
    "segclip_act_fwd": (C.c_int, [vp, vp, i64, C.c_int, C.c_int, vp]),
    "segclip_act_bwd": (C.c_int, [vp, vp, vp, i64, C.c_int, C.c_int, vp]),
    "segclip_add": (C.c_int, [vp, vp, vp, i64, C.c_int, vp]),
    "segclip_scale": (C.c_int, [vp, vp, vp, i64, vp]),
    "segclip_gumbel_from_uniform": (C.c_int, [vp, vp, i64, vp]),
    "segclip_reduce_sum": (C.c_int, [vp, vp, i64, f32, vp]),
    "segclip_ce_labels_fwd": (C.c_int, [vp, vp, i64, vp, vp, vp, i64, i64, vp]),
    "segclip_ce_labels_bwd": (C.c_int, [vp, vp, vp, i64, vp, vp, vp, i64, i64, vp]),
    "segclip_im2col": (C.c_int, [vp, vp, i64, i64, i64, i64, i64, C.c_int, C.c_int, vp]),
    "segclip_im2col_ld": (C.c_int, [vp, vp, i64, i64, i64, i64, i64, C.c_int, C.c_int, i64, vp]),
    "segclip_vis_assemble": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, C.c_int, vp]),
    "segclip_embed_fwd": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, i64, vp]),
    "segclip_embed_bwd": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, i64, vp]),
    "segclip_recon_mix_fwd": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, vp]),
    "segclip_recon_mix_bwd": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, i64, i64, vp]),
    "segclip_mean_cat_fwd": (C.c_int, [vp, vp, i64, i64, i64, vp]),
    "segclip_mean_cat_bwd": (C.c_int, [vp, vp, i64, i64, i64, vp]),
    "segclip_mae_unshuffle_fwd": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, i64, i64, vp]),
    "segclip_mae_unshuffle_bwd": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, i64, i64, vp]),
    "segclip_gather_rows": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, C.c_int, vp]),
    "segclip_scatter_rows": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, C.c_int, vp]),
    "segclip_assign_fwd": (C.c_int, [vp, vp, f32, vp, vp, vp, vp, vp, i64, i64, i64, vp]),
    "segclip_assign_bwd": (C.c_int, [vp, vp, f32, vp, i64, i64, i64, vp]),
    "segclip_l2norm_fwd": (C.c_int, [vp, vp, vp, i64, i64, vp]),
    "segclip_l2norm_bwd": (C.c_int, [vp, vp, vp, vp, i64, i64, vp]),
    "segclip_ce_fwd": (C.c_int, [vp, vp, vp, i64, i64, i64, vp]),
    "segclip_ce_bwd": (C.c_int, [vp, vp, vp, f32, vp, i64, i64, i64, vp]),
    "segclip_superpixel_kl": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, vp]),
    "segclip_masked_mse_fwd": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, C.c_int, vp]),
    "segclip_masked_mse_bwd": (C.c_int, [vp, vp, vp, vp, vp, f32, vp, i64, i64, i64, C.c_int, vp]),
    "segclip_mask_sort": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, vp]),
    "segclip_interp_bicubic": (C.c_int, [vp, vp, i64, i64, i64, i64, vp]),
    "segclip_multi_cast_bf16": (C.c_int, [vp, vp, vp, i64, vp]),
    "segclip_multi_add_f32": (C.c_int, [vp, vp, vp, i64, vp]),
    "segclip_max_tokens_fwd": (C.c_int, [vp, vp, vp, i64, i64, i64, vp]),
    "segclip_max_tokens_bwd": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, vp]),
    "segclip_l2norm_pair_fwd": (C.c_int, [vp, vp, vp, vp, i64, i64, vp]),
    "segclip_l2norm_pair_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, i64, vp]),
    "segclip_clip_ce_fwd": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, i64, vp]),
    "segclip_clip_ce_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, vp]),
    "segclip_segmean_fwd": (C.c_int, [vp, vp, C.c_int, vp, vp, i64, i64, i64, i64, vp]),
    "segclip_segmean_bwd": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp, vp, i64, i64, i64, i64, vp]),
    "segclip_center_logits_fwd": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, vp]),
    "segclip_center_logits_bwd": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, i64, i64, vp]),
    "segclip_group_linear64": (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_int, vp, i64, C.c_int, vp]),
    "segclip_grad_sqnorm_ws_bytes": (C.c_size_t, [vp, i64]),
    "segclip_grad_sqnorm": (C.c_int, [vp, vp, i64, vp, vp, f32, vp]),
    "segclip_adamw_step": (C.c_int, [vp, i64, vp, i64, vp, vp, C.c_int, vp]),
    "segclip_train_step_finish": (C.c_int, [vp, vp, vp, f32, vp]),
}


def lib_path():
    return _LIB_PATH


def load():
    """Load the shared library (once).  Raises RuntimeError when it is absent: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"segclip_amd: {_LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or segclip_amd/csrc/build.sh (hipcc, gfx950). There is no CPU / PyTorch fallback.")
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the .so is stale w.r.t. the header
        fn.restype = res
        fn.argtypes = args
    if lib.segclip_version() != 1:
        raise RuntimeError("segclip_amd: ABI version mismatch")
    _lib = lib
    return lib


class Unsupported(RuntimeError):
    """SEGCLIP_ERR_UNSUPPORTED (-2): the library has no kernel for this combination; nothing was launched."""


def check(rc, what):
    if rc != 0:
        msg = load().segclip_last_error_string().decode(errors="replace")
        raise (Unsupported if rc == -2 else RuntimeError)(f"segclip_hip {what} failed (rc={rc}): {msg}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """hipStream_t of torch's current stream on the current device.  Called once per kernel launch (~800 per step): the raw
    accessor avoids torch.cuda.current_stream()'s Python wrappers (9 us per call = 2 ms of host time per forward)."""
    if _raw_stream is not None and _cur_device is not None:
        return C.c_void_p(_raw_stream(_cur_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"segclip_amd: unsupported dtype {t.dtype}")


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("segclip_amd: HIP kernels need device tensors (there is no CPU fallback); got a "
                           f"{t.device} tensor")
    return C.c_void_p(t.data_ptr())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("segclip_amd: HIP kernels need device tensors (there is no CPU fallback); got a "
                               f"{t.device} tensor")
