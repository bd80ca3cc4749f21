"""ctypes binding of libvlatouch_hip.so (include/vlatouch.h).

PyTorch is plumbing here: it owns device memory and streams; every computation of the
refinement path is a call into the HIP library.  There is NO CPU / eager fallback: if the
library is missing, or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# VLATOUCH_LIB: another build of the same library (tools/drain_waits_check.sh compares the product against a debug build); default = the in-tree product
LIB_PATH = os.environ.get("VLATOUCH_LIB") or os.path.join(_HERE, "libvlatouch_hip.so")

# bits of an engine's range-guard word (include/vlatouch.h, vt_rdt_set_range_flag / vt_dino_set_range_flag)
RANGE_XN_SAT, RANGE_NONFINITE, RANGE_GATE_SAT, RANGE_ATTN_EMPTY = 1, 2, 4, 8
RANGE_NAMES = {RANGE_XN_SAT: "xn_saturated", RANGE_NONFINITE: "nonfinite", RANGE_GATE_SAT: "gate_saturated", RANGE_ATTN_EMPTY: "attention_row_empty"}


def csrc_sha16() -> str:
    """Fingerprint of the kernel sources this tree's library is built from (csrc/*.hip, *.h, Makefile in name order): stamped into profiles/pmc_traffic.json by
    tools/pmc_summary.py and compared by bench.py, so that a PMC figure taken on another build of the kernels is labelled as such."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(_HERE), "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def range_names(bits: int):
    return [n for b, n in RANGE_NAMES.items() if bits & b]


F32, BF16, F32X3, F16 = 0, 1, 2, 3   # F32X3: fp32 storage, split-bf16 3-MFMA compute (GEMM weights only); F16: IEEE half
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_SILU, ACT_MISH, ACT_SWIGLU = 0, 1, 2, 3, 4, 5
NORM_LAYER, NORM_RMS_MEANSQ, NORM_RMS_VAR = 0, 1, 2
IMGNORM_AUTO, IMGNORM_ON, IMGNORM_OFF = 0, 1, 2


class VtError(RuntimeError):
    pass


class GemmParams(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_long), ("ldw", C.c_long), ("ldc", C.c_long),
        ("taps", C.c_int), ("cin", C.c_int), ("tout", C.c_int), ("tin", C.c_int), ("stride", C.c_int),
        ("off0", C.c_int), ("tstep", C.c_int),
        ("bias", C.c_void_p), ("colscale", C.c_void_p), ("residual", C.c_void_p), ("ldr", C.c_long),
        ("act", C.c_int),
        ("groups", C.c_int), ("splitk", C.c_int),
        ("a_gs", C.c_long), ("w_gs", C.c_long), ("c_gs", C.c_long), ("bias_gs", C.c_long), ("r_gs", C.c_long),
        ("c_slab", C.c_long),
        ("a_dtype", C.c_int), ("w_dtype", C.c_int), ("c_dtype", C.c_int),
        ("hn_w0", C.c_void_p), ("hn_w1", C.c_void_p), ("hn_c0_end", C.c_int), ("hn_c1_end", C.c_int),
        ("hn_eps", C.c_float), ("hn_mode", C.c_int),
        ("cmap", C.c_int), ("cmap_T", C.c_int),
        ("Wp", C.c_void_p),
        ("sk_ws", C.c_void_p), ("sk_ws_bytes", C.c_size_t), ("sk_cnt", C.c_void_p), ("sk_cnt_n", C.c_int),
        # fused RMSNorm hand-off between two Linears (vt_gemm.h): producer side / consumer side
        ("xn_out", C.c_void_p), ("xn_ld", C.c_long), ("xn_gain", C.c_void_p), ("xn_part", C.c_void_p),
        ("rs_part", C.c_void_p), ("rs_n", C.c_int), ("rs_inv_k", C.c_float), ("rs_eps", C.c_float), ("rs_mode", C.c_int),
        ("pf_ptr", C.c_void_p), ("pf_bytes", C.c_size_t),
        ("range_flag", C.c_void_p),      # range-guard word (include/vlatouch.h, vt_rdt_set_range_flag); null = none
    ]


class GnParams(C.Structure):
    _fields_ = [
        ("P", C.c_void_p), ("nslabs", C.c_int), ("slab_stride", C.c_long), ("p_gs", C.c_long), ("ldp", C.c_long),
        ("bias", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("vec_gs", C.c_long),
        ("film", C.c_void_p), ("film_ld", C.c_long), ("film_off", C.c_long), ("film_gs", C.c_long),
        ("residual", C.c_void_p), ("ldr", C.c_long), ("r_gs", C.c_long),
        ("out", C.c_void_p), ("ldo", C.c_long), ("o_gs", C.c_long), ("out_dtype", C.c_int),
        ("B", C.c_int), ("T", C.c_int), ("C", C.c_int), ("ngroups", C.c_int), ("nets", C.c_int),
        ("eps", C.c_float),
    ]


class AttnParams(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p),
        ("q_bs", C.c_long), ("q_rs", C.c_long), ("q_hs", C.c_long),
        ("k_bs", C.c_long), ("k_rs", C.c_long), ("k_hs", C.c_long),
        ("v_bs", C.c_long), ("v_rs", C.c_long), ("v_hs", C.c_long),
        ("o_bs", C.c_long), ("o_rs", C.c_long),
        ("kmask", C.c_void_p), ("km_bs", C.c_long),
        ("B", C.c_int), ("H", C.c_int), ("Nq", C.c_int), ("Nk", C.c_int),
        ("scale", C.c_float), ("dtype", C.c_int), ("hd", C.c_int),
    ]


class UnetDesc(C.Structure):
    _fields_ = [("nets", C.c_int), ("input_dim", C.c_int), ("input_pad", C.c_int), ("cond_dim", C.c_int),
                ("dsed", C.c_int), ("n_groups", C.c_int), ("ksize", C.c_int), ("n_levels", C.c_int),
                ("dims", C.c_int * 4), ("cdt", C.c_int), ("adt", C.c_int)]


class DinoDesc(C.Structure):
    _fields_ = [("hidden", C.c_int), ("layers", C.c_int), ("heads", C.c_int), ("patch", C.c_int), ("kpad", C.c_int),
                ("cdt", C.c_int), ("adt", C.c_int), ("eps", C.c_float),
                ("no_cls", C.c_int), ("act", C.c_int), ("head_dim", C.c_int), ("mlp_dim", C.c_int), ("out_all", C.c_int),
                ("attn_scale", C.c_float)]


class LstmDesc(C.Structure):
    _fields_ = [("state_dim", C.c_int), ("hidden", C.c_int), ("layers", C.c_int), ("force_dim", C.c_int),
                ("force_pad", C.c_int), ("in_pad", C.c_int), ("cdt", C.c_int)]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load the HIP library (fails loudly: the product has no other compute path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VtError(f"{LIB_PATH} not found — build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                      "vlatouch has no CPU/eager fallback.")
    L = C.CDLL(LIB_PATH)
    L.vt_last_error.restype = C.c_char_p
    L.vt_version.restype = C.c_int
    for name, (res, args) in SIGNATURES.items():
        if not hasattr(L, name):
            raise VtError(f"{LIB_PATH} does not export {name} (stale build?)")
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


_P, _I, _L, _F, _Z = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t
# every entry point of include/vlatouch.h: name -> (restype, argtypes)
SIGNATURES = {
    "vt_selftest_mfma": (_I, [_P, _P]),
    "vt_prof_enable": (_I, [_I]),
    "vt_prof_collect": (_I, [_P, _P, _P, _P]),
    "vt_gemm": (_I, [_P, _P]),
    "vt_randn": (_I, [_P, _L, _P, _I, _P]),
    "vt_slice_cast": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "vt_cast": (_I, [_P, _I, _L, _P, _I, _L, _I, _I, _P]),
    "vt_pack_w32": (_I, [_P, _L, _P, _I, _I, _P]),
    "vt_tune": (_I, [_I, _I]),
    "vt_attention": (_I, [_P, _P]),
    "vt_groupnorm": (_I, [_P, _P]),
    "vt_rownorm": (_I, [_P, _I, _L, _P, _I, _L, _P, _P, _I, _I, _F, _I, _P]),
    "vt_headnorm": (_I, [_P, _I, _L, _I, _L, _P, _F, _I, _P]),
    "vt_action_normalize": (_I, [_P, _P, _P, _P, _L, _I, _F, _I, _P]),
    "vt_unet_create": (_I, [_P, _P, _I, _P]),
    "vt_unet_destroy": (None, [_P]),
    "vt_unet_num_weights": (_I, [_P]),
    "vt_unet_workspace_bytes": (_Z, [_P, _I, _I]),
    "vt_unet_fused_bytes": (_Z, [_P]),
    "vt_unet_fused_pack": (_I, [_P, _P, _P]),
    "vt_unet_fused_covers": (_I, [_P, _I, _I, _I]),
    "vt_unet_fused_plan_bytes": (_Z, [_P, _I, _I, _I]),
    "vt_unet_forward": (_I, [_P, _P, _P, _F, _P, _P, _I, _I, _P, _P]),
    "vt_si_sample": (_I, [_P, _P, _P, _P, _I, _F, _I, _I, _I, _P, _I, _I, _P, _P]),
    "vt_si_sample_ex": (_I, [_P, _P, _P, _P, _I, _F, _I, _I, _I, _I, _F, _P, _I, _I, _P, _P]),
    "vt_dino_create": (_I, [_P, _P, _I, _P]),
    "vt_dino_destroy": (None, [_P]),
    "vt_dino_num_weights": (_I, [_P]),
    "vt_dino_workspace_bytes": (_Z, [_P, _I, _I]),
    "vt_dino_packed_bytes": (_Z, [_P]),
    "vt_dino_set_packed": (_I, [_P, _P, _P]),
    "vt_dino_set_range_flag": (_I, [_P, _P]),
    "vt_dino_forward": (_I, [_P, _P, _I, _I, _I, _F, _I, _I, _I, _P, _P, _P, _P, _P]),
    "vt_mlp": (_I, [_P, _L, _I, _I, _P, _P, _P, _I, _P, _I, _L, _I, _I, _P, _P]),
    "vt_concat_obs": (_I, [_P, _P, _I, _P, _I, _P, _I, _P, _I, _L, _I, _P]),
    "vt_lstm_create": (_I, [_P, _P, _I, _P]),
    "vt_lstm_destroy": (None, [_P]),
    "vt_lstm_num_weights": (_I, [_P]),
    "vt_lstm_workspace_bytes": (_Z, [_P, _I]),
    "vt_lstm_step": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "vt_lstm_sequence": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "vt_rdt_create": (_I, [_P, _P, _I, _P]),
    "vt_rdt_destroy": (None, [_P]),
    "vt_rdt_num_weights": (_I, [_P]),
    "vt_rdt_workspace_bytes": (_Z, [_P, _I, _I]),
    "vt_rdt_set_score_bounds": (_I, [_P, _P, _I]),
    "vt_rdt_set_state_precision": (_I, [_P, _I]),
    "vt_rdt_set_io_dtype": (_I, [_P, _I]),
    "vt_rdt_packed_bytes": (_Z, [_P]),
    "vt_rdt_set_packed": (_I, [_P, _P, _P]),
    "vt_rdt_set_range_flag": (_I, [_P, _P]),
    "vt_rdt_forward": (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P, _P, _I, _I, _P, _P]),
    "vt_rdt_sample": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P, _I, _I, _P, _P]),
    "vt_im2col_t": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "vt_transpose": (_I, [_P, _P, _I, _I, _P]),
    "vt_zero_stuff": (_I, [_P, _P, _I, _I, _I, _P]),
    "vt_wflip": (_I, [_P, _P, _I, _I, _I, _P]),
    "vt_colsum": (_I, [_P, _L, _P, _I, _I, _I, _P]),
    "vt_add_": (_I, [_P, _P, _L, _P]),
    "vt_copy_cols": (_I, [_P, _L, _I, _P, _L, _I, _I, _I, _I, _P]),
    "vt_mish": (_I, [_P, _P, _P, _L, _P]),
    "vt_gn_mish_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "vt_gelu": (_I, [_P, _P, _P, _L, _P]),
    "vt_si_qsample": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _F, _P]),
    "vt_si_qsample_ex": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _F, _I, _P]),
    "vt_si_loss": (_I, [_P, _P, _P, _P, _I, _L, _P]),
    "vt_slab_sum": (_I, [_P, _I, _L, _P, _I, _P, _P]),
    "vt_adamw": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P]),
    "vt_ema_update": (_I, [_P, _P, _L, _F, _P]),
    "vt_train_hyper": (_I, [_F, _F, _F, _I, _F, _P]),
    "vt_adamw_dev": (_I, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _P]),
    "vt_adamw_ema_multi": (_I, [_P, _I, _L, _P, _F, _F, _F, _F, _P]),
    "vt_ema_update_dev": (_I, [_P, _P, _L, _P, _P]),
    "vt_posemb": (_I, [_P, _P, _I, _I, _P]),
    "vt_lstm_cell_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vt_lstm_cell_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vt_ln_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "vt_bcast_mid": (_I, [_P, _P, _L, _I, _I, _I, _I, _P]),
    "vt_sum_mid": (_I, [_P, _L, _I, _P, _I, _I, _I, _P]),
    "vt_mul_": (_I, [_P, _P, _L, _P]),
    "vt_mse_residual": (_I, [_P, _P, _P, _P, _P, _P, _L, _P]),
    "vt_marker_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "vt_marker_detect": (_I, [_P, _I, _I, _I, _I, _I, C.c_double, C.c_double, _I, _P, _P, _I, _P, _P, _P]),
    "vt_marker_displacement": (_I, [_P, _P, _I, _I, _P, _I, _P, _P, _P]),
}


class RdtDesc(C.Structure):
    _fields_ = [("hidden", C.c_int), ("depth", C.c_int), ("heads", C.c_int), ("horizon", C.c_int), ("out_dim", C.c_int),
                ("state_dim", C.c_int), ("lang_dim", C.c_int), ("img_dim", C.c_int), ("max_lang_len", C.c_int), ("img_len", C.c_int),
                ("n_lang", C.c_int), ("n_img", C.c_int), ("n_state", C.c_int), ("cdt", C.c_int), ("adt", C.c_int), ("rms_mode", C.c_int)]


def check(code: int, what: str = "") -> None:
    if code != 0:
        raise VtError(f"{what or 'vlatouch'} failed with code {code}: {lib().vt_last_error().decode()}")


def stream_ptr(device=None) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def dt_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float16:
        return F16
    raise VtError(f"unsupported dtype {dtype}")


def torch_dtype(code: int) -> torch.dtype:
    return {F32: torch.float32, F32X3: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[code]


def ptr_array(tensors: Sequence[Optional[torch.Tensor]]):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = 0 if t is None else t.data_ptr()
    return arr


def require_gpu(device) -> torch.device:
    device = torch.device(device)
    if device.type != "cuda" or not torch.cuda.is_available():
        raise VtError("vlatouch runs only on an AMD GPU through libvlatouch_hip.so (device must be 'cuda'; "
                      "no CPU fallback exists)")
    lib()
    return device
