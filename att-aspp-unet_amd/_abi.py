"""ctypes binding of ``lib/libaau.so`` (C ABI declared in ``include/aau.h``).

This is the only place the host side touches native code.  There is NO fallback:
if the library is missing or a call fails, an exception is raised (``AauError``).
"""
from __future__ import annotations

import ctypes as C
import os
import re

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB_PATH = os.environ.get("AAU_LIB") or os.path.join(PKG, "lib", "libaau.so")  # AAU_LIB: kernel-ablation builds
HEADER = os.path.join(ROOT, "include", "aau.h")

STAT_REPLICAS = 32
PROF_FAMILIES = 4
PROF_NAMES = ("igemm", "wgrad", "elementwise", "loss_optim")


class AauError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """``aau_conv_desc`` (include/aau.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "N", "H", "W", "Cin", "src_pitch", "Ho", "Wo", "Cout", "dst_pitch", "KH", "KW",
        "stride", "pad", "dil", "Cpad", "shuffle2x2", "accumulate", "relu",
        "src_split_c", "src_split_off", "dst_split_c", "dst_split_off")]


class PackEntry(C.Structure):
    """``aau_pack_entry`` (include/aau.h)."""
    _fields_ = [("src_off", C.c_int64), ("dst_off", C.c_int64), ("R", C.c_int32), ("T", C.c_int32),
                ("C", C.c_int32), ("Cpad", C.c_int32), ("s_r", C.c_int32), ("s_t", C.c_int32),
                ("s_c", C.c_int32), ("t_flip", C.c_int32), ("R2", C.c_int32), ("s_r2", C.c_int32),
                ("blk_begin", C.c_int64)]


P = C.c_void_p
I = C.c_int
L = C.c_int64
F = C.c_float
U64 = C.c_uint64

# entry points that take an aau_stat buffer: index of the pointer argument; its size in bytes is the NEXT argument
# (include/aau.h: the library itself refuses an undersized buffer)
STAT_ARG = {"aau_conv_igemm_bnin": 7, "aau_conv_igemm_bnred": 10, "aau_stats_to_red": 0, "aau_conv_igemm": 7, "aau_conv1_fwd": 3, "aau_bn_finalize": 0, "aau_gate_psi": 8, "aau_fold_stats": 0,
            "aau_stats_to_f64": 0}

# name -> argtypes (all return int unless noted)
_SIGS = {
    "aau_prof_enable": [I],
    "aau_prof_collect": [P, P, P],
    "aau_prof_collect_launches": [I, P, P, P, P, P, P],
    "aau_prof_label": [C.c_char_p],
    "aau_conv_igemm": [C.POINTER(ConvDesc), P, P, P, P, P, P, P, L, P],
    "aau_conv_is_halo3x3": [C.POINTER(ConvDesc)],
    "aau_conv_bnred_ok": [C.POINTER(ConvDesc)],
    "aau_conv_igemm_bnred": [C.POINTER(ConvDesc), P, P, P, P, I, P, P, P, P, P, L, P],
    "aau_stats_to_red": [P, L, I, P, P],
    "aau_bn_finalize_multi": [I, P, L, I, L, F, F, P],
    "aau_bn_act_multi": [I, P, I, I, L, I, I, P],
    "aau_bn_bwd_reduce_multi": [I, P, I, I, I, I, I, I, I, P],
    "aau_bn_bwd_apply_multi": [I, P, I, I, I, L, I, I, P],
    "aau_conv_bnin_ok": [C.POINTER(ConvDesc)],
    "aau_conv_wgrad_bnin_ok": [C.POINTER(ConvDesc)],
    "aau_conv_wgrad_bnin": [C.POINTER(ConvDesc), P, P, P, P, P, P, L, P],
    "aau_conv_igemm_bnin": [C.POINTER(ConvDesc), P, P, P, P, P, P, P, L, P],
    "aau_conv_wgrad_bnin_dz_ok": [C.POINTER(ConvDesc)],
    "aau_conv_wgrad_bnin_dz": [C.POINTER(ConvDesc), P, P, P, P, P, P, L, P],
    "aau_traverse": [I],
    "aau_conv_wgrad": [C.POINTER(ConvDesc), P, P, P, P, C.c_int64, P],
    "aau_conv_split_ok": [C.POINTER(ConvDesc), I],
    "aau_conv_wgrad_ws_bytes": [C.POINTER(ConvDesc), C.POINTER(C.c_int64)],
    "aau_conv_wgrad_group_ok": [C.POINTER(ConvDesc), I],
    "aau_conv_wgrad_group_member_ok": [C.POINTER(ConvDesc)],
    "aau_conv_wgrad_group": [C.POINTER(ConvDesc), P, P, P, I, P, L, P],
    "aau_conv_wgrad_group_queue_bytes": [],
    "aau_conv_igemm_multi_ok": [C.POINTER(ConvDesc), I],
    "aau_conv_igemm_multi": [C.POINTER(ConvDesc), P, P, P, P, P, I, P],
    "aau_conv_igemm_group_ok": [C.POINTER(ConvDesc), I],
    "aau_conv_igemm_group_ws_bytes": [C.POINTER(ConvDesc), I],
    "aau_conv_igemm_group": [C.POINTER(ConvDesc), P, P, I, P, P, P],
    "aau_conv1_fwd": [P, P, P, P, L, I, I, I, I, P],
    "aau_conv1_wgrad": [P, P, P, I, I, I, I, P],
    "aau_pack_weights": [P, P, P, I, L, P],
    "aau_zero_multi": [P, P, I, P, C.c_uint64, P],
    "aau_bn_finalize": [P, L, P, P, P, P, P, P, P, P, P, I, L, F, F, P],
    "aau_bn_fold_eval": [P, P, P, P, P, P, I, F, P],
    "aau_bn_act": [P, I, P, I, P, P, L, I, I, L, F, P, P],
    "aau_bn_act_pool": [P, I, P, I, P, I, P, P, I, I, I, I, P],
    "aau_maxpool2": [P, I, P, I, I, I, I, I, P],
    "aau_bn_bwd_reduce": [P, I, P, I, P, I, P, I, P, P, P, P, P, I, I, I, I, I, F, P, P, P],
    "aau_bn_bwd_apply": [P, I, P, I, P, P, P, P, P, P, L, I, P, I, P, P, I, F, P, P],
    "aau_bn_bwd_apply_pool": [P, I, P, I, P, P, P, P, P, P, I, I, I, I, P, I, P, I, P, P, I, P],
    "aau_bn_bwd_apply_conv1": [P, I, P, P, P, P, P, P, I, I, I, I, P, I, P, P, P, P, P, P, P],
    "aau_conv1_bn_act": [P, P, P, I, P, P, I, I, I, I, P],
    "aau_conv1_bn_bwd_reduce": [P, P, P, I, P, P, P, P, P, I, I, I, I, P, P],
    "aau_gap_fwd": [P, I, P, P, I, I, I, P],
    "aau_gap_bwd": [P, P, I, I, I, I, P],
    "aau_poolbranch_fwd": [P, I, P, I, P, P, P, P, P, P, P, P, P, P, I, I, I, F, F, P],
    "aau_poolbranch_bwd": [P, I, P, P, I, P, P, P, P, P, P, P, P, P, I, I, I, P],
    "aau_poolbranch_dx": [P, P, I, P, I, I, I, I, P],
    "aau_spatial_sum": [P, I, P, P, I, I, I, P],
    "aau_gate_psi": [P, P, P, P, P, P, P, P, P, L, L, I, P],
    "aau_gate_apply": [P, I, P, P, P, P, P, I, L, I, P],
    "aau_gate_bwd1": [P, I, P, I, P, P, P, P, P, I, P, P, L, I, P, P],
    "aau_gate_bwd2": [P] * 21 + [L, I, P, P],
    "aau_gate_bwd3": [P] * 17 + [L, I, P],
    "aau_gate2_fwd": [P, P, P, P, P, I, P, P, I, L, I, I, P],
    "aau_gate2_bwd": [P, I, P, I, P, P, P, P, P, I, P, P, P, P, L, I, I, P],
    "aau_fold_stats": [P, L, I, I, I, I, P, P],
    "aau_fold_stats_pair": [P, L, I, I, P, L, I, I, I, P, P],
    "aau_stats_to_f64": [P, L, I, P, P],
    "aau_outconv_fwd": [P, I, P, P, P, L, I, P],
    "aau_outconv_bwd": [P, I, P, P, P, I, P, P, P, L, I, P],
    "aau_colsum": [P, I, P, P, L, I, P],
    "aau_fold_replicas": [P, I, P, I, P],
    "aau_bn_act_outconv": [P, I, P, P, P, P, P, L, I, P],
    "aau_bn_bwd_reduce_outconv": [P, I, P, P, P, P, P, P, P, P, P, P, L, I, P],
    "aau_bn_bwd_apply_rank1": [P, I, P, I, P, P, P, P, P, P, L, I, P, P, P, P, P],
    "aau_criterion": [P, P, P, P, P, I, I, I, I, F, F, F, P],
    "aau_seg_metrics": [P, P, P, P, I, I, I, F, P],
    "aau_seg_counts": [P, I, P, I, L, P, P],
    "aau_resize_bilinear_f32": [P, I, I, P, I, I, I, P],
    "aau_resize_bilinear_u8": [P, I, I, P, I, I, I, P],
    "aau_gauss5_f32": [P, P, I, I, I, P],
    "aau_threshold_u8": [P, F, P, L, P],
    "aau_cc_label": [P, P, I, I, I, I, P],
    "aau_cc_keep_largest": [P, P, P, P, P, I, I, I, I, I, P],
    "aau_fill_holes": [P, P, P, P, I, I, I, P],
    "aau_morph": [P, P, I, I, I, I, I, P],
    "aau_normalize_minmax_u8": [P, P, P, I, I, I, P],
    "aau_clahe_u8": [P, P, P, I, I, I, F, I, P],
    "aau_hflip_frames_u8": [P, P, P, I, I, I, P],
    "aau_warp_affine_u8": [P, P, P, I, I, I, I, I, P],
    "aau_lut_u8": [P, P, P, I, L, P],
    "aau_elastic_noise": [P, P, I, I, I, P],
    "aau_gauss_sep_f32": [P, P, P, P, I, L, I, I, P],
    "aau_remap_u8": [P, P, P, P, I, I, I, I, P],
    "aau_select_frames_u8": [P, P, P, P, I, L, P],
    "aau_median3_u8": [P, P, I, I, I, P],
    "aau_u8_to_f32": [P, P, F, L, P],
    "aau_roi_origin": [P, P, P, I, I, I, I, P],
    "aau_roi_crop": [P, P, P, I, I, I, I, P],
    "aau_roi_paste_sigmoid": [P, P, P, I, I, I, I, P],
    "aau_frame_areas": [P, F, P, I, I, I, P],
    "aau_loss_terms": [P, P, P, P, P, I, I, I, C.POINTER(C.c_float), P],
    "aau_grad_sqnorm": [P, L, F, P, P],
    "aau_adamw_step": [P, P, P, P, L, P, P, F, F, F, F, F, F, F, P],
    "aau_adamw_step_dev": [P, P, P, P, L, P, P, P, P, I, F, F, F, F, F, P],
    "aau_f32_to_bf16": [P, P, L, P],
    "aau_bf16_to_f32": [P, P, L, P],
    "aau_nchw_to_nhwc": [P, P, I, I, I, I, I, P],
    "aau_nhwc_to_nchw": [P, I, P, I, I, I, I, P],
    "aau_hflip_f32": [P, P, I, I, I, P],
    "aau_tta_merge": [P, P, P, I, I, I, P],
    "aau_window_blend": [P, P, I, I, I, I, I, I, F, P],
}

_libs: dict = {}
_kind = "bf16"          # which build of the library calls resolve to: "bf16" (libaau.so) or "fp16" (libaau_f16.so)
LIB_PATHS = {"bf16": LIB_PATH, "fp16": LIB_PATH.replace("libaau.so", "libaau_f16.so")}


def declared_symbols() -> list[str]:
    """Every function name declared in include/aau.h."""
    with open(HEADER) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(aau_[a-z0-9_]+)\s*\(", text)))


def lib(kind: str | None = None) -> C.CDLL:
    """Load the library of the current 16-bit storage type (once).  Fails loudly when it has not been built."""
    kind = kind or _kind
    l = _libs.get(kind)
    if l is None:
        path = LIB_PATHS[kind]
        if not os.path.exists(path):
            raise AauError(f"{path} not found: run `python __graft_entry__.py` / build() first "
                           "(there is no non-HIP fallback)")
        l = C.CDLL(path)
        l.aau_last_error.restype = C.c_char_p
        l.aau_last_error.argtypes = []
        l.aau_version.restype = C.c_int
        l.aau_version.argtypes = []
        for name, sig in _SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = sig
            fn.restype = C.c_int
        l.aau_bn_red_ws_bytes.restype = C.c_int64
        l.aau_conv_igemm_group_ws_bytes.restype = C.c_int64
        l.aau_conv_wgrad_group_queue_bytes.restype = C.c_int64
        l.aau_bn_red_ws_bytes.argtypes = [C.c_int]
        _libs[kind] = l
    return l


class precision:
    """``with precision("fp16"):`` -- calls made (and launch lists recorded) inside resolve to the IEEE-half build of the
    library (inference only: the reference's fp16 configuration); the default is the bfloat16 build."""

    def __init__(self, kind: str):
        if kind not in LIB_PATHS:
            raise AauError(f"unknown precision {kind!r} (bf16 | fp16)")
        self.kind = kind

    def __enter__(self):
        global _kind
        self.prev, _kind = _kind, self.kind
        return self

    def __exit__(self, *exc):
        global _kind
        _kind = self.prev
        return False


def current_precision() -> str:
    return _kind


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().aau_last_error().decode(errors="replace")
        raise AauError(f"{what or 'aau call'} failed (rc={rc}): {msg}")


def fn(name: str):
    return getattr(lib(), name)


PROF_ON = False   # mirrors aau_prof_enable: the engine then names every launch (aau_prof_label)


def prof_enable(on: bool) -> None:
    global PROF_ON
    check(lib().aau_prof_enable(1 if on else 0), "aau_prof_enable")
    PROF_ON = bool(on)


def prof_collect_launches(cap: int = 8192) -> list:
    """-> [{tag, ms, flops, bytes, family}] in issue order (clears the records)."""
    n = C.c_int(0)
    tags = C.create_string_buffer(cap * 96)
    ms, fl, by = (C.c_double * cap)(), (C.c_double * cap)(), (C.c_double * cap)()
    fam = (C.c_int * cap)()
    check(lib().aau_prof_collect_launches(cap, C.byref(n), tags, ms, fl, by, fam), "aau_prof_collect_launches")
    out = []
    for i in range(n.value):
        label, _, t = tags.raw[i * 96:(i + 1) * 96].split(b"\0", 1)[0].decode().partition("|")
        out.append({"label": label, "tag": t, "ms": ms[i], "flops": fl[i], "bytes": by[i], "family": PROF_NAMES[fam[i]]})
    return out


def prof_collect() -> dict:
    ms = (C.c_double * PROF_FAMILIES)()
    n = (C.c_int64 * PROF_FAMILIES)()
    fl = (C.c_double * PROF_FAMILIES)()
    check(lib().aau_prof_collect(ms, n, fl), "aau_prof_collect")
    return {PROF_NAMES[i]: {"ms": ms[i], "launches": n[i], "flops": fl[i]} for i in range(PROF_FAMILIES)}
