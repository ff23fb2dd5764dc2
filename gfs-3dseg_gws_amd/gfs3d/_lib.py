"""ctypes binding of libgfs3d_hip.so (C ABI: include/gfs3d.h).

The library is built in-tree by ``build_library()`` (hipcc, gfx950 only) and
loaded lazily.  There is NO fallback: if the shared object is missing or a call
fails, a RuntimeError is raised - the hot path is HIP or nothing.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_ulonglong, c_void_p

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CSRC = os.path.join(_PKG, "csrc")
_SO = os.environ.get("GFS3D_LIB") or os.path.join(_PKG, "lib", "libgfs3d_hip.so")   # GFS3D_LIB: ablation builds only
_SOURCES = ("api.hip", "knn.hip", "edgeconv.hip", "proto.hip", "pointwise.hip", "attention.hip", "head.hip", "kmeanspp.hip", "optim.hip")
_lib = None
KNN_IMG_SLACK_ROWS = 128      # GFS3D_KNN_IMG_SLACK_ROWS


class EdgeConvArgs(Structure):
    _fields_ = [
        ("B", c_int), ("C", c_int), ("N", c_int), ("k", c_int),
        ("training", c_int),
        ("eps", c_float), ("momentum", c_float), ("slope", c_float),
        ("x", c_void_p), ("idx", c_void_p), ("w1", c_void_p),
        ("bn1_w", c_void_p), ("bn1_b", c_void_p), ("bn1_rm", c_void_p), ("bn1_rv", c_void_p),
        ("w2", c_void_p),
        ("bn2_w", c_void_p), ("bn2_b", c_void_p), ("bn2_rm", c_void_p), ("bn2_rv", c_void_p),
        ("y_cf", c_void_p), ("y_cf_bstride", c_longlong), ("y_pm", c_void_p),
        ("P", c_void_p), ("Q", c_void_p), ("SP", c_void_p), ("SQ", c_void_p), ("indeg", c_void_p),
        ("stats", c_void_p), ("hsel", c_void_p), ("jsel", c_void_p),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("x_pm", c_void_p),
        ("Pp", c_void_p), ("Qp", c_void_p),
        ("y_sqnorm", c_void_p),
        ("compute", c_int),
        ("csr_rowptr", c_void_p), ("csr_pos", c_void_p),
        ("y_knn_img", c_void_p),
    ]


class EdgeConvBwdArgs(Structure):
    _fields_ = [
        ("f", EdgeConvArgs),
        ("dy_cf", c_void_p), ("dy_bstride", c_longlong),
        ("dx", c_void_p),
        ("dw1", c_void_p), ("dbn1_w", c_void_p), ("dbn1_b", c_void_p),
        ("dw2", c_void_p), ("dbn2_w", c_void_p), ("dbn2_b", c_void_p),
        ("dy_pm", c_void_p),
    ]


PW_MAX_LAYERS, PW_MAX_SEGS, PW_SEG_WIDTH = 4, 4, 64
COMPUTE_F32, COMPUTE_BF16, COMPUTE_F16 = 0, 1, 2          # include/gfs3d.h: GFS3D_COMPUTE_*


class PointwiseArgs(Structure):
    _fields_ = [
        ("B", c_int), ("N", c_int), ("n_seg", c_int), ("n_layers", c_int),
        ("width", c_int * PW_MAX_LAYERS),
        ("training", c_int),
        ("eps", c_float), ("momentum", c_float), ("slope", c_float),
        ("x_seg", c_void_p * PW_MAX_SEGS),
        ("w", c_void_p * PW_MAX_LAYERS),
        ("bn_w", c_void_p * PW_MAX_LAYERS), ("bn_b", c_void_p * PW_MAX_LAYERS),
        ("bn_rm", c_void_p * PW_MAX_LAYERS), ("bn_rv", c_void_p * PW_MAX_LAYERS),
        ("h", c_void_p * PW_MAX_LAYERS), ("stats", c_void_p * PW_MAX_LAYERS),
        ("y_cf", c_void_p),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("compute", c_int),
        ("seg_width", c_int),
        ("bias", c_void_p * PW_MAX_LAYERS),
        ("per_layer_slope", c_int),
        ("slope_l", c_float * PW_MAX_LAYERS),
        ("y_pm", c_void_p),
        ("store16", c_int),
        ("x16", c_void_p),
        ("x_type", c_int),
    ]


class PointwiseBwdArgs(Structure):
    _fields_ = [
        ("f", PointwiseArgs),
        ("dy_cf", c_void_p),
        ("dx_seg", c_void_p * PW_MAX_SEGS),
        ("dw", c_void_p * PW_MAX_LAYERS), ("dbn_w", c_void_p * PW_MAX_LAYERS), ("dbn_b", c_void_p * PW_MAX_LAYERS),
        ("dy_pm", c_void_p),
        ("dx_type", c_int),
    ]


PACK_MAX_SRC = 8


class PackArgs(Structure):
    _fields_ = [
        ("B", c_int), ("N", c_int), ("n_src", c_int),
        ("src", c_void_p * PACK_MAX_SRC),
        ("width", c_int * PACK_MAX_SRC),
        ("channel_first", c_int * PACK_MAX_SRC),
        ("dst", c_void_p),
        ("dst_ld", c_int),
        ("dst_type", c_int),
    ]


ADAM_MAX_TENSORS = 64


class AdamArgs(Structure):
    _fields_ = [
        ("n_tensors", c_int),
        ("param", c_void_p * ADAM_MAX_TENSORS), ("grad", c_void_p * ADAM_MAX_TENSORS),
        ("exp_avg", c_void_p * ADAM_MAX_TENSORS), ("exp_avg_sq", c_void_p * ADAM_MAX_TENSORS),
        ("numel", c_longlong * ADAM_MAX_TENSORS),
        ("lr", c_float * ADAM_MAX_TENSORS), ("weight_decay", c_float * ADAM_MAX_TENSORS),
        ("first_block", c_int * ADAM_MAX_TENSORS),
        ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("bias_correction1", c_float), ("bias_correction2", c_float),
        ("one_minus_beta1", c_float), ("one_minus_beta2", c_float),
    ]


HEAD_MAX_FAKE = 16


class CaplHeadArgs(Structure):
    _fields_ = [
        ("B", c_int), ("C", c_int), ("N", c_int),
        ("feat", c_void_p), ("feat_bstride", c_longlong),
        ("label", c_void_p), ("ignore_index", c_longlong),
        ("main_proto", c_void_p), ("classes", c_int),
        ("bg_proto", c_void_p), ("base_num", c_int),
        ("n_fake", c_int), ("fake", c_int * HEAD_MAX_FAKE), ("half", c_int),
        ("fsum", c_void_p),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("loss", c_void_p), ("pred", c_void_p),
        ("dloss", c_void_p), ("dfsum", c_void_p), ("dfeat", c_void_p), ("dmain", c_void_p), ("dbg", c_void_p),
        ("stream", c_void_p),
    ]


class CaplEvalArgs(Structure):
    _fields_ = [
        ("B", c_int), ("C", c_int), ("N", c_int),
        ("feat", c_void_p), ("feat_bstride", c_longlong),
        ("main_proto", c_void_p), ("classes", c_int), ("base_num", c_int),
        ("gened_proto", c_void_p), ("word", c_void_p), ("coding", c_void_p), ("Kw", c_int), ("th", c_float),
        ("label", c_void_p), ("logits", c_void_p), ("stats", c_void_p),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("stream", c_void_p),
    ]


# every symbol include/gfs3d.h declares: (restype, argtypes)
SYMBOLS = {
    "gfs3d_version": (c_int, []),
    "gfs3d_last_error": (c_char_p, []),
    "gfs3d_profile_enable": (c_int, [c_int]),
    "gfs3d_profile_read": (c_int, [c_int, POINTER(c_double), POINTER(c_longlong)]),
    "gfs3d_knn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "gfs3d_knn_topk_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gfs3d_knn_topk_norms_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gfs3d_knn_topk_pm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                      c_void_p]),
    "gfs3d_knn_topk_img_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                       c_size_t, c_void_p]),
    "gfs3d_edgeconv_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "gfs3d_set_deterministic": (c_int, [c_int]),
    "gfs3d_set_bn_tail": (c_int, [c_int]),
    "gfs3d_edgeconv_forward_f32": (c_int, [POINTER(EdgeConvArgs), c_void_p]),
    "gfs3d_edgeconv_backward_f32": (c_int, [POINTER(EdgeConvBwdArgs), c_void_p]),
    "gfs3d_edgeconv_forward_phase_f32": (c_int, [POINTER(EdgeConvArgs), c_int, c_void_p, c_double, c_void_p]),
    "gfs3d_edgeconv_backward_phase_f32": (c_int, [POINTER(EdgeConvBwdArgs), c_int, c_void_p, c_double, c_void_p]),
    "gfs3d_pointwise_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, POINTER(c_int)]),
    "gfs3d_pointwise_workspace_bytes_ex": (c_size_t, [c_int, c_int, c_int, c_int, c_int, POINTER(c_int)]),
    "gfs3d_linear_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "gfs3d_linear_forward_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, POINTER(c_void_p), c_int, c_void_p, c_size_t,
                                         c_void_p]),
    "gfs3d_linear_backward_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, POINTER(c_void_p), c_void_p, c_void_p, c_int,
                                          c_void_p, c_size_t, c_void_p]),
    "gfs3d_pack_rows_f32": (c_int, [POINTER(PackArgs), c_void_p]),
    "gfs3d_unpack_rows_f32": (c_int, [POINTER(PackArgs), c_void_p]),
    "gfs3d_pointwise_forward_f32": (c_int, [POINTER(PointwiseArgs), c_void_p]),
    "gfs3d_pointwise_backward_f32": (c_int, [POINTER(PointwiseBwdArgs), c_void_p]),
    "gfs3d_pointwise_forward_phase_f32": (c_int, [POINTER(PointwiseArgs), c_int, c_void_p, c_double, c_void_p]),
    "gfs3d_pointwise_backward_phase_f32": (c_int, [POINTER(PointwiseBwdArgs), c_int, c_void_p, c_double, c_void_p]),
    "gfs3d_attention_forward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_ulonglong, c_void_p,
                                            c_void_p, c_void_p, c_void_p]),
    "gfs3d_attention_backward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                             c_float, c_ulonglong, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gfs3d_attention_dropout_mask": (c_int, [c_int, c_int, c_float, c_ulonglong, c_void_p, c_void_p]),
    "gfs3d_attention_forward_mixed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_ulonglong, c_void_p,
                                              c_void_p, c_void_p, c_int, c_void_p]),
    "gfs3d_attention_backward_mixed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                               c_float, c_ulonglong, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gfs3d_word_assign_f32": (c_int, [c_void_p, c_longlong, c_int, c_int, c_int, c_void_p, c_int, c_float,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "gfs3d_proto_cosine_f32": (c_int, [c_void_p, c_longlong, c_int, c_int, c_int, c_void_p, c_int, c_int, c_float,
                                       c_void_p, c_void_p]),
    "gfs3d_proto_backward_feat_f32": (c_int, [c_void_p, c_longlong, c_int, c_int, c_int, c_void_p, c_int, c_int, c_float,
                                              c_void_p, c_void_p, c_void_p, c_void_p]),
    "gfs3d_kmeans_assign_f32": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    "gfs3d_kmeans_workspace_bytes": (c_size_t, [c_longlong, c_int, c_int]),
    "gfs3d_kmeans_assign_ws_f32": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    "gfs3d_kmeans_assign_fast_f32": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_size_t, c_void_p]),
    "gfs3d_kmeans_fix_from_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gfs3d_kmeans_fix_to_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gfs3d_kmeans_fix_step": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gfs3d_kmeans_delta_fix": (c_int, [c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gfs3d_kmeanspp_workspace_bytes": (c_size_t, [c_longlong, c_int]),
    "gfs3d_kmeanspp_seed_f32": (c_int, [c_void_p, c_longlong, c_int, c_int, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                        c_int, c_void_p]),
    "gfs3d_kmeanspp_greedy_workspace_bytes": (c_size_t, [c_longlong, c_int, c_int]),
    "gfs3d_kmeanspp_greedy_f32": (c_int, [c_void_p, c_longlong, c_int, c_int, c_longlong, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                          c_size_t, c_int, c_void_p]),
    "gfs3d_proto_backward_proto_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gfs3d_proto_backward_proto_f32": (c_int, [c_void_p, c_longlong, c_int, c_int, c_int, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p,
                                               c_void_p, c_size_t, c_void_p]),
    "gfs3d_label_presence": (c_int, [c_void_p, c_longlong, c_void_p, c_void_p]),
    "gfs3d_capl_head_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gfs3d_capl_head_eval": (c_int, [POINTER(CaplEvalArgs)]),
    "gfs3d_capl_head_forward_phase": (c_int, [POINTER(CaplHeadArgs), c_int]),
    "gfs3d_capl_head_backward_phase": (c_int, [POINTER(CaplHeadArgs), c_int]),
    "gfs3d_word_assign_mixed": (c_int, [c_void_p, c_longlong, c_int, c_int, c_int, c_void_p, c_int, c_float,
                                        c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gfs3d_adam_multi_f32": (c_int, [POINTER(AdamArgs), c_void_p]),
    "gfs3d_word_soft_backward_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gfs3d_proto_backward_feat_mixed": (c_int, [c_void_p, c_longlong, c_int, c_int, c_int, c_void_p, c_int, c_int, c_float,
                                                c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gfs3d_proto_backward_feat_workspace_bytes": (c_size_t, [c_int, c_int]),
    "gfs3d_proto_backward_feat_ws": (c_int, [c_void_p, c_longlong, c_int, c_int, c_int, c_void_p, c_int, c_int, c_float,
                                             c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
}


def library_path() -> str:
    return _SO


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into lib/libgfs3d_hip.so (hipcc
    cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, s) for s in _SOURCES]
    deps = srcs + [os.path.join(_CSRC, "common.h"), os.path.join(_PKG, "..", "include", "gfs3d.h")]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(d) for d in deps):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value"]
    objdir = os.path.join(os.path.dirname(_SO), "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = deps[len(srcs):]
    jobs = []
    for src in srcs:                                  # one hipcc per source, side by side (a full build: 47 s -> ~20 s)
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in [src] + hdrs):
            cmd = [hipcc, *flags, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    cmd = [hipcc, *flags, "-shared", *[os.path.join(objdir, os.path.basename(s_) + ".o") for s_ in srcs], "-o", _SO + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(_SO + ".tmp", _SO)
    return _SO


def lib():
    """Load the shared object (after ``import torch`` so that the HIP runtime
    torch already mapped - same SONAME libamdhip64.so.7 - is the one we bind to)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(
                f"gfs3d: {_SO} is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path.")
        import torch  # noqa: F401  (maps torch's libamdhip64 first)
        L = ctypes.CDLL(_SO)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)      # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().gfs3d_last_error()
        raise RuntimeError(f"gfs3d: {what} failed (code {rc}): {msg.decode() if msg else ''}")
