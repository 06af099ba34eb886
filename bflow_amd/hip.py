"""ctypes binding of libbflow_hip.so (include/bflow_hip.h) for PyTorch-ROCm tensors.

This is the only place the package touches the C ABI.  Every wrapper
  * requires CUDA(HIP)-resident, contiguous tensors of the declared dtype,
  * passes raw `data_ptr()`s plus sizes, and torch's CURRENT stream (so launches are captured by hipGraph
    stream capture and ordered with whatever else torch enqueues on the same stream),
  * raises `BflowHipError` on a non-zero status.

There is NO fallback: if the shared library is missing or a tensor lives on the CPU the call fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import numpy as np
import torch

# BFLOW_HIP_LIB points at an alternative build of the SAME ABI (A/B timing of two kernel versions inside one gpurun call)
_LIB_PATH = os.environ.get("BFLOW_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libbflow_hip.so")

ABI_VERSION = 2        # include/bflow_hip.h BFLOW_ABI_VERSION this binding was written against (checked on load)
# tools only (A/B timing against a library built from an OLDER revision whose entry points for the timed kernels kept their meaning):
# BFLOW_HIP_ABI_ANY=1 skips the version check and binds the entry points the variant has
_ABI_ANY = bool(os.environ.get("BFLOW_HIP_LIB")) and os.environ.get("BFLOW_HIP_ABI_ANY") == "1"
MAX_PLANES, MAX_TARGETS, MAX_DEGREE, LOOKUP_RADIUS = 16, 8, 16, 4
ACT_NONE, ACT_RELU = 0, 1

EXPORTS = (
    "bflow_version", "bflow_last_error_string", "bflow_corr_build_f32", "bflow_split_pack", "bflow_corr_build_split", "bflow_corr_build_split_tiled", "bflow_corr_build_tiled", "bflow_split_to_x8", "bflow_corr_pool2x2_tiled", "bflow_corr_lookup_bezier_split_tiled", "bflow_corr_build_f16_tiled", "bflow_corr_pool2x2_tiled_f16", "bflow_corr_lookup_bezier_split_tiled_f16", "bflow_conv_pack_weights", "bflow_conv_pack_weights_adjoint", "bflow_conv_split", "bflow_conv_thin_acc", "bflow_conv_thin_mfma_acc", "bflow_wgrad_pack", "bflow_blocked_f32_to_nchw", "bflow_pow2_scale", "bflow_rows_to_split", "bflow_grad_stats", "bflow_wgrad_reduce", "bflow_conv_wgrad_halo", "bflow_conv_wgrad_finish", "bflow_norm_train_finalize", "bflow_norm_train_apply", "bflow_norm_train_bwd_stats", "bflow_norm_train_bwd_finalize", "bflow_norm_train_bwd_apply", "bflow_gru_zr_fwd", "bflow_gru_zr_bwd", "bflow_gru_blend_fwd", "bflow_gru_blend_bwd", "bflow_conv_stem", "bflow_plane_stats", "bflow_norm_act_split", "bflow_split_to_nchw", "bflow_bezier_update", "bflow_im2col_small", "bflow_corr_pool2x2", "bflow_corr_lookup",
    "bflow_corr_lookup_bezier", "bflow_corr_lookup_bezier_split", "bflow_bezier_coeffs", "bflow_bezier_eval", 
    "bflow_cvx_upsample",
    "bflow_clock_stamp", "bflow_shader_clock_stamp", "bflow_voxel_workspace_bytes", "bflow_voxel_grid_f32xy", "bflow_voxel_grid_i16xy", "bflow_voxel_grid_i32xy", "bflow_voxel_norm", "bflow_voxel_merge_norm", "bflow_epe_accumulate",
    "bflow_flow_metrics_accumulate", "bflow_traj_len", "bflow_pad_replicate", "bflow_voxel_grid_rectified", "bflow_voxel_grid_rectified_window", "bflow_maxabs_diff",
    "bflow_corr_lookup_bwd", "bflow_corr_lookup_bezier_bwd", "bflow_corr_pool2x2_bwd", "bflow_cvx_upsample_bwd", "bflow_l1_masked_accumulate",
    "bflow_l1_masked_grad", "bflow_conv_split_pair", "bflow_corr_lookup_im2col", "bflow_cvx_upsample_blocked",
)


def tensor_version(t) -> int:
    """In-place version counter of a tensor for cache keys.  Tensors created under torch.inference_mode() -- what val.py:75 runs the forward
    in: every temporary derived from a weight there is one -- carry no counter (`_version` raises); they cannot be edited in place outside
    inference mode either, so a constant is a valid key component for them."""
    return -1 if t.is_inference() else t._version


class BflowHipError(RuntimeError):
    pass


class PlaneDesc(ctypes.Structure):
    """struct bflow_plane (include/bflow_hip.h)."""
    _fields_ = [("base", ctypes.c_void_p), ("h", ctypes.c_int), ("w", ctypes.c_int),
                ("level", ctypes.c_int), ("target", ctypes.c_int)]


class ConvDesc(ctypes.Structure):
    """struct bflow_conv_desc (include/bflow_hip.h)."""
    _fields_ = [("x_hi", ctypes.c_void_p), ("x_lo", ctypes.c_void_p), ("w_hi", ctypes.c_void_p), ("w_lo", ctypes.c_void_p),
                ("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("C", ctypes.c_int), ("Cout", ctypes.c_int),
                ("cout_pad", ctypes.c_int), ("KH", ctypes.c_int), ("KW", ctypes.c_int), ("stride", ctypes.c_int),
                ("pad_h", ctypes.c_int), ("pad_w", ctypes.c_int), ("tile_n", ctypes.c_int),
                ("out_f32", ctypes.c_void_p), ("out_hi", ctypes.c_void_p), ("out_lo", ctypes.c_void_p),
                ("out_channel_stride", ctypes.c_int), ("out_channel_offset", ctypes.c_int), ("out_rows_per_image", ctypes.c_int),
                ("in_rows_per_image", ctypes.c_int), ("x2_hi", ctypes.c_void_p), ("x2_lo", ctypes.c_void_p),
                ("x_split_channels", ctypes.c_int), ("addend", ctypes.c_void_p),
                ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p), ("act", ctypes.c_int), ("stats", ctypes.c_void_p),
                ("gate", ctypes.c_int), ("gate_h_hi", ctypes.c_void_p), ("gate_h_lo", ctypes.c_void_p), ("gate_z", ctypes.c_void_p),
                ("stats_replicas", ctypes.c_int), ("acc_nchw", ctypes.c_void_p), ("weight_sets", ctypes.c_int),
                ("x_raw", ctypes.c_void_p), ("x_stats", ctypes.c_void_p), ("x_stats_replicas", ctypes.c_int), ("x_eps", ctypes.c_float),
                ("keep_pad_channels", ctypes.c_int)]


class StemDesc(ctypes.Structure):
    """struct bflow_stem_desc (include/bflow_hip.h)."""
    _fields_ = [("x", ctypes.c_void_p), ("w_hi", ctypes.c_void_p), ("w_lo", ctypes.c_void_p),
                ("B", ctypes.c_int), ("Cin", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("Cout", ctypes.c_int),
                ("cout_pad", ctypes.c_int), ("k_blocks", ctypes.c_int), ("ksize", ctypes.c_int), ("stride", ctypes.c_int), ("pad", ctypes.c_int),
                ("out_f32", ctypes.c_void_p), ("out_hi", ctypes.c_void_p), ("out_lo", ctypes.c_void_p), ("out_rows_per_image", ctypes.c_int),
                ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p), ("act", ctypes.c_int), ("stats", ctypes.c_void_p),
                ("stats_replicas", ctypes.c_int), ("n_windows", ctypes.c_int), ("src_channels", ctypes.c_int),
                ("window_starts", ctypes.POINTER(ctypes.c_int)), ("layout", ctypes.c_int),
                ("window_bases", ctypes.POINTER(ctypes.c_void_p)), ("x2", ctypes.c_void_p), ("x2_channels", ctypes.c_int),
                ("x_dtype", ctypes.c_int), ("x2_dtype", ctypes.c_int), ("x_image_norm", ctypes.c_int), ("x2_image_norm", ctypes.c_int)]


class NormDesc(ctypes.Structure):
    """struct bflow_norm_desc (include/bflow_hip.h)."""
    _fields_ = [("a", ctypes.c_void_p), ("stats_a", ctypes.c_void_p), ("scale_a", ctypes.c_void_p), ("shift_a", ctypes.c_void_p),
                ("a_is_nchw", ctypes.c_int), ("act_a", ctypes.c_int), ("b", ctypes.c_void_p), ("stats_b", ctypes.c_void_p),
                ("res_hi", ctypes.c_void_p), ("res_lo", ctypes.c_void_p), ("act_out", ctypes.c_int),
                ("out_hi", ctypes.c_void_p), ("out_lo", ctypes.c_void_p), ("out_f32", ctypes.c_void_p),
                ("B", ctypes.c_int), ("HW", ctypes.c_int), ("C", ctypes.c_int), ("eps", ctypes.c_float), ("rows_per_image", ctypes.c_int),
                ("stats_replicas", ctypes.c_int), ("act_b", ctypes.c_int)]


_lib = None


def library_path() -> str:
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    """Loads libbflow_hip.so once.  Raises if it has not been built (python -m bflow_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(_LIB_PATH):
        raise BflowHipError(f"{_LIB_PATH} is missing: build it with `python -m bflow_amd.build` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = ctypes.CDLL(_LIB_PATH)
    vp, i, ll, f = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float
    L.bflow_version.restype = i
    L.bflow_version.argtypes = []
    L.bflow_last_error_string.restype = ctypes.c_char_p
    L.bflow_last_error_string.argtypes = []
    if L.bflow_version() != ABI_VERSION and not _ABI_ANY:      # a stale build (or a BFLOW_HIP_LIB variant of another round): entry points changed meaning
        raise BflowHipError(f"{_LIB_PATH} reports ABI version {L.bflow_version()}, this binding needs {ABI_VERSION}: rebuild it "
                            "(python -m bflow_amd.build)")
    sig = {
        "bflow_corr_build_f32": [vp, vp, vp, i, i, i, i, ll, vp],
        "bflow_split_pack": [vp, vp, vp, i, i, i, i, vp],
        "bflow_corr_build_split": [vp, vp, vp, vp, vp, i, i, i, i, i, ll, vp],
        "bflow_corr_build_split_tiled": [vp, vp, vp, vp, vp, i, i, i, i, i, i, ll, vp],
        "bflow_corr_pool2x2_tiled": [vp, vp, ll, i, i, vp],
        "bflow_corr_pool2x2_tiled_f16": [vp, vp, ll, i, i, vp],
        "bflow_corr_build_f16_tiled": [vp, vp, vp, i, i, i, i, i, i, ll, vp],
        "bflow_corr_build_tiled": [vp, vp, vp, vp, vp, i, i, i, i, i, i, ll, i, i, vp, ctypes.POINTER(ctypes.c_int), vp],
        "bflow_split_to_x8": [vp, vp, vp, ll, vp],
        "bflow_corr_lookup_bezier_split_tiled_f16": [ctypes.POINTER(PlaneDesc), i, vp, ctypes.POINTER(ctypes.c_float), i, i, vp, vp, i, i, i, i, i, vp],
        "bflow_corr_lookup_bezier_split_tiled": [ctypes.POINTER(PlaneDesc), i, vp, ctypes.POINTER(ctypes.c_float), i, i, vp, vp, i, i, i, i, i, vp],
        "bflow_conv_pack_weights": [vp, vp, vp, i, i, i, i, i, i, vp],
        "bflow_conv_pack_weights_adjoint": [vp, vp, vp, i, i, i, i, i, i, vp],
        "bflow_conv_stem": [ctypes.POINTER(StemDesc), vp],
        "bflow_conv_split": [ctypes.POINTER(ConvDesc), vp],
        "bflow_conv_split_pair": [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc), ctypes.POINTER(ctypes.c_int), vp],
        "bflow_corr_lookup_im2col": [ctypes.POINTER(PlaneDesc), i, vp, ctypes.POINTER(ctypes.c_float), i, i, vp, vp, i, i, i, i, i, i, vp, vp, i, i, i, i, i, vp],
        "bflow_conv_thin_acc": [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, i, i, i, vp],
        "bflow_conv_thin_mfma_acc": [vp, vp, vp, vp, i, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, i, vp],
        "bflow_wgrad_pack": [vp, vp, vp, i, i, i, i, i, i, i, i, i, i, i, i, i, i, vp, vp],
        "bflow_blocked_f32_to_nchw": [vp, vp, i, i, i, i, i, vp, vp],
        "bflow_pow2_scale": [vp, ll, f, vp, vp, vp],
        "bflow_rows_to_split": [vp, vp, vp, i, i, i, vp, vp],
        "bflow_grad_stats": [vp, i, i, i, f, vp, vp, vp, vp],
        "bflow_wgrad_reduce": [vp, vp, i, i, i, i, i, i, i, vp, vp],
        "bflow_conv_wgrad_halo": [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, vp],
        "bflow_conv_wgrad_finish": [vp, vp, i, i, i, i, vp, vp],
        "bflow_norm_train_finalize": [vp, i, i, i, i, f, vp, vp, vp, vp, f, vp, vp, vp, vp, vp],
        "bflow_norm_train_apply": [vp, vp, vp, vp, ll, i, i, vp],
        "bflow_norm_train_bwd_stats": [vp, vp, vp, vp, vp, vp, vp, ll, i, i, vp],
        "bflow_norm_train_bwd_finalize": [vp, i, i, i, i, vp, vp, vp, vp, vp],
        "bflow_norm_train_bwd_apply": [vp, vp, vp, vp, vp, vp, vp, vp, vp, ll, i, i, vp],
        "bflow_gru_zr_fwd": [vp, vp, vp, vp, vp, i, i, ll, vp],
        "bflow_gru_zr_bwd": [vp, vp, vp, vp, vp, vp, vp, i, i, ll, vp],
        "bflow_gru_blend_fwd": [vp, vp, vp, vp, vp, i, i, ll, vp],
        "bflow_gru_blend_bwd": [vp, vp, vp, vp, vp, vp, vp, i, i, ll, vp],
        "bflow_plane_stats": [vp, vp, ll, i, vp],
        "bflow_norm_act_split": [ctypes.POINTER(NormDesc), vp],
        "bflow_split_to_nchw": [vp, vp, vp, i, i, i, i, i, ll, vp],
        "bflow_bezier_update": [vp, vp, i, vp, vp, i, i, vp, vp, i, i, i, i, i, vp],
        "bflow_im2col_small": [vp, vp, vp, i, i, i, i, i, i, i, i, i, vp],
        "bflow_corr_pool2x2": [vp, vp, ll, i, i, vp],
        "bflow_corr_lookup": [ctypes.POINTER(PlaneDesc), i, vp, vp, i, i, i, i, vp],
        "bflow_corr_lookup_bezier": [ctypes.POINTER(PlaneDesc), i, vp, ctypes.POINTER(ctypes.c_float), i, i, vp, i, i, i, vp],
        "bflow_corr_lookup_bezier_split": [ctypes.POINTER(PlaneDesc), i, vp, ctypes.POINTER(ctypes.c_float), i, i, vp, vp, i, i, i, i, i, vp],
        "bflow_bezier_coeffs": [ctypes.POINTER(ctypes.c_double), i, i, ctypes.POINTER(ctypes.c_float)],
        "bflow_bezier_eval": [vp, ctypes.POINTER(ctypes.c_float), i, i, i, i, i, i, vp, vp],
        "bflow_cvx_upsample": [vp, vp, vp, f, vp, i, i, i, i, vp],
        "bflow_cvx_upsample_blocked": [vp, vp, f, vp, i, i, i, i, i, vp],
        "bflow_clock_stamp": [vp, vp],
        "bflow_shader_clock_stamp": [vp, vp],
        "bflow_voxel_workspace_bytes": [ll, i, i, i, i],
        "bflow_voxel_grid_f32xy": [vp, vp, vp, vp, ll, ll, ll, vp, i, i, i, vp, ll, vp],
        "bflow_voxel_grid_i16xy": [vp, vp, vp, vp, ll, ll, ll, vp, i, i, i, vp, ll, vp],
        "bflow_voxel_grid_i32xy": [vp, vp, vp, vp, ll, ll, ll, vp, i, i, i, vp, ll, vp],
        "bflow_voxel_norm": [vp, ll, vp, vp],
        "bflow_voxel_merge_norm": [vp, ll, vp, ll, vp, vp, vp],
        "bflow_voxel_grid_rectified": [vp, vp, vp, vp, ll, vp, ll, ll, vp, i, i, i, vp, vp, ll, vp],
        "bflow_voxel_grid_rectified_window": [vp, vp, vp, vp, ll, ll, vp, vp, vp, i, i, i, vp, vp, ll, vp],
        "bflow_maxabs_diff": [vp, vp, ll, vp, vp],
        "bflow_epe_accumulate": [vp, vp, vp, i, i, ll, vp, vp],
        "bflow_flow_metrics_accumulate": [vp, vp, vp, i, i, ll, f, f, f, vp, vp],
        "bflow_traj_len": [vp, vp, i, i, i, ll, vp],
        "bflow_pad_replicate": [vp, vp, ll, i, i, i, i, i, i, vp],
        "bflow_corr_lookup_bwd": [ctypes.POINTER(PlaneDesc), ctypes.POINTER(vp), i, vp, i, vp, vp, i, i, i, vp],
        "bflow_corr_lookup_bezier_bwd": [ctypes.POINTER(PlaneDesc), ctypes.POINTER(vp), i, vp, ctypes.POINTER(ctypes.c_float), i, i, vp, vp, i, i, i, vp],
        "bflow_corr_pool2x2_bwd": [vp, vp, ll, i, i, vp],
        "bflow_cvx_upsample_bwd": [vp, vp, vp, vp, vp, vp, i, i, i, i, vp],
        "bflow_l1_masked_accumulate": [vp, vp, vp, i, i, ll, vp, vp],
        "bflow_l1_masked_grad": [vp, vp, vp, i, i, ll, vp, vp, f, vp, vp],
    }
    for name, args in sig.items():
        if _ABI_ANY and not hasattr(L, name):
            continue
        fn = getattr(L, name)
        fn.restype = ll if name == "bflow_voxel_workspace_bytes" else i
        fn.argtypes = args
    _lib = L
    return L


def _check(rc: int, what: str):
    if rc != 0:
        msg = lib().bflow_last_error_string().decode("utf-8", "replace")
        raise BflowHipError(f"{what} failed (status {rc}): {msg}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, dtype=torch.float32, name: str = "tensor", contiguous: bool = True) -> int:
    if not isinstance(t, torch.Tensor):
        raise BflowHipError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise BflowHipError(f"{name}: tensor is on {t.device}; the HIP path needs a GPU tensor (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise BflowHipError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if contiguous and not t.is_contiguous():
        raise BflowHipError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def _slice(t: torch.Tensor, name: str):
    """(ptr, batch_stride) of a (B, C, ...) channel slice of a contiguous NCHW buffer."""
    if not t.is_cuda or t.dtype != torch.float32:
        raise BflowHipError(f"{name}: needs a float32 GPU tensor")
    B = t.shape[0]
    inner = t[0] if B > 0 else t
    if not inner.is_contiguous():
        raise BflowHipError(f"{name}: per-sample (C, H, W) block must be contiguous")
    return t.data_ptr(), (t.stride(0) if B > 1 else int(np.prod(t.shape[1:])))


def _opt(t: Optional[torch.Tensor], name: str) -> Optional[int]:
    return None if t is None else _dev(t, torch.float32, name)


# ------------------------------------------------------------------------------------------------ K5 / K6 / K7
def corr_build_f32(f1: torch.Tensor, f2: torch.Tensor, out: torch.Tensor):
    """f1 (B,D,N) [shared] or (T,B,D,N); f2 (T,B,D,N); out (T,B,N,N)."""
    T, B, D, N = f2.shape
    if f1.dim() == 3:
        assert f1.shape == (B, D, N)
        tstride = 0
    else:
        assert f1.shape == (T, B, D, N)
        tstride = B * D * N
    assert out.shape == (T, B, N, N)
    _check(lib().bflow_corr_build_f32(_dev(f1, name="f1"), _dev(f2, name="f2"), _dev(out, name="out"), T, B, D, N, tstride, _stream()),
           "bflow_corr_build_f32")


def padded_rows(n: int, tile: int = 128) -> int:
    return (n + tile - 1) // tile * tile


def split_pack(src: torch.Tensor) -> torch.Tensor:
    """src (R, D, N) fp32 -> (2, R, D/32, Np, 32) fp16 k-blocked operands: [0] = hi, [1] = lo (x ~= hi + lo * 2^-11),
    rows N..Np zero."""
    R, D, N = src.shape
    Np = padded_rows(N)
    out = torch.empty((2, R, D // 32, Np, 32), dtype=torch.float16, device=src.device)
    _check(lib().bflow_split_pack(_dev(src, name="src"), out[0].data_ptr(), out[1].data_ptr(), R, D, N, Np, _stream()), "bflow_split_pack")
    return out


TILE_H, TILE_W = 4, 8    # tiled volume planes (include/bflow_hip.h, bflow_corr_build_split_tiled)


def tiled_plane_size(h: int, w: int) -> int:
    """Elements of one h x w plane in the tiled layout (4 x 8 tiles, edge tiles padded)."""
    return ((h + TILE_H - 1) // TILE_H) * ((w + TILE_W - 1) // TILE_W) * TILE_H * TILE_W


def untile_planes(t: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """(..., tiled_plane_size(h, w)) -> (..., h, w) row-major copy (debug / reference-shaped accessors only)."""
    th, tw = (h + TILE_H - 1) // TILE_H, (w + TILE_W - 1) // TILE_W
    lead = t.shape[:-1]
    v = t.reshape(*lead, th, tw, TILE_H, TILE_W).transpose(-3, -2).reshape(*lead, th * TILE_H, tw * TILE_W)
    return v[..., :h, :w].contiguous()


ARITH_SPLIT, ARITH_F16, ARITH_SPLIT8 = 0, 1, 2    # bflow_corr_build_tiled `arithmetic`


def split_to_x8(p: torch.Tensor) -> torch.Tensor:
    """p (2, R, KB, Np, 32) fp16 split planes -> (R, KB, Np, 64) uint8 "x8" plane: per (row, 32-channel block) [e4m3(hi) x 32 | e4m3(lo) x 32],
    the second operand plane of the fp8-cross-term correlation (bflow_corr_build_tiled, arithmetic = 2)."""
    assert p.dtype == torch.float16 and p.is_cuda and p.shape[0] == 2 and p.shape[-1] == 32 and p[0].is_contiguous() and p[1].is_contiguous()
    out = torch.empty(tuple(p.shape[1:-1]) + (64,), dtype=torch.uint8, device=p.device)
    _check(lib().bflow_split_to_x8(p[0].data_ptr(), p[1].data_ptr(), out.data_ptr(), p[0].numel() // 32, _stream()), "bflow_split_to_x8")
    return out


def pool_fusable(arithmetic: int, fp16_volume: bool, D: int, T: int) -> bool:
    """Combinations for which bflow_corr_build_tiled writes the level-1 planes itself (`pool=`)."""
    return D in (128, 256) and T <= 8 and ((arithmetic in (ARITH_SPLIT, ARITH_SPLIT8) and not fp16_volume) or (arithmetic == ARITH_F16 and fp16_volume))


def corr_build_tiled(p1: torch.Tensor, p2: torch.Tensor, out: torch.Tensor, T: int, B: int, N: int, shared_f1: bool, tiled_hw: Sequence[int],
                     arithmetic: int = ARITH_SPLIT, x8: Optional[Sequence[torch.Tensor]] = None, pool=None):
    """The tiled volume with any arithmetic (ARITH_*) and fp32 or fp16 storage (out.dtype).  ARITH_SPLIT8 takes the x8 planes of both
    operands (`x8 = (split_to_x8(p1), split_to_x8(p2))`; computed here when omitted).
    pool = (level1, index): level1 (T1, B, N, tiled_plane_size(h//2, w//2)) of out.dtype receives the 2 x 2 mean (K6, level 0 -> 1) of the
    level-0 planes of every target t with index[t] >= 0 (its row in level1), written by the same launch (pool_fusable(...) combinations)."""
    _, R2, KB, Np, _32 = p2.shape
    D = KB * 32
    h, w = int(tiled_hw[0]), int(tiled_hw[1])
    assert R2 == T * B and p1.shape[1] == (B if shared_f1 else T * B) and h * w == N
    assert p1.dtype == torch.float16 and p2.dtype == torch.float16 and p1.is_cuda and p2.is_cuda
    assert p1[0].is_contiguous() and p1[1].is_contiguous() and p2[0].is_contiguous() and p2[1].is_contiguous()
    assert out.shape == (T, B, N, tiled_plane_size(h, w)) and out.is_cuda and out.is_contiguous() and out.dtype in (torch.float16, torch.float32)
    if arithmetic == ARITH_SPLIT8:
        if x8 is None:
            x8 = (split_to_x8(p1), split_to_x8(p2))
        s1, s2 = x8[0].data_ptr(), x8[1].data_ptr()
    else:
        s1, s2 = p1[1].data_ptr(), p2[1].data_ptr()
    pout, pidx = None, None
    if pool is not None:
        lvl1, index = pool
        assert len(index) == T and pool_fusable(arithmetic, out.dtype == torch.float16, D, T)
        assert lvl1.dtype == out.dtype and lvl1.is_cuda and lvl1.is_contiguous() and tuple(lvl1.shape[1:]) == (B, N, tiled_plane_size(h // 2, w // 2))
        assert all(-1 <= k < lvl1.shape[0] for k in index)
        pout, pidx = lvl1.data_ptr(), (ctypes.c_int * T)(*[int(k) for k in index])
    _check(lib().bflow_corr_build_tiled(p1[0].data_ptr(), s1, p2[0].data_ptr(), s2, out.data_ptr(), T, B, D, h, w, Np,
                                        0 if shared_f1 else B * Np * D, int(arithmetic), 1 if out.dtype == torch.float16 else 0, pout, pidx, _stream()),
           "bflow_corr_build_tiled")


def corr_build_split(p1: torch.Tensor, p2: torch.Tensor, out: torch.Tensor, T: int, B: int, N: int, shared_f1: bool,
                     tiled_hw: Optional[Sequence[int]] = None):
    """p1 = split_pack(f1 viewed (B or T*B, D, N)), p2 = split_pack(f2 viewed (T*B, D, N)); out (T, B, N, N), or with
    tiled_hw = (h, w), h*w == N: out (T, B, N, tiled_plane_size(h, w)) -- every plane stored as 4 x 8 tiles."""
    _, R2, KB, Np, _32 = p2.shape
    D = KB * 32
    assert R2 == T * B and p1.shape[1] == (B if shared_f1 else T * B)
    assert p1.dtype == torch.float16 and p2.dtype == torch.float16 and p1.is_cuda and p2.is_cuda
    assert p1[0].is_contiguous() and p1[1].is_contiguous() and p2[0].is_contiguous() and p2[1].is_contiguous()
    if out.dtype == torch.float16:
        assert tiled_hw is not None, "the fp16 volume exists in the tiled layout only"
        h, w = int(tiled_hw[0]), int(tiled_hw[1])
        assert h * w == N and out.shape == (T, B, N, tiled_plane_size(h, w)) and out.is_cuda and out.is_contiguous()
        _check(lib().bflow_corr_build_f16_tiled(p1[0].data_ptr(), p2[0].data_ptr(), out.data_ptr(), T, B, D, h, w, Np,
                                                0 if shared_f1 else B * Np * D, _stream()), "bflow_corr_build_f16_tiled")
    elif tiled_hw is None:
        assert out.shape == (T, B, N, N)
        _check(lib().bflow_corr_build_split(p1[0].data_ptr(), p1[1].data_ptr(), p2[0].data_ptr(), p2[1].data_ptr(), _dev(out, name="out"),
                                            T, B, D, N, Np, 0 if shared_f1 else B * Np * D, _stream()), "bflow_corr_build_split")
    else:
        h, w = int(tiled_hw[0]), int(tiled_hw[1])
        assert h * w == N and out.shape == (T, B, N, tiled_plane_size(h, w))
        _check(lib().bflow_corr_build_split_tiled(p1[0].data_ptr(), p1[1].data_ptr(), p2[0].data_ptr(), p2[1].data_ptr(), _dev(out, name="out"),
                                                  T, B, D, h, w, Np, 0 if shared_f1 else B * Np * D, _stream()), "bflow_corr_build_split_tiled")


def corr_pool2x2_tiled(src: torch.Tensor, dst: torch.Tensor, h: int, w: int):
    """src (planes, tiled_plane_size(h, w)) -> dst (planes, tiled_plane_size(h//2, w//2)): 2x2 mean on tiled planes."""
    planes = src.numel() // tiled_plane_size(h, w)
    assert src.shape[-1] == tiled_plane_size(h, w) and dst.shape[-1] == tiled_plane_size(h // 2, w // 2) and dst.numel() == planes * dst.shape[-1]
    if src.dtype == torch.float16:
        assert dst.dtype == torch.float16 and src.is_cuda and src.is_contiguous() and dst.is_contiguous()
        _check(lib().bflow_corr_pool2x2_tiled_f16(src.data_ptr(), dst.data_ptr(), planes, h, w, _stream()), "bflow_corr_pool2x2_tiled_f16")
        return
    _check(lib().bflow_corr_pool2x2_tiled(_dev(src, name="src"), _dev(dst, name="dst"), planes, h, w, _stream()), "bflow_corr_pool2x2_tiled")


def corr_pool2x2(src: torch.Tensor, dst: torch.Tensor):
    """src (..., h, w) -> dst (..., h//2, w//2)."""
    h, w = src.shape[-2:]
    planes = src.numel() // (h * w)
    assert dst.shape[-2:] == (h // 2, w // 2) and dst.numel() == planes * (h // 2) * (w // 2)
    _check(lib().bflow_corr_pool2x2(_dev(src, name="src"), _dev(dst, name="dst"), planes, h, w, _stream()), "bflow_corr_pool2x2")


def make_plane_table(planes: Sequence[dict]):
    """planes: [{tensor: (B*N, h, w) slab -- or (B*N, tiled size) with "hw": (h, w) for tiled planes --, level: int, target: int}]
    -> ctypes array (keeps no reference to tensors)."""
    arr = (PlaneDesc * len(planes))()
    for k, p in enumerate(planes):
        t = p["tensor"]
        arr[k].base = _dev(t, None, name=f"plane{k}")     # fp32, or fp16 for the tiled fp16 volume
        hw = p.get("hw")
        arr[k].h, arr[k].w = (int(t.shape[-2]), int(t.shape[-1])) if hw is None else (int(hw[0]), int(hw[1]))
        arr[k].level, arr[k].target = int(p["level"]), int(p["target"])
    return arr


def corr_lookup(table, coords: torch.Tensor, out: torch.Tensor):
    T, B, two, h1, w1 = coords.shape
    assert two == 2
    P = len(table)
    assert out.shape == (B, P * 81, h1, w1)
    _check(lib().bflow_corr_lookup(table, P, _dev(coords, name="coords"), _dev(out, name="out"), T, B, h1, w1, _stream()),
           "bflow_corr_lookup")


def corr_lookup_bezier(table, params: torch.Tensor, coef: np.ndarray, out: torch.Tensor):
    B, C2, h1, w1 = params.shape
    T, deg = coef.shape
    assert C2 == 2 * deg and coef.dtype == np.float32 and coef.flags["C_CONTIGUOUS"]
    P = len(table)
    assert out.shape == (B, P * 81, h1, w1)
    _check(lib().bflow_corr_lookup_bezier(table, P, _dev(params, name="params"), coef.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                          T, deg, _dev(out, name="out"), B, h1, w1, _stream()), "bflow_corr_lookup_bezier")


def corr_lookup_bezier_split(table, params: torch.Tensor, coef: np.ndarray, out_planes: torch.Tensor, tiled: bool = False,
                             f16_planes: bool = False, im2col=None):
    """out_planes: (2, B, CBk, rows, 32) fp16 (hi, lo), zero-initialised once by the caller (pad channels are never written by the row-major
    kernel).  tiled: the plane table describes tiled planes (bflow_corr_build_split_tiled / bflow_corr_pool2x2_tiled)."""
    B, C2, h1, w1 = params.shape
    T, deg = coef.shape
    assert C2 == 2 * deg and coef.dtype == np.float32 and coef.flags["C_CONTIGUOUS"]
    P = len(table)
    assert out_planes.dtype == torch.float16 and out_planes.is_contiguous() and out_planes.shape[0] == 2 and out_planes.shape[1] == B \
        and out_planes.shape[4] == 32 and out_planes.shape[2] * 32 >= P * 81 and out_planes.shape[3] >= h1 * w1
    assert tiled or not f16_planes
    if im2col is not None:
        # + the filter windows of the same parameters (bflow_im2col_small) as the first workgroups of the look-up launch
        col_planes, kh, kw, ph, pw = im2col
        assert tiled and col_planes.dtype == torch.float16 and col_planes.is_contiguous() and col_planes.shape[0] == 2 and col_planes.shape[1] == B \
            and col_planes.shape[2] == (kh * kw * C2 + 31) // 32 and col_planes.shape[3] >= h1 * w1 and col_planes.shape[4] == 32
        _check(lib().bflow_corr_lookup_im2col(table, P, _dev(params, name="params"), coef.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), T, deg,
                                              out_planes[0].data_ptr(), out_planes[1].data_ptr(), out_planes.shape[2], out_planes.shape[3], B, h1, w1,
                                              int(f16_planes), col_planes[0].data_ptr(), col_planes[1].data_ptr(), kh, kw, ph, pw, col_planes.shape[3],
                                              _stream()), "bflow_corr_lookup_im2col")
        return
    fn = lib().bflow_corr_lookup_bezier_split_tiled_f16 if f16_planes else lib().bflow_corr_lookup_bezier_split_tiled if tiled \
        else lib().bflow_corr_lookup_bezier_split
    _check(fn(table, P, _dev(params, name="params"), coef.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
              T, deg, out_planes[0].data_ptr(), out_planes[1].data_ptr(), out_planes.shape[2],
              out_planes.shape[3], B, h1, w1, _stream()), "bflow_corr_lookup_bezier_split" + ("_tiled" if tiled else ""))


def bezier_coeffs(times: Sequence[float], degree: int) -> np.ndarray:
    ts = np.ascontiguousarray(np.asarray(times, dtype=np.float64).reshape(-1))
    out = np.empty((ts.size, degree), dtype=np.float32)
    _check(lib().bflow_bezier_coeffs(ts.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ts.size, degree,
                                     out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))), "bflow_bezier_coeffs")
    return out


def bezier_eval(params: torch.Tensor, coef: np.ndarray, add_coords0: bool = False) -> torch.Tensor:
    B, C2, h, w = params.shape
    T, deg = coef.shape
    assert C2 == 2 * deg and coef.dtype == np.float32
    coef = np.ascontiguousarray(coef)
    out = torch.empty((T, B, 2, h, w), dtype=torch.float32, device=params.device)
    _check(lib().bflow_bezier_eval(_dev(params, name="params"), coef.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), T, deg, B, h, w,
                                   int(add_coords0), _dev(out), _stream()), "bflow_bezier_eval")
    return out


# ------------------------------------------------------------------------------------------------ K13
def cvx_upsample_blocked(data: torch.Tensor, mask_blocked: torch.Tensor, mask_scale: float = 1.0) -> torch.Tensor:
    """cvx_upsample on the mask in the conv engine's blocked fp32 layout (B, 18, rows, 32) (bias included)."""
    B, C, h, w = data.shape
    assert mask_blocked.dtype == torch.float32 and mask_blocked.is_contiguous() and mask_blocked.shape[:2] == (B, 18) and mask_blocked.shape[3] == 32 \
        and mask_blocked.shape[2] >= h * w
    out = torch.empty((B, C, 8 * h, 8 * w), dtype=torch.float32, device=data.device)
    _check(lib().bflow_cvx_upsample_blocked(_dev(data, name="data"), _dev(mask_blocked, name="mask"), float(mask_scale), _dev(out), B, C, h, w,
                                            mask_blocked.shape[2], _stream()), "bflow_cvx_upsample_blocked")
    return out


def cvx_upsample(data: torch.Tensor, mask: torch.Tensor, mask_bias: Optional[torch.Tensor] = None, mask_scale: float = 1.0) -> torch.Tensor:
    B, C, h, w = data.shape
    assert mask.shape == (B, 576, h, w)
    out = torch.empty((B, C, 8 * h, 8 * w), dtype=torch.float32, device=data.device)
    _check(lib().bflow_cvx_upsample(_dev(data, name="data"), _dev(mask, name="mask"), _opt(mask_bias, "mask_bias"), float(mask_scale),
                                    _dev(out), B, C, h, w, _stream()), "bflow_cvx_upsample")
    return out


CLOCK_TABLE_ROWS = 1024      # include/bflow_hip.h BFLOW_CLOCK_TABLE_ROWS


def shader_clock_tables(n: int, device) -> torch.Tensor:
    """Zeroed (n, CLOCK_TABLE_ROWS, 2) int64: one table per stamp."""
    return torch.zeros((n, CLOCK_TABLE_ROWS, 2), dtype=torch.int64, device=device)


def shader_clock_stamp(tables: torch.Tensor, index: int):
    """Writes (s_memtime, s_memrealtime) of every CU into tables[index] on the current stream (capture-safe)."""
    assert tables.dtype == torch.int64 and tables.dim() == 3 and tuple(tables.shape[1:]) == (CLOCK_TABLE_ROWS, 2) and 0 <= index < tables.shape[0]
    _check(lib().bflow_shader_clock_stamp(_dev(tables, torch.int64, "tables") + 16 * CLOCK_TABLE_ROWS * index, _stream()), "bflow_shader_clock_stamp")


def shader_clock_ghz(tables: torch.Tensor, first: int = 0, last: int = 1) -> float:
    """Average shader clock (GHz) between two stamps: per CU cycles / 100-MHz ticks x 0.1, median over the CUs both stamps reached
    (s_memtime is a per-CU counter: only differences on the same CU mean anything)."""
    t = tables.cpu()
    a, b = t[first], t[last]
    ok = (a[:, 1] != 0) & (b[:, 1] != 0) & (b[:, 1] > a[:, 1])
    if not bool(ok.any()):
        return float("nan")
    ghz = (b[ok, 0] - a[ok, 0]).double() / (b[ok, 1] - a[ok, 1]).double() * 0.1
    return float(ghz.median())


def clock_stamp(slots: torch.Tensor, index: int):
    """Writes the device's 100 MHz wall clock into slots[index] (int64) on the current stream (capture-safe measurement hook)."""
    assert slots.dtype == torch.int64 and 0 <= index < slots.numel()
    _check(lib().bflow_clock_stamp(_dev(slots, torch.int64, "slots") + 8 * index, _stream()), "bflow_clock_stamp")


# ------------------------------------------------------------------------------------------------ K1 / K2
_VOXEL_WS = {}      # (device index, stream) -> grow-only byte buffer


def _voxel_workspace(n: int, C: int, H: int, W: int, float_xy: bool, device) -> torch.Tensor:
    """K1's workspace (counts, bases and one 16-B record per (event, bin): up to 8 records per float-xy event with more than 8 bins --
    ~256 MB for a 2 M-event, 15-bin window).  ONE grow-only buffer per (device, stream) instead of an allocation per call: launches
    of one stream are ordered, so consecutive calls may share it (a buffer that had to grow is replaced; the old one stays alive until its
    last launch has run, as any torch allocation does)."""
    nbytes = int(lib().bflow_voxel_workspace_bytes(n, C, H, W, int(float_xy)))
    if nbytes < 0:
        raise BflowHipError("voxel grid: " + lib().bflow_last_error_string().decode("utf-8", "replace"))
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream())
    buf = _VOXEL_WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if len(_VOXEL_WS) > 8:       # streams come and go: keep the table small
            _VOXEL_WS.clear()
        buf = torch.empty(nbytes + nbytes // 4, dtype=torch.uint8, device=device)
        _VOXEL_WS[key] = buf
    return buf


def voxel_workspace(n: int, C: int, H: int, W: int, float_xy: bool, device) -> torch.Tensor:
    """A caller-owned K1 workspace for windows of at most `n` events (captured launch sequences keep theirs for the graph's lifetime)."""
    nbytes = int(lib().bflow_voxel_workspace_bytes(n, C, H, W, int(float_xy)))
    if nbytes < 0:
        raise BflowHipError("voxel grid: " + lib().bflow_last_error_string().decode("utf-8", "replace"))
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def voxel_grid(x: torch.Tensor, y: torch.Tensor, pol: torch.Tensor, t: torch.Tensor, t0_center: int, t1_center: int,
               grid: torch.Tensor, workspace: Optional[torch.Tensor] = None):
    """K1: `grid` (C, H, W) is written whole (it need not be zeroed)."""
    C, H, W = grid.shape
    n = x.numel()
    ppol = _dev(pol, torch.int8, "pol")
    pt = _dev(t, torch.int64, "time")
    if x.dtype == torch.float32:
        fn, px, py = lib().bflow_voxel_grid_f32xy, _dev(x, torch.float32, "x"), _dev(y, torch.float32, "y")
    elif x.dtype == torch.int16:
        fn, px, py = lib().bflow_voxel_grid_i16xy, _dev(x, torch.int16, "x"), _dev(y, torch.int16, "y")
    elif x.dtype == torch.int32:
        fn, px, py = lib().bflow_voxel_grid_i32xy, _dev(x, torch.int32, "x"), _dev(y, torch.int32, "y")
    else:
        raise BflowHipError(f"voxel_grid: x/y dtype {x.dtype} unsupported (float32, int16 or int32)")
    if workspace is None:
        workspace = _voxel_workspace(n, C, H, W, x.dtype == torch.float32, grid.device)
    _check(fn(px, py, ppol, pt, n, int(t0_center), int(t1_center), _dev(grid, name="grid"), C, H, W, _dev(workspace, torch.uint8, "workspace"),
              workspace.numel(), _stream()), "bflow_voxel_grid")


def voxel_grid_rectified(x: torch.Tensor, y: torch.Tensor, pol: torch.Tensor, t: torch.Tensor, rectify_map: torch.Tensor,
                         t0_center: int, t1_center: int, grid: torch.Tensor, bad_count: Optional[torch.Tensor] = None,
                         workspace: Optional[torch.Tensor] = None):
    """Raw uint16 sensor coordinates -> rectify_map[y, x] -> tri-linear accumulation (f-1); `grid` is written whole."""
    C, H, W = grid.shape
    assert tuple(rectify_map.shape) == (H, W, 2)
    if workspace is None:
        workspace = _voxel_workspace(x.numel(), C, H, W, True, grid.device)
    _check(lib().bflow_voxel_grid_rectified(_dev(x, torch.uint16, "x"), _dev(y, torch.uint16, "y"), _dev(pol, torch.uint8, "pol"),
                                            _dev(t, torch.int64, "time"), x.numel(), _dev(rectify_map, name="rectify_map"), int(t0_center),
                                            int(t1_center), _dev(grid, name="grid"), C, H, W,
                                            None if bad_count is None else _dev(bad_count, torch.int32, "bad_count"),
                                            _dev(workspace, torch.uint8, "workspace"), workspace.numel(), _stream()),
           "bflow_voxel_grid_rectified")


def voxel_grid_rectified_window(x: torch.Tensor, y: torch.Tensor, pol: torch.Tensor, t: torch.Tensor, window: torch.Tensor, max_events: int,
                                rectify_map: torch.Tensor, grid: torch.Tensor, workspace: torch.Tensor, bad_count: Optional[torch.Tensor] = None):
    """bflow_voxel_grid_rectified with the window {first, count, t0_center, t1_center} read from the device tensor `window` (4 int64) at run
    time: hipGraph-capturable, one capture for every frame of a recording.  `workspace`: hip.voxel_workspace(max_events, ...) owned by the caller."""
    C, H, W = grid.shape
    assert tuple(rectify_map.shape) == (H, W, 2) and window.dtype == torch.int64 and window.numel() == 4 and window.is_contiguous()
    _check(lib().bflow_voxel_grid_rectified_window(_dev(x, torch.uint16, "x"), _dev(y, torch.uint16, "y"), _dev(pol, torch.uint8, "pol"),
                                                   _dev(t, torch.int64, "time"), x.numel(), int(max_events), _dev(window, torch.int64, "window"),
                                                   _dev(rectify_map, name="rectify_map"), _dev(grid, name="grid"), C, H, W,
                                                   None if bad_count is None else _dev(bad_count, torch.int32, "bad_count"),
                                                   _dev(workspace, torch.uint8, "workspace"), workspace.numel(), _stream()),
           "bflow_voxel_grid_rectified_window")


def maxabs_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """0-dim float32 GPU tensor max |a - b|."""
    assert a.shape == b.shape
    out = torch.zeros((), dtype=torch.float32, device=a.device)
    _check(lib().bflow_maxabs_diff(_dev(a, name="a"), _dev(b, name="b"), a.numel(), _dev(out), _stream()), "bflow_maxabs_diff")
    return out


VOXEL_NORM_WS_DOUBLES = 3072      # include/bflow_hip.h BFLOW_VOXEL_NORM_WS_DOUBLES


def voxel_norm_workspace(device) -> torch.Tensor:
    return torch.empty(VOXEL_NORM_WS_DOUBLES, dtype=torch.float64, device=device)


def voxel_norm(grid: torch.Tensor, workspace: Optional[torch.Tensor] = None):
    if workspace is None or workspace.numel() < VOXEL_NORM_WS_DOUBLES:
        workspace = voxel_norm_workspace(grid.device)
    _check(lib().bflow_voxel_norm(_dev(grid, name="grid"), grid.numel(), _dev(workspace, torch.float64, "workspace"), _stream()),
           "bflow_voxel_norm")


def voxel_merge_norm(a: torch.Tensor, b: Optional[torch.Tensor], out: torch.Tensor, workspace: Optional[torch.Tensor] = None):
    """out = norm_voxel_grid(cat(a.flatten(), b.flatten())) (representations.py:9-18 on the merged grid of twostep.py:77-85); out may be a."""
    if workspace is None or workspace.numel() < VOXEL_NORM_WS_DOUBLES:
        workspace = voxel_norm_workspace(out.device)
    nb = 0 if b is None else b.numel()
    assert out.numel() == a.numel() + nb and out.is_contiguous()
    _check(lib().bflow_voxel_merge_norm(_dev(a, name="a"), a.numel(), None if b is None else _dev(b, name="b"), nb, _dev(out, name="out"),
                                        _dev(workspace, torch.float64, "workspace"), _stream()), "bflow_voxel_merge_norm")
    return out


# ------------------------------------------------------------------------------------------------ K15
def epe_accumulate(pred: torch.Tensor, gt: torch.Tensor, valid: Optional[torch.Tensor], acc: torch.Tensor):
    B, C = pred.shape[:2]
    HW = int(np.prod(pred.shape[2:]))
    assert pred.shape == gt.shape
    pv = None
    if valid is not None:
        assert valid.dtype in (torch.bool, torch.uint8) and valid.numel() == B * HW
        pv = _dev(valid.view(torch.uint8) if valid.dtype == torch.bool else valid, torch.uint8, "valid")
    _check(lib().bflow_epe_accumulate(_dev(pred, name="pred"), _dev(gt, name="gt"), pv, B, C, HW, _dev(acc, torch.float64, "acc"), _stream()),
           "bflow_epe_accumulate")


def flow_metrics_accumulate(pred: torch.Tensor, gt: torch.Tensor, valid: Optional[torch.Tensor], n_pixels: Sequence[float], acc: torch.Tensor):
    """acc (6,) float64 += [sum epe, #valid, sum angular error (rad), #n-pixel errors for the 3 thresholds]."""
    B, C = pred.shape[:2]
    HW = int(np.prod(pred.shape[2:]))
    assert pred.shape == gt.shape and len(n_pixels) == 3 and acc.numel() == 6
    pv = None
    if valid is not None:
        assert valid.dtype in (torch.bool, torch.uint8) and valid.numel() == B * HW
        pv = _dev(valid.view(torch.uint8) if valid.dtype == torch.bool else valid, torch.uint8, "valid")
    _check(lib().bflow_flow_metrics_accumulate(_dev(pred, name="pred"), _dev(gt, name="gt"), pv, B, C, HW, float(n_pixels[0]), float(n_pixels[1]),
                                               float(n_pixels[2]), _dev(acc, torch.float64, "acc"), _stream()), "bflow_flow_metrics_accumulate")


def traj_len(targets: torch.Tensor) -> torch.Tensor:
    """targets (M, B, C, *) -> (B, *) summed length of the per-pixel polyline."""
    M, B, C = targets.shape[:3]
    out = torch.empty((B,) + tuple(targets.shape[3:]), dtype=torch.float32, device=targets.device)
    _check(lib().bflow_traj_len(_dev(targets, name="targets"), _dev(out), M, B, C, int(np.prod(targets.shape[3:])), _stream()), "bflow_traj_len")
    return out


def pad_replicate(x: torch.Tensor, pad: Sequence[int]) -> torch.Tensor:
    """F.pad(x, [left, right, top, bottom], mode='replicate') on the last two dims."""
    H, W = x.shape[-2:]
    pl, pr, pt, pb = (int(v) for v in pad)
    out = torch.empty(tuple(x.shape[:-2]) + (H + pt + pb, W + pl + pr), dtype=torch.float32, device=x.device)
    _check(lib().bflow_pad_replicate(_dev(x, name="x"), _dev(out), int(np.prod(x.shape[:-2])), H, W, pl, pr, pt, pb, _stream()),
           "bflow_pad_replicate")
    return out


# ----------------------------------------------------------------------------------------------------------------------
# Two-branch concurrency.  The batch-1 kernels of the update block launch <= 240 workgroups for 256 CUs and independent
# branches (correlation features vs. Bezier features in the motion encoder, context encoder vs. feature encoder) exist,
# so a branch is issued on a side stream between fork() and join().  Under hipGraph capture the stream dependencies become
# graph edges; in eager mode they are stream waits.  Buffers that cross the branches are owned by the caller's workspace
# (alive until after the join), branch-local temporaries are stream-ordered on the side stream.
# ----------------------------------------------------------------------------------------------------------------------
_side_streams = {}
BRANCHING = True        # tools: False issues every branch on the current stream


# ------------------------------------------------------------------------------------------------ training path (f-4)
_const_cache: dict = {}
_captured_consts: dict = {}


def const_tensor(arr, device) -> torch.Tensor:
    """Device copy of a small host constant (Bezier coefficient tables), cached by value: no host-to-device copy after the first
    use -- a pageable copy is not allowed inside a stream capture (training.GraphedTrainStep warms the cache before it captures)."""
    import numpy as np
    a = np.ascontiguousarray(arr)
    key = (a.dtype.str, a.shape, a.tobytes(), str(device))
    t = _const_cache.pop(key, None)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            # a miss here would issue a pageable host-to-device copy inside the capture, and an eviction could free a table an earlier
            # graph still points at: both are silent failures -- refuse (warm the cache with one eager step first)
            raise BflowHipError("const_tensor: constant not cached while a stream capture is in progress (run the step once eagerly first)")
        while len(_const_cache) >= 256:          # least recently used first (dict order = recency, see the re-insert below); never all at once
            _const_cache.pop(next(iter(_const_cache)))
        t = torch.from_numpy(a.copy()).to(device)
    _const_cache[key] = t                       # (re-)insert as most recent
    if torch.cuda.is_current_stream_capturing():
        # a captured graph keeps only the raw pointer of the table: pin the tensor for the life of the process (a few hundred bytes each), so
        # that a later eviction can never hand its memory to somebody else while a graph still replays on it
        _captured_consts[id(t)] = t
    return t


def make_grad_table(tensors: Sequence[torch.Tensor]):
    """HOST array of device pointers to the per-plane gradient slabs (same order as make_plane_table)."""
    arr = (ctypes.c_void_p * len(tensors))()
    for k, t in enumerate(tensors):
        arr[k] = _dev(t, name=f"grad_plane{k}")
    return arr


def corr_lookup_bwd(table, grad_table, coords: torch.Tensor, grad_out: torch.Tensor) -> torch.Tensor:
    """-> per-plane coordinate gradients (P, B, 2, h1, w1); the plane gradients are accumulated in place."""
    T, B, two, h1, w1 = coords.shape
    P = len(table)
    assert two == 2 and grad_out.shape == (B, P * 81, h1, w1) and len(grad_table) == P
    gc = torch.empty((P, B, 2, h1, w1), dtype=torch.float32, device=coords.device)
    _check(lib().bflow_corr_lookup_bwd(table, grad_table, P, _dev(coords, name="coords"), T, _dev(grad_out, name="grad_out"), _dev(gc),
                                       B, h1, w1, _stream()), "bflow_corr_lookup_bwd")
    return gc


def corr_lookup_bezier_bwd(table, grad_table, params: torch.Tensor, coef: np.ndarray, grad_out: torch.Tensor) -> torch.Tensor:
    B, C2, h1, w1 = params.shape
    T, deg = coef.shape
    P = len(table)
    assert C2 == 2 * deg and coef.dtype == np.float32 and coef.flags["C_CONTIGUOUS"]
    assert grad_out.shape == (B, P * 81, h1, w1) and len(grad_table) == P
    gc = torch.empty((P, B, 2, h1, w1), dtype=torch.float32, device=params.device)
    _check(lib().bflow_corr_lookup_bezier_bwd(table, grad_table, P, _dev(params, name="params"),
                                              coef.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), T, deg, _dev(grad_out, name="grad_out"),
                                              _dev(gc), B, h1, w1, _stream()), "bflow_corr_lookup_bezier_bwd")
    return gc


def corr_pool2x2_bwd(grad_cur: torch.Tensor, grad_prev: torch.Tensor):
    """grad_prev (M, h, w) += 0.25 * grad_cur (M, h//2, w//2) spread over the 2x2 cells."""
    M, h, w = grad_prev.shape
    assert grad_cur.shape == (M, h // 2, w // 2)
    _check(lib().bflow_corr_pool2x2_bwd(_dev(grad_cur, name="grad_cur"), _dev(grad_prev, name="grad_prev"), M, h, w, _stream()),
           "bflow_corr_pool2x2_bwd")


def cvx_upsample_bwd(grad_up: torch.Tensor, data: torch.Tensor, mask: torch.Tensor):
    B, C, h, w = data.shape
    assert mask.shape == (B, 576, h, w) and grad_up.shape == (B, C, 8 * h, 8 * w)
    gd = torch.empty_like(data, dtype=torch.float32)
    gm = torch.empty_like(mask, dtype=torch.float32)
    scratch = torch.empty(B * C * 72 * h * w, dtype=torch.float32, device=data.device)
    _check(lib().bflow_cvx_upsample_bwd(_dev(grad_up, name="grad_up"), _dev(data, name="data"), _dev(mask, name="mask"), _dev(gd), _dev(gm),
                                        _dev(scratch), B, C, h, w, _stream()), "bflow_cvx_upsample_bwd")
    return gd, gm


def _valid_ptr(valid: Optional[torch.Tensor], count: int):
    if valid is None:
        return None
    assert valid.dtype in (torch.bool, torch.uint8) and valid.numel() == count
    return _dev(valid.view(torch.uint8) if valid.dtype == torch.bool else valid, torch.uint8, "valid")


def l1_masked_accumulate(src: torch.Tensor, tgt: torch.Tensor, valid: Optional[torch.Tensor], acc: torch.Tensor):
    B, C = src.shape[:2]
    HW = int(np.prod(src.shape[2:]))
    assert src.shape == tgt.shape
    _check(lib().bflow_l1_masked_accumulate(_dev(src, name="src"), _dev(tgt, name="tgt"), _valid_ptr(valid, B * HW), B, C, HW,
                                            _dev(acc, torch.float64, "acc"), _stream()), "bflow_l1_masked_accumulate")


def l1_masked_grad(src: torch.Tensor, tgt: torch.Tensor, valid: Optional[torch.Tensor], acc: torch.Tensor,
                   upstream: Optional[torch.Tensor], weight: float) -> torch.Tensor:
    B, C = src.shape[:2]
    HW = int(np.prod(src.shape[2:]))
    grad = torch.empty_like(src)
    _check(lib().bflow_l1_masked_grad(_dev(src, name="src"), _dev(tgt, name="tgt"), _valid_ptr(valid, B * HW), B, C, HW,
                                      _dev(acc, torch.float64, "acc"), _opt(upstream, "upstream"), float(weight), _dev(grad), _stream()),
           "bflow_l1_masked_grad")
    return grad


class Branch:
    """`with Branch(enabled) as br:` runs the block on the side stream; `br.join()` makes the current stream wait for it."""

    def __init__(self, enabled: bool = True):
        self.enabled = enabled and BRANCHING and torch.cuda.is_available()
        self._ctx = None
        if self.enabled:
            self.main = torch.cuda.current_stream()
            key = (self.main.device_index, self.main.cuda_stream)      # one side stream per stream that forks: branches nest
            if key not in _side_streams:
                _side_streams[key] = torch.cuda.Stream(device=self.main.device)
            self.side = _side_streams[key]

    def __enter__(self):
        if self.enabled:
            self.side.wait_stream(self.main)
            self._ctx = torch.cuda.stream(self.side)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            self._ctx.__exit__(*exc)
        return False

    def join(self):
        if self.enabled:
            self.main.wait_stream(self.side)


# ------------------------------------------------------------------------------------------------- SepConvGRU gates of the training path
def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().float().contiguous()


def gru_zr_fwd(zr_pre: torch.Tensor, h: torch.Tensor):
    """(z, r, r*h) from the merged z | r pre-activations (B, 2C, H, W) and the hidden state (B, C, H, W): bflow_gru_zr_fwd."""
    zr_pre, h = _f32c(zr_pre), _f32c(h)
    B, C, H, W = h.shape
    assert zr_pre.shape == (B, 2 * C, H, W)
    z, r, rh = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
    _check(lib().bflow_gru_zr_fwd(_dev(zr_pre, name="zr_pre"), _dev(h, name="h"), _dev(z), _dev(r), _dev(rh), B, C, H * W, _stream()), "bflow_gru_zr_fwd")
    return z, r, rh


def gru_zr_bwd(dz: Optional[torch.Tensor], drh: torch.Tensor, z: torch.Tensor, r: torch.Tensor, h: torch.Tensor):
    """-> (d zr_pre (B, 2C, H, W), dh through r*h): bflow_gru_zr_bwd."""
    B, C, H, W = h.shape
    drh = _f32c(drh)
    dz = None if dz is None else _f32c(dz)
    dzr = torch.empty((B, 2 * C, H, W), dtype=torch.float32, device=h.device)
    dh = torch.empty_like(h)
    _check(lib().bflow_gru_zr_bwd(None if dz is None else _dev(dz, name="dz"), _dev(drh, name="drh"), _dev(z), _dev(r), _dev(h), _dev(dzr), _dev(dh),
                                  B, C, H * W, _stream()), "bflow_gru_zr_bwd")
    return dzr, dh


def gru_blend_fwd(q_pre: torch.Tensor, z: torch.Tensor, h: torch.Tensor):
    """(q = tanh(q_pre), h_new = (1 - z) h + z q): bflow_gru_blend_fwd."""
    q_pre, z, h = _f32c(q_pre), _f32c(z), _f32c(h)
    B, C, H, W = h.shape
    q, hn = torch.empty_like(h), torch.empty_like(h)
    _check(lib().bflow_gru_blend_fwd(_dev(q_pre, name="q_pre"), _dev(z, name="z"), _dev(h, name="h"), _dev(q), _dev(hn), B, C, H * W, _stream()),
           "bflow_gru_blend_fwd")
    return q, hn


def gru_blend_bwd(dhn: torch.Tensor, q: torch.Tensor, z: torch.Tensor, h: torch.Tensor):
    """-> (d q_pre, dz, dh through the blend): bflow_gru_blend_bwd."""
    dhn = _f32c(dhn)
    B, C, H, W = h.shape
    dq, dz, dh = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
    _check(lib().bflow_gru_blend_bwd(_dev(dhn, name="dh_new"), _dev(q), _dev(z), _dev(h), _dev(dq), _dev(dz), _dev(dh), B, C, H * W, _stream()),
           "bflow_gru_blend_bwd")
    return dq, dz, dh
