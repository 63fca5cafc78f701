"""ctypes bindings of the in-tree HIP libraries (include/gsr.h, include/galbs.h).

There is NO fallback: if a library is missing the import of the op fails loudly — run
`python -m gaussianavatar_amd.build` (or `__graft_entry__.build()`) first.
"""
from __future__ import annotations

import ctypes
import os

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so). It must be the FIRST HIP runtime
# in the process so that the in-tree libraries bind to the same one (same devices, streams and
# allocations as the tensors they are handed); loading /opt/rocm's copy first gives the process
# two runtimes and "no ROCm-capable device is detected".
import torch  # noqa: F401  (import order matters, see above)
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.path.join(_HERE, "_lib")


class NativeLibraryMissing(RuntimeError):
    pass


class GsrSettings(ctypes.Structure):
    """struct GsrSettings (include/gsr.h)."""
    _fields_ = [
        ("image_height", c_int32), ("image_width", c_int32),
        ("tanfovx", c_float), ("tanfovy", c_float), ("scale_modifier", c_float),
        ("sh_degree", c_int32), ("prefiltered", c_int32), ("debug", c_int32),
        ("bg", c_void_p), ("viewmatrix", c_void_p), ("projmatrix", c_void_p), ("campos", c_void_p),
    ]


class GsrBatch(ctypes.Structure):
    """struct GsrBatch (include/gsr.h): element strides between frames, 0 = shared."""
    _fields_ = [("frames", c_int32), ("means3D_stride", c_int64), ("colors_stride", c_int64),
                ("opacities_stride", c_int64), ("scales_stride", c_int64), ("rotations_stride", c_int64),
                ("cov3D_stride", c_int64), ("viewmatrix_stride", c_int64), ("projmatrix_stride", c_int64),
                ("shs_stride", c_int64), ("campos_stride", c_int64)]


_LAYOUT_FIELDS = ["total_bytes", "depth", "xy", "conic_opacity", "rgb", "cov3d", "rect",
                  "tiles_touched", "clamped", "tile_count", "tile_offset", "tile_cursor",
                  "pair_key", "point_list", "pair_tmp", "final_T", "n_contrib", "grad_acc", "status", "seg_heads",
                  "seg_count", "xyext", "seg_entries", "seg_ckpt", "seg_info", "pix_accum", "pair_grad", "seg_list",
                  "eval_bytes", "train_bytes", "sort_work"]

GSR_WS_EVAL, GSR_WS_TRAIN, GSR_WS_DEBUG = 0, 1, 2


class GsrLayout(ctypes.Structure):
    """struct GsrLayout (include/gsr.h)."""
    _fields_ = [(n, c_uint64) for n in _LAYOUT_FIELDS]


def _load(name: str) -> ctypes.CDLL:
    from ._dev import knobs                     # GA_DEV=lib_dir=...: A/B an alternative build (development only)
    path = os.path.join(knobs.lib_dir or _LIBDIR, name)
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            f"{path} not found: the HIP extension is not built. Run "
            f"`python -m gaussianavatar_amd.build` (needs hipcc). There is no CPU fallback.")
    return ctypes.CDLL(path)


_gsr = None
_galbs = None

GSR_SYMBOLS = ["gsr_workspace_bytes", "gsr_workspace_bytes_for", "gsr_workspace_layout", "gsr_forward", "gsr_backward",
               "gsr_forward_eval", "gsr_forward_eval_batch",
               "gsr_mark_visible", "gsr_batch_status", "gsr_read_status", "gsr_last_error", "gsr_abi_version", "gsr_set_trace",
               "gsr_profile_create", "gsr_profile_destroy", "gsr_profile_bind", "gsr_profile_read",
               "gsr_profile_kernel_name", "gsr_render_block_edge",
               "gsr_forward_batch", "gsr_backward_batch"]
GALBS_SYMBOLS = ["galbs_joint_saved_floats", "galbs_joint_transforms_fwd",
                 "galbs_joint_transforms_bwd", "galbs_skin_fwd", "galbs_skin_bwd",
                 "galbs_last_error", "galbs_abi_version"]


def gsr() -> ctypes.CDLL:
    global _gsr
    if _gsr is None:
        lib = _load("libgsr_hip.so")
        P = c_void_p
        lib.gsr_workspace_bytes.restype = c_size_t
        lib.gsr_workspace_bytes.argtypes = [c_int32, c_int32, c_int32, c_int64]
        lib.gsr_workspace_bytes_for.restype = c_size_t
        lib.gsr_workspace_bytes_for.argtypes = [c_int32, c_int32, c_int32, c_int64, c_int32]
        lib.gsr_workspace_layout.restype = c_int
        lib.gsr_workspace_layout.argtypes = [c_int32, c_int32, c_int32, c_int64, ctypes.POINTER(GsrLayout)]
        lib.gsr_forward.restype = c_int
        lib.gsr_forward.argtypes = [ctypes.POINTER(GsrSettings), c_int32, P, P, P, c_int32, P, P, P, P,
                                    P, c_size_t, c_int64, P, P, P]
        lib.gsr_forward_eval.restype = c_int
        lib.gsr_forward_eval.argtypes = lib.gsr_forward.argtypes
        lib.gsr_backward.restype = c_int
        lib.gsr_backward.argtypes = [ctypes.POINTER(GsrSettings), c_int32, P, P, P, c_int32, P, P, P, P,
                                     P, P, c_size_t, c_int64, P, P, P, P, P, P, P, P, P, P, P]
        B_ = ctypes.POINTER(GsrBatch)
        lib.gsr_forward_batch.restype = c_int
        lib.gsr_forward_batch.argtypes = [ctypes.POINTER(GsrSettings), B_, c_int32, P, P, P, c_int32, P, P, P, P,
                                          P, c_size_t, c_int64, P, P, P]
        lib.gsr_forward_eval_batch.restype = c_int
        lib.gsr_forward_eval_batch.argtypes = lib.gsr_forward_batch.argtypes
        lib.gsr_backward_batch.restype = c_int
        lib.gsr_backward_batch.argtypes = [ctypes.POINTER(GsrSettings), B_, c_int32, P, P, P, c_int32, P, P, P, P,
                                           P, P, c_size_t, c_int64, P, P, P, P, P, P, P, P, P, P, P]
        lib.gsr_mark_visible.restype = c_int
        lib.gsr_mark_visible.argtypes = [c_int32, P, P, P, P, P]
        lib.gsr_batch_status.restype = c_int
        lib.gsr_batch_status.argtypes = [P, c_int32, c_int32, c_int32, c_int32, c_int64, c_int32, P, P]
        lib.gsr_read_status.restype = c_int
        lib.gsr_read_status.argtypes = [P, c_int32, c_int32, c_int32, c_int64, P, P]
        lib.gsr_last_error.restype = c_char_p
        lib.gsr_abi_version.restype = c_int
        lib.gsr_profile_create.restype = c_void_p
        lib.gsr_profile_create.argtypes = []
        lib.gsr_profile_destroy.restype = None
        lib.gsr_profile_destroy.argtypes = [P]
        lib.gsr_profile_bind.restype = c_int
        lib.gsr_profile_bind.argtypes = [P, c_int]
        lib.gsr_profile_read.restype = c_int
        lib.gsr_profile_read.argtypes = [P, P, P, c_int]
        lib.gsr_render_block_edge.restype = c_int
        lib.gsr_profile_kernel_name.restype = c_char_p
        lib.gsr_profile_kernel_name.argtypes = [c_int]
        if lib.gsr_abi_version() != 6:
            raise RuntimeError("libgsr_hip.so ABI version mismatch; rebuild")
        _gsr = lib
    return _gsr


def galbs() -> ctypes.CDLL:
    global _galbs
    if _galbs is None:
        lib = _load("libgalbs_hip.so")
        P = c_void_p
        lib.galbs_joint_saved_floats.restype = c_size_t
        lib.galbs_joint_saved_floats.argtypes = [c_int32]
        lib.galbs_joint_transforms_fwd.restype = c_int
        lib.galbs_joint_transforms_fwd.argtypes = [c_int32, c_int32, P, P, P, P, P, c_int64, P, P, P, P]
        lib.galbs_joint_transforms_bwd.restype = c_int
        lib.galbs_joint_transforms_bwd.argtypes = [c_int32, c_int32, P, P, P, P, c_int64, P, P, P, P, P, P]
        lib.galbs_skin_fwd.restype = c_int
        lib.galbs_skin_fwd.argtypes = [c_int32, c_int32, c_int32, P, c_int64, P, c_int64, P, c_int64, P, P, P]
        lib.galbs_skin_bwd.restype = c_int
        lib.galbs_skin_bwd.argtypes = [c_int32, c_int32, c_int32, P, c_int64, P, c_int64, P, c_int64, P, P,
                                       P, P, P]
        lib.galbs_last_error.restype = c_char_p
        lib.galbs_abi_version.restype = c_int
        if lib.galbs_abi_version() != 1:
            raise RuntimeError("libgalbs_hip.so ABI version mismatch; rebuild")
        _galbs = lib
    return _galbs


_ganet = None
GANET_SYMBOLS = ["ganet_linear_wgrad_workspace", "ganet_linear_wgrad", "ganet_bn_workspace",
                 "ganet_bn_act_fwd", "ganet_bn_act_bwd", "ganet_ssim_sums_floats", "ganet_ssim_fwd", "ganet_ssim_bwd",
                 "ganet_mlp_stats_floats", "ganet_mlp_fwd", "ganet_mlp_stats",
                 "ganet_wgrad_act_workspace", "ganet_wgrad_act", "ganet_wgrad_reduce_batch", "ganet_adam_step", "ganet_flag_clear", "ganet_mlp_bwd_data_parts",
                 "ganet_mlp_head_bwd_parts", "ganet_mlp_bwd_data", "ganet_mlp_head_bwd", "ganet_mlp_bwd_stats",
                 "ganet_mlp_bwd_fused_parts", "ganet_mlp_bwd_fused_workspace", "ganet_mlp_bwd_fused", "ganet_mlp_bwd_fused_input",
                 "ganet_decoder_saved_floats", "ganet_decoder_fwd_workspace", "ganet_decoder_fwd",
                 "ganet_decoder_bwd_workspace", "ganet_decoder_bwd",
                 "ganet_decode_pack_fwd", "ganet_decode_pack_bwd", "ganet_records_bwd", "ganet_mean_sq_fwd", "ganet_mean_sq_bwd", "ganet_weighted_sum_fwd", "ganet_weighted_sum_bwd", "ganet_upsample_cat_fwd", "ganet_upsample_cat_bwd",
                 "ganet_unet_saved_floats", "ganet_unet_fwd_workspace", "ganet_unet_fwd",
                 "ganet_unet_bwd_workspace", "ganet_unet_bwd", "ganet_profile_create", "ganet_profile_destroy", "ganet_profile_bind", "ganet_profile_count", "ganet_profile_read", "ganet_profile_kernel_name",
                 "ganet_conv5_packed_bytes", "ganet_conv5_pack", "ganet_conv5_apply",
                 "ganet_conv5_wgrad_workspace", "ganet_conv5_wgrad", "ganet_last_error", "ganet_abi_version"]


class GanetWgradJob(ctypes.Structure):
    """include/ganet.h GanetWgradJob"""
    _fields_ = [("workspace", ctypes.c_void_p), ("M", ctypes.c_int64), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
                ("dW", ctypes.c_void_p), ("db", ctypes.c_void_p), ("nblocks", ctypes.c_int32)]


GANET_DEC_LAYERS = 11


class GanetDecoderParams(ctypes.Structure):
    """include/ganet.h GanetDecoderParams"""
    _L = GANET_DEC_LAYERS
    _fields_ = [("cin", c_int32), ("W", c_void_p * _L), ("bias", c_void_p * _L), ("gamma", c_void_p * _L),
                ("beta", c_void_p * _L), ("running_mean", c_void_p * _L), ("running_var", c_void_p * _L),
                ("num_batches_tracked", c_void_p * _L), ("eps", c_float * _L), ("momentum", c_float * _L),
                ("W8", c_void_p * 3), ("b8", c_void_p * 3), ("n8", c_int32 * 3)]


class GanetDecoderGrads(ctypes.Structure):
    """include/ganet.h GanetDecoderGrads"""
    _L = GANET_DEC_LAYERS
    _fields_ = [("dW", c_void_p * _L), ("db", c_void_p * _L), ("dgamma", c_void_p * _L), ("dbeta", c_void_p * _L),
                ("dW8", c_void_p * 3), ("db8", c_void_p * 3), ("dx", c_void_p), ("x_cols", c_int32)]


class GanetUnetParams(ctypes.Structure):
    """include/ganet.h GanetUnetParams"""
    _fields_ = [("cin", c_int32), ("nf", c_int32), ("cout", c_int32), ("S", c_int32),
                ("Wd", c_void_p * 5), ("Wu", c_void_p * 5), ("bias5", c_void_p),
                ("running_mean", c_void_p * 7), ("running_var", c_void_p * 7), ("num_batches_tracked", c_void_p * 7),
                ("eps", c_float), ("momentum", c_float)]


class GanetUnetGrads(ctypes.Structure):
    """include/ganet.h GanetUnetGrads"""
    _fields_ = [("dWd", c_void_p * 5), ("dWu", c_void_p * 5), ("dbias5", c_void_p)]


class GanetAdamTensor(ctypes.Structure):
    """include/ganet.h GanetAdamTensor"""
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("n", ctypes.c_int64), ("lr", ctypes.c_float),
                ("bias_correction1", ctypes.c_float), ("bias_correction2", ctypes.c_float)]


def ganet() -> ctypes.CDLL:
    global _ganet
    if _ganet is None:
        lib = _load("libganet_hip.so")
        P = c_void_p
        lib.ganet_linear_wgrad_workspace.restype = c_size_t
        lib.ganet_linear_wgrad_workspace.argtypes = [c_int64, c_int32, c_int32]
        lib.ganet_linear_wgrad.restype = c_int
        lib.ganet_linear_wgrad.argtypes = [c_int64, c_int32, c_int32, P, c_int64, P, c_int64, P, P, P, c_size_t, P]
        lib.ganet_bn_workspace.restype = c_size_t
        lib.ganet_bn_workspace.argtypes = [c_int64, c_int32]
        lib.ganet_bn_act_fwd.restype = c_int
        lib.ganet_bn_act_fwd.argtypes = [c_int64, c_int32, P, P, P, c_float, c_int32, P, P, P, P, P, c_float, P, P,
                                         c_size_t, P]
        lib.ganet_bn_act_bwd.restype = c_int
        lib.ganet_bn_act_bwd.argtypes = [c_int64, c_int32, P, P, P, P, P, c_int32, P, P, P, P, P, c_size_t, P]
        lib.ganet_ssim_sums_floats.restype = c_int64
        lib.ganet_ssim_sums_floats.argtypes = []
        lib.ganet_ssim_fwd.restype = c_int
        lib.ganet_ssim_fwd.argtypes = [c_int32, c_int32, c_int32, P, P, c_float, P, P, P]
        lib.ganet_ssim_bwd.restype = c_int
        lib.ganet_ssim_bwd.argtypes = [c_int32, c_int32, c_int32, P, P, P, c_float, P, P, P, P]
        lib.ganet_mlp_stats_floats.restype = c_size_t
        lib.ganet_mlp_stats_floats.argtypes = [c_int32]
        lib.ganet_mlp_fwd.restype = c_int
        lib.ganet_mlp_fwd.argtypes = [c_int64, c_int32, c_int32, c_int32, P, c_int64, P, c_int64, P, P, P, P, P,
                                      c_int64, P, P, c_int32, P]
        lib.ganet_mlp_stats.restype = c_int
        lib.ganet_mlp_stats.argtypes = [c_int64, c_int32, P, P, P, c_float, P, P, P, P, P, P, c_float, P, P, P]
        lib.ganet_wgrad_act_workspace.restype = c_size_t
        lib.ganet_wgrad_act_workspace.argtypes = [c_int64, c_int32, c_int32]
        lib.ganet_wgrad_act.restype = c_int
        lib.ganet_wgrad_act.argtypes = [c_int64, c_int32, c_int32, P, c_int64, P, c_int64, P, P, c_int64, P, P, P,
                                        P, P, c_size_t, c_int32, P]
        lib.ganet_mlp_bwd_data_parts.restype = c_int32
        lib.ganet_mlp_head_bwd_parts.restype = c_int32
        lib.ganet_mlp_bwd_data.restype = c_int
        lib.ganet_mlp_bwd_data.argtypes = [c_int64, c_int32, P, c_int64, P, c_int64, P, P, c_int64, P, c_int64, c_int32,
                                           P, c_int64, P, P, P, c_int32, P]
        lib.ganet_mlp_bwd_fused_parts.restype = c_int32
        lib.ganet_mlp_bwd_fused_workspace.restype = c_size_t
        lib.ganet_mlp_bwd_fused.restype = c_int
        lib.ganet_mlp_bwd_fused.argtypes = [c_int64, P, P, P, P, c_int64, P, c_int32, P, P, P, c_int32, P, P, c_size_t,
                                            c_int32, P]
        lib.ganet_decoder_saved_floats.restype = c_size_t
        lib.ganet_decoder_saved_floats.argtypes = [c_int64]
        lib.ganet_decoder_fwd_workspace.restype = c_size_t
        lib.ganet_decoder_fwd.restype = c_int
        lib.ganet_decoder_fwd.argtypes = [c_int64, P, P, P, P, P, c_size_t, P]
        lib.ganet_decoder_bwd_workspace.restype = c_size_t
        lib.ganet_decoder_bwd_workspace.argtypes = [c_int64]
        lib.ganet_decoder_bwd.restype = c_int
        lib.ganet_decoder_bwd.argtypes = [c_int64, P, P, P, P, P, P, c_size_t, P, P]
        lib.ganet_mlp_bwd_fused_input.restype = c_int
        lib.ganet_mlp_bwd_fused_input.argtypes = [c_int64, P, P, P, P, c_int64, c_int32, P, c_int64, c_int32, P, P, c_size_t,
                                                  c_int32, P]
        lib.ganet_mlp_head_bwd.restype = c_int
        lib.ganet_mlp_head_bwd.argtypes = [c_int64, c_int32, P, P, P, c_int64, P, P, P, c_int64, P, P, P]
        lib.ganet_mlp_bwd_stats.restype = c_int
        lib.ganet_mlp_bwd_stats.argtypes = [c_int64, c_int32, P, P, P, P, P, P, P, P]
        lib.ganet_decode_pack_fwd.restype = c_int
        lib.ganet_decode_pack_fwd.argtypes = [c_int32, c_int64, c_int64, P, P, P, P, c_float, c_float, c_float, c_float, P, P, P]
        lib.ganet_records_bwd.restype = c_int
        lib.ganet_records_bwd.argtypes = [c_int32, c_int32, c_int64, P, P, P, P, P]
        lib.ganet_decode_pack_bwd.restype = c_int
        lib.ganet_decode_pack_bwd.argtypes = [c_int32, c_int64, c_int64, P, P, P, P, c_float, c_float, c_float, c_float, P, P, P, P, P, P, P]
        lib.ganet_wgrad_reduce_batch.restype = c_int
        lib.ganet_wgrad_reduce_batch.argtypes = [c_int32, P, P]
        lib.ganet_adam_step.restype = c_int
        lib.ganet_adam_step.argtypes = [c_int32, P, c_float, c_float, c_float, P, P]
        lib.ganet_flag_clear.restype = c_int
        lib.ganet_flag_clear.argtypes = [P, P]
        lib.ganet_mean_sq_fwd.restype = c_int
        lib.ganet_mean_sq_fwd.argtypes = [c_int64, P, c_float, P, P]
        lib.ganet_mean_sq_bwd.restype = c_int
        lib.ganet_mean_sq_bwd.argtypes = [c_int64, P, c_float, P, P, P]
        lib.ganet_weighted_sum_fwd.restype = c_int
        lib.ganet_weighted_sum_fwd.argtypes = [c_int32, P, P, c_float, P, P]
        lib.ganet_weighted_sum_bwd.restype = c_int
        lib.ganet_weighted_sum_bwd.argtypes = [c_int32, P, P, P, P]
        lib.ganet_upsample_cat_fwd.restype = c_int
        lib.ganet_upsample_cat_fwd.argtypes = [c_int32, c_int32, c_int32, c_int32, P, P, P, P, P, P, P, c_int64, P]
        lib.ganet_upsample_cat_bwd.restype = c_int
        lib.ganet_upsample_cat_bwd.argtypes = [c_int32, c_int32, c_int32, c_int32, P, c_int64, P, P, P, P, P, P, P, P, P]
        lib.ganet_unet_saved_floats.restype = c_size_t
        lib.ganet_unet_saved_floats.argtypes = [P, c_int32]
        lib.ganet_unet_fwd_workspace.restype = c_size_t
        lib.ganet_unet_fwd_workspace.argtypes = [P, c_int32]
        lib.ganet_unet_fwd.restype = c_int
        lib.ganet_unet_fwd.argtypes = [P, c_int32, P, c_int32, P, P, P, c_size_t, P]
        lib.ganet_unet_bwd_workspace.restype = c_size_t
        lib.ganet_unet_bwd_workspace.argtypes = [P, c_int32]
        lib.ganet_unet_bwd.restype = c_int
        lib.ganet_unet_bwd.argtypes = [P, c_int32, P, P, P, P, P, c_size_t, P, P]
        lib.ganet_profile_create.restype = c_void_p
        lib.ganet_profile_create.argtypes = []
        lib.ganet_profile_destroy.restype = None
        lib.ganet_profile_destroy.argtypes = [P]
        lib.ganet_profile_bind.restype = c_int
        lib.ganet_profile_bind.argtypes = [P, c_int]
        lib.ganet_profile_count.restype = c_int
        lib.ganet_profile_read.restype = c_int
        lib.ganet_profile_read.argtypes = [P, P, P, c_int]
        lib.ganet_profile_kernel_name.restype = c_char_p
        lib.ganet_profile_kernel_name.argtypes = [c_int]
        lib.ganet_conv5_packed_bytes.restype = c_size_t
        lib.ganet_conv5_packed_bytes.argtypes = [c_int32]
        lib.ganet_conv5_pack.restype = c_int
        lib.ganet_conv5_pack.argtypes = [c_int32, P, P, P]
        lib.ganet_conv5_apply.restype = c_int
        lib.ganet_conv5_apply.argtypes = [c_int32, c_int32, c_int32, P, P, c_int32, c_int32, P, P]
        lib.ganet_conv5_wgrad_workspace.restype = c_size_t
        lib.ganet_conv5_wgrad_workspace.argtypes = [c_int32, c_int32, c_int32]
        lib.ganet_conv5_wgrad.restype = c_int
        lib.ganet_conv5_wgrad.argtypes = [c_int32, c_int32, c_int32, P, P, P, P, c_size_t, P]
        lib.ganet_last_error.restype = c_char_p
        lib.ganet_abi_version.restype = c_int
        if lib.ganet_abi_version() != 9:
            raise RuntimeError("libganet_hip.so ABI version mismatch; rebuild")
        _ganet = lib
    return _ganet


def raw_stream(device) -> int:
    """torch's current HIP stream of `device` as an integer handle. (torch.cuda.current_stream(...).cuda_stream builds a
    Stream object per call: ~7 us, 13 times per training iteration — this is the C accessor underneath.)"""
    if not isinstance(device, torch.device):       # ("cuda" / "cuda:1" / an ordinal: str.index is a method, not None)
        device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    idx = device.index
    if idx is None:
        idx = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


def ganet_check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("ganet: " + ganet().ganet_last_error().decode())


def gsr_check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("gsr: " + gsr().gsr_last_error().decode())


def galbs_check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("galbs: " + galbs().galbs_last_error().decode())
