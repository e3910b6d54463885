"""ctypes binding of libl3d_hip.so (the C ABI declared in include/l3d_hip.h).

The product path has NO fallback: if the shared library is missing, fails to load, or a call
returns a non-zero status, this module raises.  PyTorch is used only for device memory
(`tensor.data_ptr()`), streams (`torch.cuda.current_stream()`) and torch.distributed.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# L3D_LIB_PATH: a library built from the same sources with other -D flags (tools/build_variant_lib.py: A/B runs of bench.py on one box)
LIB_PATH = os.environ.get("L3D_LIB_PATH") or os.path.join(_HERE, "libl3d_hip.so")
_lib = None

_P, _I, _F, _SZ, _L, _D = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_long, C.c_double

# name -> argtypes (restype is int unless listed in _RESTYPE)
SIGNATURES = {
    "l3d_version": [],
    "l3d_status_string": [_I],
    "l3d_last_hip_error": [],
    "l3d_knn_graph": [_P, _I, _I, _I, _P, _P],
    "l3d_knn_graph_variant": [_P, _I, _I, _I, _P, _I, _P],
    "l3d_knn_feature_workspace_bytes": [_I, _I, _I],
    "l3d_knn_feature": [_P, _I, _I, _I, _I, _P, _P, _P],
    "l3d_lpfa_group": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "l3d_graph_feature": [_P, _P, _I, _I, _I, _I, _P, _P],
    "l3d_chamfer_forward": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P],
    "l3d_chamfer_forward_variant": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P],
    "l3d_chamfer_backward": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "l3d_chamfer_backward_variant": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P],
    "l3d_chamfer_partials": [_P, _P, _I, _I, _I, _P, _P],
    "l3d_chamfer_combine": [_P, _I, _P, _P],
    "l3d_chamfer_loss_local_ws_bytes": [],
    "l3d_chamfer_loss_local_mb": [_P, _P, _I, _I, _I, _P, _P, _P, _P],
    "l3d_chamfer_forward_loss_ws_bytes": [_I, _I, _I],
    "l3d_chamfer_forward_loss": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "l3d_ball_query": [_I, _I, _I, _F, _I, _P, _P, _P, _P, _P],
    "l3d_group_points": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "l3d_group_points_grad": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "l3d_group_concat": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "l3d_group_concat2": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "l3d_group_first_layer": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "l3d_absmax4_partials": [_P, _SZ, _P, _SZ, _P, _SZ, _P, _SZ, _P, _P],
    "l3d_group_first_layer_planes_auto": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _F, _F, _P, _P, _P],
    "l3d_scatter_add_det_workspace_bytes": [_I, _I, _I],
    "l3d_scatter_add_det": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "l3d_edge_gather_max": [_P, _P, _I, _I, _I, _I, _I, _P, _L, _P],
    "l3d_gather_points": [_I, _I, _I, _I, _P, _P, _P, _P],
    "l3d_gather_points_grad": [_I, _I, _I, _I, _P, _P, _P, _P],
    "l3d_furthest_point_sampling": [_I, _I, _I, _P, _P, _P, _P],
    "l3d_knn": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "l3d_knn_variant": [_I, _I, _I, _I, _P, _P, _P, _P, _I, _P],
    "l3d_three_nn": [_I, _I, _I, _P, _P, _P, _P, _P],
    "l3d_three_interpolate": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "l3d_three_interpolate_concat": [_I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P],
    "l3d_three_interpolate_grad": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "l3d_square_distance": [_P, _P, _I, _I, _I, _I, _P, _P],
    "l3d_gaussian_density": [_P, _I, _I, _F, _P, _P],
    "l3d_query_ball_point": [_F, _I, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "l3d_index_points": [_P, _P, _I, _I, _I, _I, _P, _P],
    "l3d_farthest_point_sample": [_P, _I, _I, _I, _P, _P, _P, _P],
    "l3d_knn_point": [_I, _P, _P, _I, _I, _I, _P, _P, _P],
    "l3d_knn_point_expanded": [_I, _P, _P, _I, _I, _I, _P, _P],
    "l3d_kabsch": [_P, _P, _I, _I, _P, _P, _P, _P],
    "l3d_svd3x3_rotation": [_P, _I, _P, _P],
    "l3d_soft_correspondence_workspace_floats": [_I, _I, _I],
    "l3d_layernorm_backward_workspace_floats": [C.c_long, _I],
    "l3d_layernorm_ref_backward": [_P, _P, _P, _F, C.c_long, _I, _P, _P, _P, _P, _P],
    "l3d_soft_correspondence": [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P],
    "l3d_attention_forward_strided": [_P, _P, _P, _I, _I, _I, _I, _I, _L, _L, _L, _F, _P, _P],
    "l3d_attention_forward_f16b": [_P, _P, _P, _I, _I, _I, _I, _I, _L, _L, _L, _F, _P, _I, _P, _P, _P],
    "l3d_sa_mlp3_fused": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "l3d_bmm_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _I, _P, _P],
    "l3d_softmax_rows": [_P, _P, _L, _I, _F, _P, _P],
    "l3d_split_f16_operand": [_P, _L, _I, _L, _I, _P, _P, _P],
    "l3d_colsum_rows_workspace_bytes": [_L, _I],
    "l3d_colsum_rows": [_P, _L, _I, _L, _P, _P, _P],
    "l3d_layernorm_planes": [_P, _P, _P, _F, _L, _I, _P, _P, _P],
    "l3d_add_transposed": [_P, _P, _I, _I, _I, _P, _P],
    "l3d_max_last": [_P, _L, _I, _P, _P, _P],
    "l3d_bn_finalize": [_P, _I, _I, _D, _P, _P, _P, _D, _I, _D, _P, _P, _P, _P, _P, _P, _P, _P],
    "l3d_bn_backward_finalize": [_P, _I, _P, _I, _I, _D, _I, _P, _P, _P, _P, _P, _P, _P],
    "l3d_max_last_backward": [_P, _P, _L, _I, _P, _P],
    "l3d_linear_rows": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "l3d_layernorm_planes_cf": [_P, _P, _P, _F, _I, _I, _I, _P, _P, _I, _P],
    "l3d_edgeconv_packed_floats": [_I, _I, _I, _I],
    "l3d_edgeconv_pack": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "l3d_edgeconv_forward": [_P, _P, _I, _I, _I, _P, _I, _I, _I, _I, _P, _P],
    "l3d_edgeconv_forward_split": [_P, _P, _I, _I, _I, _P, _P, _P],
    "l3d_edgeconv_forward_f16b": [_P, _P, _I, _I, _I, _P, _P, _I, _P, _P],
    "l3d_edgeconv_packed_v2_flag_index": [],
    "l3d_pointwise_conv": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "l3d_split_bytes": [_I, _I],
    "l3d_split_rows": [_P, _I, _I, _P, _P],
    "l3d_pointwise_conv_split": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "l3d_f16_image_bytes": [_I, _L, _I],
    "l3d_conv_f16_split_weights": [_P, _I, _I, _P, _P],
    "l3d_split_f16_rows": [_P, _L, _I, _I, _I, _P, _P, _P],
    "l3d_pointwise_conv_f16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P],
    "l3d_first_layer_f16_planes": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "l3d_fold_mlp": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P],
    "l3d_fold_mlp_f16": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P],
    "l3d_channel_stats": [_P, _I, _I, _L, _P, _P],
    "l3d_bn_act_forward": [_P, _P, _P, _I, _I, _L, _I, _P, _P],
    "l3d_bn_backward_stats": [_P, _P, _P, _P, _P, _P, _I, _I, _L, _I, _P, _P, _P, _I, _P],
    "l3d_bn_act_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _I, _P, _P, _P, _I, _P],
    "l3d_sum_clouds_f64": [_P, _I, _L, _P, _P],
    "l3d_wgrad_workspace_bytes": [_I, _I, _I, _L, _I],
    "l3d_wgrad": [_P, _P, _I, _I, _I, _L, _I, _P, _P, _P],
    "l3d_uniform_clouds": [C.c_ulonglong, _I, _I, _F, _F, _P, _P],
    "l3d_euler_transform": [_P, _P, _P, _I, _I, _P, _P, _P],
    "l3d_twist_transform": [_P, _P, _I, _I, _P, _P, _P, _P],
    "l3d_quat_transform": [_P, _P, _I, _I, _P, _P],
    "l3d_sceneflow_batch": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "l3d_emd_workspace_bytes": [_I, _I, _I],
    "l3d_probe_mfma_sustained": [_I, _P, _P, _P],
    "l3d_emd_forward": [_P, _P, _I, _I, _I, _P, _P, _P, _I, _P],
    "l3d_emd_backward": [_P, _P, _P, _I, _I, _I, _P, _P, _P],
}
_RESTYPE = {"l3d_status_string": C.c_char_p, "l3d_edgeconv_packed_floats": _SZ, "l3d_split_bytes": _SZ,
            "l3d_soft_correspondence_workspace_floats": _SZ, "l3d_layernorm_backward_workspace_floats": _SZ, "l3d_knn_feature_workspace_bytes": _SZ,
            "l3d_scatter_add_det_workspace_bytes": _SZ, "l3d_chamfer_loss_local_ws_bytes": _SZ, "l3d_chamfer_forward_loss_ws_bytes": _SZ, "l3d_f16_image_bytes": _SZ,
            "l3d_wgrad_workspace_bytes": _SZ, "l3d_emd_workspace_bytes": _SZ, "l3d_colsum_rows_workspace_bytes": _SZ}


class L3DError(RuntimeError):
    pass


def lib():
    """Load libl3d_hip.so once.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise L3DError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -m learning3d_amd.build, or __graft_entry__.build()). "
                "learning3d_amd has no CPU / eager fallback by design.")
        handle = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the ABI drifted
            fn.argtypes = argtypes
            fn.restype = _RESTYPE.get(name, _I)
        _lib = handle
    return _lib


LAUNCH_LOG = None      # set to a list to record the name of every C-ABI call that passes through check() (tests: which route ran)
FAILED_CALLS = 0       # number of C-ABI calls that returned a non-zero status (holders of cross-call device state re-arm on a change)


def check(status, what):
    if LAUNCH_LOG is not None:
        LAUNCH_LOG.append(what)
    if status != 0:
        global FAILED_CALLS
        FAILED_CALLS += 1
        l = lib()
        msg = l.l3d_status_string(status).decode()
        raise L3DError(f"{what}: {msg} (status {status}, hipError {l.l3d_last_hip_error()})")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def require_gpu(*tensors):
    """Device tensors only, all on ONE device, which must be the current device: the C ABI launches on the current
    device's current stream (stream_ptr), so a tensor living elsewhere would be read through a foreign pointer.  Raises
    instead (callers switch with `with torch.cuda.device(t.device):`)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise L3DError("learning3d_amd operates on MI355X device tensors only "
                           "(got a CPU tensor; there is no CPU fallback)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise L3DError(f"tensors on different devices in one call ({dev} and {t.device})")
    if dev is not None and dev.index != torch.cuda.current_device():
        raise L3DError(f"tensors live on {dev} but the current device is cuda:{torch.cuda.current_device()}; "
                       f"wrap the call in `with torch.cuda.device({dev.index}):`")


class on_device_of:
    """`with on_device_of(t, ...):` -- make the first device tensor's GPU the current device for the calls inside (the C ABI
    launches on the current device's current stream).  A no-op when it already is, or when no tensor is a device tensor."""

    def __init__(self, *tensors):
        self.dev = next((t.device for t in tensors if isinstance(t, torch.Tensor) and t.is_cuda), None)
        self.ctx = None

    def __enter__(self):
        if self.dev is not None and self.dev.index != torch.cuda.current_device():
            self.ctx = torch.cuda.device(self.dev)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def f32c(t):
    """contiguous fp32 view/copy of a device tensor"""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
