"""ctypes binding of librpnet_hip.so (C ABI: include/rpnet_abi.h).

This is the whole FFI surface: plain pointers, ints and a stream.  torch is used only
to own device memory and to give the current HIP stream.  There is NO fallback: if the
library is missing or a call fails, a RuntimeError is raised (the product path never
routes through the CPU oracle).
"""
import ctypes as C
import os

import torch

_LIB_PATH = os.environ.get("RPNET_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "librpnet_hip.so")
_lib = None

vp, ci, cf, cs, cd = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_double


class ConvDesc(C.Structure):
    """struct rpnet_conv_desc"""
    _fields_ = [("x0", vp), ("x1", vp), ("C0", ci), ("C1", ci), ("w", vp), ("bias", vp), ("in_scale", vp),
                ("in_scale_mode", ci), ("y0", vp), ("y1", vp), ("Co0", ci), ("Co1", ci), ("ep_scale", vp),
                ("ep_shift", vp), ("ep_relu", ci), ("out_scale", vp), ("out_scale_mode", ci), ("accumulate", ci),
                ("N", ci), ("H", ci), ("W", ci), ("taps", ci), ("upsample", ci), ("groups", ci), ("dilation", ci), ("stats_partial", vp), ("split_planes", ci), ("y_split", vp), ("split_out_planes", ci),
                ("acc_scale_col", vp), ("acc_scale_x", vp), ("acc_scale_dy", vp), ("bnb_y", vp), ("bnb_stats", vp), ("bnb_partial", vp), ("bnb_pmax", vp), ("bnb_groups", ci),
                ("acc_scale_x1", vp), ("out_absmax", vp), ("tune", ci), ("y_split_scale", vp), ("splitk_ws", vp), ("splitk_ws_bytes", cs),
                ("skip_mask", vp), ("skip_mode", ci), ("skip_halo", ci), ("skip_ws", vp), ("tile_skip", vp),
                ("y_enc", vp), ("y_enc_stride", ci)]


PACK_MAX = 24     # layers per rpnet_pack_conv_weights_split call


class PackItem(C.Structure):
    """struct rpnet_pack_item"""
    _fields_ = [("w", vp), ("wp", vp), ("wd", vp), ("row_scale_wp", vp), ("row_scale_wd", vp), ("cout", ci), ("cin", ci),
                ("taps", ci), ("cin_off0", ci), ("cin_split", ci), ("cin_off1", ci), ("cin_pad", ci)]


_SIGS = {
    "rpnet_version": (ci, []),
    "rpnet_last_error_string": (C.c_char_p, []),
    "rpnet_pack_conv_weight": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]),
    "rpnet_split_bf16": (ci, [vp, vp, ci, vp, cs, ci, ci, vp]),
    "rpnet_split_f16": (ci, [vp, vp, ci, vp, vp, vp, vp, cs, ci, ci, ci, vp]),
    "rpnet_pack_conv_weight_split": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp]),
    "rpnet_pack_conv_weights_split": (ci, [C.POINTER(PackItem), ci, ci, vp]),
    "rpnet_conv_fwd": (ci, [C.POINTER(ConvDesc), vp]),
    "rpnet_upconv_collapse_weights": (ci, [vp, vp, ci, ci, vp]),
    "rpnet_conv_up4_supported": (ci, [C.POINTER(ConvDesc), ci]),
    "rpnet_conv_up4_stats_blocks": (ci, [C.POINTER(ConvDesc)]),
    "rpnet_conv_up4": (ci, [C.POINTER(ConvDesc), ci, vp]),
    "rpnet_conv_wgrad_up4_supported": (ci, [C.POINTER(ConvDesc)]),
    "rpnet_conv_wgrad_up4_workspace_bytes": (cs, [ci, ci, ci, ci, ci]),
    "rpnet_conv_wgrad_up4": (ci, [C.POINTER(ConvDesc), vp, vp, vp, cs, vp]),
    "rpnet_conv_stats_blocks": (ci, [C.POINTER(ConvDesc)]),
    "rpnet_conv_tile_variant": (ci, [C.POINTER(ConvDesc)]),
    "rpnet_conv_splitk_workspace_bytes": (cs, [C.POINTER(ConvDesc)]),
    "rpnet_bn_stats_from_partial": (ci, [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp, vp, vp]),
    "rpnet_conv_wgrad_workspace_bytes": (cs, [ci, ci, ci, ci, ci, ci]),
    "rpnet_conv_wgrad": (ci, [C.POINTER(ConvDesc), vp, vp, ci, ci, ci, ci, vp, cs, vp]),
    "rpnet_conv1_fwd": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, ci, vp]),
    "rpnet_conv1_stats_blocks": (ci, [ci, ci, ci, ci, ci]),
    "rpnet_pow2_scale": (ci, [vp, vp, vp]),
    "rpnet_predict_scales": (ci, [vp, vp, vp, ci, cf, ci, vp, vp]),
    "rpnet_conv1_wgrad_workspace_bytes": (cs, [ci, ci, ci, ci]),
    "rpnet_conv1_wgrad": (ci, [vp, vp, vp, ci, ci, ci, ci, vp, cs, vp]),
    "rpnet_conv1_wgrad_bn": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, cs, vp, vp, vp]),
    "rpnet_conv1_bn_relu": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "rpnet_conv1_bn_bwd_rows": (ci, [ci, ci, ci, ci, ci]),
    "rpnet_conv1_bn_bwd_partial": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "rpnet_bn_bwd_coef_offset": (cs, [ci, ci]),
    "rpnet_bn_workspace_bytes": (cs, [ci, ci]),
    "rpnet_bn_stats": (ci, [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp, vp, vp, cs, vp]),
    "rpnet_bn_eval_affine": (ci, [vp, vp, vp, vp, cf, vp, vp, ci, vp]),
    "rpnet_bn_relu": (ci, [vp, vp, vp, vp, vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp, ci, vp]),
    "rpnet_bn_act_scale": (ci, [vp, vp, vp, ci, ci, ci, ci, vp]),
    "rpnet_bn_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, ci, ci, vp, cs, vp, ci, vp]),
    "rpnet_bias_relu_bwd_workspace_bytes": (cs, [ci]),
    "rpnet_bias_relu_bwd": (ci, [vp, vp, vp, vp, cs, ci, vp, cs, vp]),
    "rpnet_maxpool3_fwd": (ci, [vp, vp, ci, ci, ci, ci, ci, vp]),
    "rpnet_maxpool3_bwd": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "rpnet_maxpool2_fwd": (ci, [vp, vp, ci, ci, ci, ci, vp]),
    "rpnet_maxpool2_bwd": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "rpnet_upsample2_bwd": (ci, [vp, vp, ci, ci, ci, ci, vp]),
    "rpnet_mask_avgpool": (ci, [vp, vp, ci, ci, ci, ci, vp]),
    "rpnet_local_corr_fwd": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "rpnet_local_corr_bwd_workspace_bytes": (cs, [ci, ci, ci, ci]),
    "rpnet_local_corr_bwd": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, cs, vp]),
    "rpnet_local_corr_split_fwd": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]),
    "rpnet_local_corr_split_bwd": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, cs, vp]),
    "rpnet_affine_register": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cd, cd, cd, cd, vp]),
    "rpnet_sum_n": (ci, [vp, ci, vp, cs, vp]),
    "rpnet_demons_workspace_bytes": (cs, [ci, ci, ci]),
    "rpnet_demons_register": (ci, [vp, vp, vp, ci, vp, vp, vp, ci, ci, ci, ci, cd, cd, cd, cd, vp, cs, vp]),
    "rpnet_displacement_warp": (ci, [vp, vp, vp, ci, ci, ci, cf, cf, cf, vp]),
    "rpnet_affine_warp": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, cf, cf, cf, vp]),
    "rpnet_identity_grid_warp": (ci, [vp, vp, ci, ci, ci, cf, cf, cf, vp]),
    "rpnet_mask_adjoint": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "rpnet_masked_pool_workspace_bytes": (cs, [ci, ci, ci, ci]),
    "rpnet_masked_pool_fwd": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp, cs, vp]),
    "rpnet_masked_pool_bwd": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "rpnet_cosine_match_fwd": (ci, [vp, vp, vp, ci, ci, ci, ci, cf, vp]),
    "rpnet_cosine_match_bwd_workspace_bytes": (cs, [ci, ci, ci, ci]),
    "rpnet_cosine_match_bwd": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, ci, vp, cs, vp]),
    "rpnet_bilinear_up_fwd": (ci, [vp, vp, ci, ci, ci, ci, ci, vp]),
    "rpnet_bilinear_up_bwd": (ci, [vp, vp, ci, ci, ci, ci, ci, vp]),
    "rpnet_softmax_thresh_pool": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "rpnet_refine_glue_supported": (ci, [ci, ci, ci, ci, ci, ci]),
    "rpnet_refine_glue_fwd": (ci, [vp, vp, vp, vp, cf, vp, vp, vp, vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]),
    "rpnet_refine_glue_bwd_workspace_bytes": (cs, [ci, ci, ci, ci, ci]),
    "rpnet_refine_glue_bwd": (ci, [vp, vp, vp, cf, vp, vp, ci, ci, ci, ci, ci, vp, cs, vp]),
    "rpnet_rowdot_scale": (ci, [vp, vp, vp, vp, vp, cs, ci, ci, ci, vp]),
    "rpnet_softmax_pool_bwd": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "rpnet_loss_workspace_bytes": (cs, [ci, ci, ci, ci]),
    "rpnet_dice_ce_fwd": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp, cs, vp]),
    "rpnet_dice_ce_bwd": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, ci, vp]),
    "rpnet_dice_ce_multi_fwd": (ci, [C.POINTER(vp), ci, vp, vp, vp, ci, ci, ci, ci, vp, cs, vp]),
    "rpnet_dice_ce_multi_bwd": (ci, [C.POINTER(vp), C.POINTER(vp), ci, vp, vp, vp, ci, ci, ci, ci, vp]),
    "rpnet_objective_fwd": (ci, [C.POINTER(vp), C.POINTER(cf), ci, vp, vp, cf, vp, vp, ci, ci, ci, ci, vp, cs, vp]),
    "rpnet_objective_bwd": (ci, [C.POINTER(vp), C.POINTER(vp), C.POINTER(cf), ci, vp, vp, vp, vp, cf, ci, ci, ci, ci, vp]),
    "rpnet_argmax_masks": (ci, [vp, vp, vp, vp, ci, ci, ci, vp]),
    "rpnet_align_labels": (ci, [vp, vp, vp, cs, vp]),
    "rpnet_debug_lds_canary": (ci, [ci, ci, C.c_longlong, vp, vp]),
    "rpnet_debug_fastdiv_selftest": (C.c_longlong, [ci]),
    "rpnet_debug_mfma_spin": (ci, [ci, ci, C.c_longlong, vp, vp]),
}
ABI_SYMBOLS = tuple(_SIGS)
ABI_VERSION = 108      # RPNET_ABI_VERSION of include/rpnet_abi.h


def lib_path():
    return _LIB_PATH


def load():
    """dlopen librpnet_hip.so once; loud failure if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} not found: build it with `make -C rpnet_amd/csrc` "
                               "(or __graft_entry__.build()); rpnet_amd has no CPU fallback")
        lib = C.CDLL(_LIB_PATH)
        lib.rpnet_version.restype = ci
        if lib.rpnet_version() != ABI_VERSION:
            raise RuntimeError(f"{_LIB_PATH} has ABI version {lib.rpnet_version()}, this binding was written for {ABI_VERSION} "
                               "(include/rpnet_abi.h RPNET_ABI_VERSION): rebuild it with `make -C rpnet_amd/csrc`")
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def ptr(t):
    return None if t is None else t.data_ptr()


try:        # the raw handle of the current stream without building a torch.cuda.Stream object (10 us -> 0.5 us per C-ABI call)
    _raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:      # a torch build without the private accessors
    _raw_stream = None


def stream():
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke an int-returning entry point on the current stream; raise on failure."""
    lib = load()
    rc = getattr(lib, name)(*args, stream())
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib.rpnet_last_error_string().decode()}")


def query(name, *args):
    return getattr(load(), name)(*args)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("rpnet_amd runs on MI355X only: got a CPU tensor (there is no CPU fallback; "
                               "the CPU restatement lives in oracle/ and is test infrastructure)")
