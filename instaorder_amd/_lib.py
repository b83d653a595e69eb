"""ctypes binding of libinstaorder_hip.so (the C ABI declared in include/instaorder_hip.h).

There is NO fallback: if the shared library is missing, or an op is invoked without a
gfx950 device, a RuntimeError is raised.  Build with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C instaorder_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# IO_LIB_PATH: a tuning build of the same library (csrc/Makefile VARIANT=...), for same-box A/B runs of kernel variants
LIB_PATH = os.environ.get("IO_LIB_PATH") or os.path.join(_HERE, "libinstaorder_hip.so")
_lib = None

c_float_p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())


def csrc_digest():
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, the public header), in name order: what a measurement of a
    kernel (a PMC profile under profiles/) was taken on.  bench.py quotes such a measurement only while this matches."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "instaorder_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


class TensorInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("kind", C.c_int), ("ndim", C.c_int), ("shape", C.c_long * 4),
                ("offset", C.c_long), ("numel_storage", C.c_long), ("cin_storage", C.c_int),
                ("bn_index", C.c_int), ("running_offset", C.c_long)]


class PairDesc(C.Structure):
    _fields_ = [("image_off", C.c_int64), ("mask1_off", C.c_int64), ("mask2_off", C.c_int64), ("H", C.c_int32),
                ("W", C.c_int32), ("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32),
                ("flip", C.c_int32), ("interp", C.c_int32)]


class DgradFused(C.Structure):
    """io_dgrad_fused of include/instaorder_hip.h (device pointers as integers, None = NULL)"""
    _fields_ = [(n, C.c_void_p) for n in ("xb_y", "xb_coef", "xb_dy_out", "add", "relu_mask", "ep_y", "ep_mean", "ep_rstd",
                                          "ep_scale", "ep_shift", "ep_act_out", "ep_p1", "ep_p2", "wino_scratch")] + \
               [("wino_scratch_floats", C.c_size_t), ("relu_maskbits", C.c_void_p)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_long), ("total_ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double), ("flops_executed", C.c_double)]


_P, _I, _L, _F, _Z = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol include/instaorder_hip.h declares
SIGNATURES = {
    "io_abi_version": (_I, []),
    "io_last_error_string": (C.c_char_p, []),
    "io_device_count": (_I, []),
    "io_set_winograd": (_I, [_I]),
    "io_get_winograd": (_I, []),
    "io_set_bf16_p256": (_I, [_I]),
    "io_get_bf16_p256": (_I, []),
    "io_set_bf16_p256_xop": (_I, [_I]),
    "io_get_bf16_p256_xop": (_I, []),
    "io_conv2d_fwd_resid_dt": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P]),
    "io_debug_last_nt_route": (_I, []),
    "io_conv2d_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "io_conv2d_bnstats_workspace_floats": (_Z, [_I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "io_conv2d_fwd_bnstats": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P,
                                   _P, _P, _P, _Z, _P]),
    "io_conv2d_dgrad": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "io_conv2d_dgrad_bnbwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P,
                                   _P, _Z, _P]),
    "io_conv2d_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "io_conv2d_wgrad": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "io_filter_transpose": (_I, [_P, _I, _I, _I, _P, _P]),
    "io_bn_partial_floats": (_Z, [_I, _I, _I]),
    "io_bn_stats_finalize": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _Z, _P]),
    "io_bn_eval_prepare": (_I, [_I, _P, _P, _P, _P, _F, _P, _P, _P, _P]),
    "io_bn_apply": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "io_bn_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _P]),
    "io_maxpool_fwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "io_maxpool_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "io_avgpool_fc_fwd": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _P]),
    "io_avgpool_fc_bwd": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "io_pack_planes_nhwc8": (_I, [C.POINTER(C.c_void_p), C.POINTER(C.c_long), _I, _I, _I, _I, _P, _P]),
    "io_pair_planes_u8": (_I, [_P, _Z, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "io_pair_planes_u8_hw": (_I, [_P, _Z, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "io_order_loss": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _F, _F, _F, _P, _P, _P]),
    "io_sgd_momentum": (_I, [_P, _P, _P, _Z, _F, _F, _F, _P]),
    "io_net_create": (_P, [_I, _I, C.POINTER(C.c_int)]),
    "io_net_destroy": (None, [_P]),
    "io_net_param_floats": (_L, [_P]),
    "io_net_running_floats": (_L, [_P]),
    "io_net_num_tensors": (_I, [_P]),
    "io_net_tensor_info": (_I, [_P, _I, C.POINTER(TensorInfo)]),
    "io_net_num_logits": (_I, [_P]),
    "io_net_workspace_bytes": (_Z, [_P, _I, _I, _I]),
    "io_net_workspace_bytes_hw": (_Z, [_P, _I, _I, _I]),
    "io_net_forward_eval_hw": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _Z, _P, _P]),
    "io_net_activation_offset": (_L, [_P, _I, _I, _I]),
    "io_net_forward": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _P, _P]),
    "io_net_backward": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "io_net_backward_num_stages": (_I, [_P]),
    "io_net_backward_stages": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _Z, _I, _I, _P]),
    "io_net_set_dtype": (_I, [_P, _I]),
    "io_net_get_dtype": (_I, [_P]),
    "io_pack_planes_nhwc8_dt": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "io_conv2d_fwd_dt": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "io_conv2d_dgrad_dt": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "io_conv2d_wgrad_dt": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _I, _I, _P]),
    "io_filter_prepare": (_I, [_P, _I, _I, _I, _P, _I, _I, _P]),
    "io_bn_stats_finalize_dt": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _Z, _I, _P]),
    "io_bn_apply_dt": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P]),
    "io_bn_apply_bits_dt": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "io_bn_bwd_dt": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _I, _P]),
    "io_gconv_pack": (_I, [_P, _I, _I, _I, _P, _P, _I, _P]),
    "io_gconv_unpack_grad": (_I, [_P, _I, _I, _I, _P, _P]),
    "io_gconv2d_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "io_gconv2d_dgrad": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "io_gconv2d_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "io_gconv2d_wgrad": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _I, _P]),
    "io_upsample2x_bilinear_fwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "io_upsample2x_bilinear_bwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "io_bias_act": (_I, [_P, _P, _I, _I, _I, _P, _I, _P]),
    "io_relu_bwd": (_I, [_P, _P, _Z, _P, _I, _P]),
    "io_add": (_I, [_P, _P, _Z, _P, _I, _P]),
    "io_colsum_partial_floats": (_Z, [_I, _I]),
    "io_colsum": (_I, [_P, _I, _I, _P, _P, _Z, _I, _P]),
    "io_head1_fwd": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _I, _P]),
    "io_head1_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _Z, _I, _P]),
    "io_conv2d_fwd_bnstats_dt": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _F, _P,
                                      _P, _P, _P, _P, _Z, _I, _I, _P]),
    "io_conv2d_fwd_bias_dt": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P]),
    "io_stem_packed_floats": (_Z, [_I]),
    "io_stem_wgrad_exact_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "io_stem_fwd_bnstats_exact": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _Z, _P,
                                       _P]),
    "io_stem_wgrad_exact": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _Z, _P, _P]),
    "io_stem_wgrad_exact_bn": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _Z, _P,
                                    _P]),
    # x8, da, y, dw, N, H, W, G, gamma, mean, rstd, scale, shift, dgamma, dbeta, coef, bn_partial, bn_partial_floats, ws, ws_bytes, stream
    "io_stem_wgrad_bn_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _Z, _P]),
    "io_stem_wgrad_bf16_workspace_bytes": (_Z, []),
    "io_debug_last_wgrad_route": (_I, []),
    "io_conv2d_fwd_resid": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P,
                                 _P, _P, _Z, _P]),
    "io_smooth_loss_workspace_floats": (_Z, [_I, _I, _I]),
    "io_smooth_loss_fwd": (_I, [_P, _P, _I, _I, _I, _F, _P, _P, _P, _Z, _P]),
    "io_smooth_loss_bwd": (_I, [_P, _P, _P, _F, _I, _I, _I, _I, _P, _P]),
    "io_disp_order_workspace_floats": (_Z, [_I, _I, _I]),
    "io_disp_order_count": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _Z, _P]),
    "io_bn_tile_partial_floats": (_Z, [_I, _I, _I]),
    "io_bn_bwd_coefs_dt": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _P]),
    "io_bn_bwd_coefs_from_tile_partials": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "io_conv2d_dgrad_fused_dt": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(DgradFused), _I, _P]),
    "io_conv2d_dgrad_bnbwd_dt": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                      _Z, _I, _I, _P]),
    "io_prof_launches": (_I, [_P, _I]),
    "io_weights_prepare": (_I, [_P, _I, _P, _P, _I, _P]),
    "io_weights_unpack_grads": (_I, [_P, _I, _P, _P, _P]),
    "io_maxpool_fwd_xf_dt": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _I, _P]),
    "io_conv2d_fwd_xf_dt": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _P,
                                 _P, _P, _P, _P, _Z, _I, _P]),
    "io_conv2d_wino_scratch_floats": (_Z, [_I, _I]),
    "io_conv2d_fwd_wino": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _Z,
                                _P, _Z, _P]),
    "io_maxpool_fwd_dt": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _P]),
    "io_maxpool_bwd_dt": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "io_avgpool_fc_fwd_dt": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P]),
    "io_avgpool_fc_bwd_dt": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "io_prof_begin": (_I, []),
    "io_prof_begin_ex": (_I, [_I]),
    "io_prof_end": (_I, [C.POINTER(ProfEntry), _I]),
}


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                "instaorder_amd: %s is missing -- the HIP extension has not been built "
                "(run `make -C instaorder_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def last_error():
    return lib().io_last_error_string().decode()


def check(rc, what):
    if rc != 0:
        raise RuntimeError("instaorder_hip %s failed (%d): %s" % (what, rc, last_error()))


_gpu_ok = None


def require_gpu():
    """Fail loudly unless a gfx950 device is visible to HIP."""
    global _gpu_ok
    if _gpu_ok is None:
        _gpu_ok = lib().io_device_count() > 0
    if not _gpu_ok:
        raise RuntimeError("instaorder_amd: no gfx950 (MI355X) device visible; the HIP path cannot run "
                           "and there is no CPU fallback")
