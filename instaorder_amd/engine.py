"""Thin object layer over the C ABI: the ResNet-50 plan (`Net`) and the stand-alone ops.

Tensors are torch tensors used purely as device memory; every call is enqueued on torch's
current HIP stream and returns without synchronising.
"""
import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_f32(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous float32 tensor on the GPU" % name)
    return t


class Net(object):
    """Plan of resnet50_cls(in_channels, num_classes) -- mirrors models/backbone/resnet_cls.py:259-268.
    Creating it needs no GPU (it only lays out the flat parameter buffer)."""

    DTYPES = {"fp32": 0, "f32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}

    def __init__(self, in_channels=5, num_classes=2, dtype="fp32"):
        self.lib = _lib.lib()
        if dtype not in self.DTYPES:
            raise ValueError("dtype must be 'fp32' or 'bf16', got %r" % (dtype,))
        self.dtype = "bf16" if self.DTYPES[dtype] else "fp32"
        heads = list(num_classes) if isinstance(num_classes, (list, tuple)) else [int(num_classes)]
        self.head_dims = heads
        arr = (C.c_int * len(heads))(*heads)
        self.handle = self.lib.io_net_create(int(in_channels), len(heads), arr)
        if not self.handle:
            raise RuntimeError("io_net_create failed: " + _lib.last_error())
        _lib.check(self.lib.io_net_set_dtype(self.handle, self.DTYPES[dtype]), "io_net_set_dtype")
        self.in_channels = int(in_channels)
        self.param_floats = int(self.lib.io_net_param_floats(self.handle))
        self.running_floats = int(self.lib.io_net_running_floats(self.handle))
        self.num_logits = int(self.lib.io_net_num_logits(self.handle))
        self.tensors = []
        info = _lib.TensorInfo()
        for i in range(self.lib.io_net_num_tensors(self.handle)):
            _lib.check(self.lib.io_net_tensor_info(self.handle, i, C.byref(info)), "io_net_tensor_info")
            self.tensors.append(dict(name=info.name.decode(), kind=info.kind,
                                     shape=tuple(info.shape[k] for k in range(info.ndim)),
                                     offset=int(info.offset), numel_storage=int(info.numel_storage),
                                     cin_storage=int(info.cin_storage), bn_index=int(info.bn_index),
                                     running_offset=int(info.running_offset)))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.io_net_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def workspace_bytes(self, N, S, training):
        n = int(self.lib.io_net_workspace_bytes(self.handle, int(N), int(S), int(bool(training))))
        if n == 0:
            raise RuntimeError("io_net_workspace_bytes: " + _lib.last_error())
        return n

    def activation_offset(self, N, S, which):
        off = int(self.lib.io_net_activation_offset(self.handle, int(N), int(S), int(which)))
        if off < 0:
            raise RuntimeError("io_net_activation_offset: " + _lib.last_error())
        return off

    def forward(self, params, running, x8, N, S, G, training, ws, logits):
        _lib.require_gpu()
        _lib.check(self.lib.io_net_forward(self.handle, _ptr(params), _ptr(running), _ptr(x8), int(N), int(S),
                                           int(G), int(bool(training)), _ptr(ws), ws.numel() * ws.element_size(),
                                           _ptr(logits), _stream()), "io_net_forward")

    def workspace_bytes_hw(self, N, H, W):
        n = int(self.lib.io_net_workspace_bytes_hw(self.handle, int(N), int(H), int(W)))
        if n == 0:
            raise RuntimeError("io_net_workspace_bytes_hw: " + _lib.last_error())
        return n

    def forward_eval_hw(self, params, running, x8, N, H, W, ws, logits):
        """inference forward on H x W inputs (multiples of 32): the 'orig' mode of the reference's inference.py"""
        _lib.require_gpu()
        _lib.check(self.lib.io_net_forward_eval_hw(self.handle, _ptr(params), _ptr(running), _ptr(x8), int(N), int(H),
                                                   int(W), _ptr(ws), ws.numel() * ws.element_size(), _ptr(logits),
                                                   _stream()), "io_net_forward_eval_hw")

    def backward(self, params, grads, x8, dlogits, N, S, G, ws, stages=None):
        """stages = (lo, hi): only the backward stages [lo, hi) of io_net_backward_stages (0 = heads + layer4 .. 3 =
        layer1 + stem); None = the whole pass"""
        _lib.require_gpu()
        if stages is None:
            _lib.check(self.lib.io_net_backward(self.handle, _ptr(params), _ptr(grads), _ptr(x8), _ptr(dlogits),
                                                int(N), int(S), int(G), _ptr(ws), ws.numel() * ws.element_size(),
                                                _stream()), "io_net_backward")
        else:
            _lib.check(self.lib.io_net_backward_stages(self.handle, _ptr(params), _ptr(grads), _ptr(x8), _ptr(dlogits),
                                                       int(N), int(S), int(G), _ptr(ws), ws.numel() * ws.element_size(),
                                                       int(stages[0]), int(stages[1]), _stream()),
                       "io_net_backward_stages")

    @property
    def backward_stages(self):
        return int(self.lib.io_net_backward_num_stages(self.handle))


TORCH_DTYPE = {"fp32": torch.float32, "bf16": torch.bfloat16}


def pack_planes(planes, strides, N, H, W, out):
    """planes: list of (fp32 tensor, element offset) giving channel 0.. of sample 0; strides: floats between
    samples.  Writes out[N,H,W,8] (torch.cat + NCHW->NHWC + pad, supervised_order.py:537); ``out`` is fp32 or
    bf16 -- the storage type of the network that will read it."""
    _lib.require_gpu()
    n = len(planes)
    pa = (C.c_void_p * n)(*[t.data_ptr() + 4 * off for t, off in planes])
    sa = (C.c_long * n)(*[int(s) for s in strides])
    if out.dtype not in (torch.float32, torch.bfloat16) or not out.is_contiguous():
        raise ValueError("pack_planes: out must be a contiguous fp32 or bf16 tensor")
    _lib.check(_lib.lib().io_pack_planes_nhwc8_dt(pa, sa, n, int(N), int(H), int(W), _ptr(out),
                                                  1 if out.dtype == torch.bfloat16 else 0, _stream()),
               "io_pack_planes_nhwc8_dt")


def pack_nchw(x, out=None, dtype="fp32"):
    """x[N,C<=5,H,W] (NCHW fp32) -> [N,H,W,8] of ``dtype``."""
    _dev_f32(x, "x")
    N, Cc, H, W = x.shape
    if out is None:
        out = torch.empty((N, H, W, 8), device=x.device, dtype=TORCH_DTYPE[dtype])
    pack_planes([(x, c * H * W) for c in range(Cc)], [Cc * H * W] * Cc, N, H, W, out)
    return out


def pack_pair_directions(rgb, modal1, modal2, out=None, dtype="fp32"):
    """Both mask orders of a pair batch in one buffer: rows [0,B) = (modal1, modal2, rgb), rows
    [B,2B) = (modal2, modal1, rgb) -- the two model calls of supervised_order.py:537-538.  ``rgb=None``
    is the reference's ``use_rgb=False`` form (masks only)."""
    for t, nme in ((rgb, "rgb"), (modal1, "modal1"), (modal2, "modal2")):
        if t is not None:
            _dev_f32(t, nme)
    B, _, H, W = modal1.shape
    if out is None:
        out = torch.empty((2 * B, H, W, 8), device=modal1.device, dtype=TORCH_DTYPE[dtype])
    HW = H * W
    rgbp = [(rgb, c * HW) for c in range(3)] if rgb is not None else []
    st = [HW, HW] + [3 * HW] * len(rgbp)
    pack_planes([(modal1, 0), (modal2, 0)] + rgbp, st, B, H, W, out[:B])
    pack_planes([(modal2, 0), (modal1, 0)] + rgbp, st, B, H, W, out[B:])
    return out


def order_loss(logits, B, Kocc, Kdep, occ_target=None, depth_target=None, is_overlap=None,
               overlap_weight=0.0, distinct_weight=0.0, inv_world=1.0, want_grad=True):
    """losses[3] = (total/world, occlusion, depth) and d(total/world)/dlogits."""
    _lib.require_gpu()
    N = logits.shape[0]
    losses = torch.empty(3, device=logits.device, dtype=torch.float32)
    dlogits = torch.empty_like(logits) if want_grad else None
    _lib.check(_lib.lib().io_order_loss(_ptr(logits), int(N), int(B), int(Kocc), int(Kdep), _ptr(occ_target),
                                        _ptr(depth_target), _ptr(is_overlap), float(overlap_weight),
                                        float(distinct_weight), float(inv_world), _ptr(losses), _ptr(dlogits),
                                        _stream()), "io_order_loss")
    return losses, dlogits


def sgd_momentum(params, grads, buf, lr, momentum, weight_decay):
    _lib.require_gpu()
    _lib.check(_lib.lib().io_sgd_momentum(_ptr(params), _ptr(grads), _ptr(buf), params.numel(), float(lr),
                                          float(momentum), float(weight_decay), _stream()), "io_sgd_momentum")


_prof_on = False


def set_winograd(on):
    """Product form of the fp32 3x3 stride-1 layers: True = Winograd row forms where the shape allows (default; env
    IO_WINOGRAD=0 starts a process with them off), False = the direct implicit GEMM everywhere.  Returns the previous
    setting.  A captured hipGraph keeps the form it was captured with."""
    return bool(_lib.lib().io_set_winograd(1 if on else 0))


def get_winograd():
    return bool(_lib.lib().io_get_winograd())


def prof_active():
    return _prof_on


def prof_begin(share_events=True):
    """Start HIP-event timing of every library launch (per kernel class, on the launch stream).  share_events=False:
    own start event per launch group -- for op-by-op graphs (instaorder_amd.ops), where torch kernels and host gaps sit
    between the library's launches and must not be charged to them."""
    global _prof_on
    _lib.check(_lib.lib().io_prof_begin_ex(1 if share_events else 0), "io_prof_begin_ex")
    _prof_on = True


def prof_launches(max_entries=8192):
    """Per-launch records of the running profile, in launch order: [(class, ms, flops, bytes)] (synchronises)."""
    arr = (_lib.ProfEntry * max_entries)()
    n = _lib.lib().io_prof_launches(arr, max_entries)
    return [(arr[i].name.decode(), float(arr[i].total_ms), float(arr[i].flops), float(arr[i].bytes)) for i in range(n)]


def prof_end():
    """Stop and return {class name: dict(launches, total_ms, flops, bytes, flops_executed)} (synchronises); flops = the
    algorithmic count of the direct convolution, flops_executed = what the matrix pipes multiplied (Winograd forms: less)."""
    global _prof_on
    _prof_on = False
    arr = (_lib.ProfEntry * 32)()
    n = _lib.lib().io_prof_end(arr, 32)
    return {arr[i].name.decode(): dict(launches=int(arr[i].launches), total_ms=float(arr[i].total_ms),
                                       flops=float(arr[i].flops), bytes=float(arr[i].bytes),
                                       flops_executed=float(arr[i].flops_executed)) for i in range(n)}
