"""MI355X-native ``resnet50_cls`` -- host-side mirror of models/backbone/resnet_cls.py.

Same constructor surface (``resnet50_cls(in_channels=5, num_classes=2 | 3 | 4 | [2, 3])``), same
``state_dict`` keys / logical shapes (OIHW filters, torchvision-style names), same call contract
(``forward(x[B,5,H,W]) -> [B,C]`` or ``(occ[B,2], depth[B,3])``) as the reference module
(resnet_cls.py:121-222, 259-268) -- but there are no torch.nn layers inside.  All parameters are
strided views into ONE flat fp32 buffer laid out for the HIP kernels (KRSC filters, the 5-channel
stem padded to 8), gradients into a second flat buffer, BN running statistics into a third; the
arithmetic is ``io_net_forward`` / ``io_net_backward`` of libinstaorder_hip.so.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import engine

__all__ = ["ResNet", "resnet50_cls"]


class _Node(nn.Module):
    """Name-space node of the parameter tree (holds Parameters / buffers, no computation)."""

    def forward(self, *a, **k):   # pragma: no cover
        raise RuntimeError("instaorder_amd: sub-modules are parameter holders; call the ResNet itself")


class _WorkspacePool(object):
    """Byte arenas for in-flight training forwards, reused across steps."""

    def __init__(self):
        self.free = []

    def take(self, nbytes, device):
        for i, t in enumerate(self.free):
            if t.numel() >= nbytes and t.device == device:
                return self.free.pop(i)
        self.free = []          # sizes changed: drop stale arenas before allocating a larger one
        return torch.empty(nbytes, dtype=torch.uint8, device=device)

    def give(self, t):
        self.free.append(t)


class _NetFunction(torch.autograd.Function):
    """Autograd bridge: forward = io_net_forward (training), backward = io_net_backward; the
    gradients are produced directly in the module's flat gradient buffer and handed to autograd
    as the per-parameter views."""

    @staticmethod
    def forward(ctx, module, x8, N, S, G, *params):
        logits, ws = module._run_forward(x8, N, S, G, True)
        ctx.module, ctx.x8, ctx.ws, ctx.dims = module, x8, ws, (N, S, G)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        m = ctx.module
        N, S, G = ctx.dims
        m._run_backward(ctx.x8, dlogits.contiguous(), N, S, G, ctx.ws)
        m._pool.give(ctx.ws)
        ctx.ws = None
        grads = tuple(g.clone() for g in m._grad_views)
        return (None, None, None, None, None) + grads


class ResNet(nn.Module):
    def __init__(self, in_channels=5, num_classes=2, device=None, dtype="fp32"):
        """dtype='bf16' (not in the reference, which is fp32 only): activations, their gradients and the GEMM
        operands are bf16 on the device (BASELINE configs[2], [3]); parameters, gradients, BN statistics and the
        module's inputs / outputs stay fp32, so the API and checkpoints are unchanged."""
        super(ResNet, self).__init__()
        self.plan = engine.Net(in_channels, num_classes, dtype)
        self.dtype = self.plan.dtype
        self.in_channels = in_channels
        self.is_occ_and_depth = isinstance(num_classes, (list, tuple))
        self.head_dims = list(num_classes) if self.is_occ_and_depth else [int(num_classes)]
        dev = torch.device(device) if device is not None else torch.device("cpu")
        self._flat = torch.zeros(self.plan.param_floats, dtype=torch.float32, device=dev)
        self._flat_grad = torch.zeros_like(self._flat)
        self._running = torch.zeros(self.plan.running_floats, dtype=torch.float32, device=dev)
        self._nbt = None            # num_batches_tracked of every BN layer, one int64 vector (one add per step)
        self._pool = _WorkspacePool()
        self._eval_ws = None
        self._param_list = []
        self._bn_nodes = []
        self._build_tree()
        self._bind_views()
        self.reset_parameters()

    # ---- parameter tree ----------------------------------------------------------------------
    def _node(self, path):
        cur = self
        for part in path:
            if part not in cur._modules:
                cur.add_module(part, _Node())
            cur = cur._modules[part]
        return cur

    def _build_tree(self):
        for t in self.plan.tensors:
            parts = t["name"].split(".")
            node = self._node(parts[:-1])
            p = nn.Parameter(torch.empty(0), requires_grad=True)
            node.register_parameter(parts[-1], p)
            self._param_list.append((t, p))
            if t["kind"] == 1:     # bn weight: this node also owns the running statistics
                node.register_buffer("running_mean", torch.empty(0))
                node.register_buffer("running_var", torch.empty(0))
                node.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long))
                self._bn_nodes.append((t, node))

    @staticmethod
    def _view(flat, t):
        """Strided view with the reference's logical shape over the kernel-friendly storage."""
        off, shp = t["offset"], t["shape"]
        if t["kind"] == 0:
            O, I, R, S = shp
            cs = t["cin_storage"]
            v = flat[off:off + O * R * S * cs].view(O, R, S, cs)[:, :, :, :I]
            return v.permute(0, 3, 1, 2)          # [O, I, R, S] over KRSC storage
        n = 1
        for s in shp:
            n *= s
        return flat[off:off + n].view(*shp)

    def _bind_views(self):
        self._grad_views = []
        if self._nbt is None:
            self._nbt = torch.zeros(len(self._bn_nodes), dtype=torch.long, device=self._flat.device)
        for t, p in self._param_list:
            p.data = self._view(self._flat, t)
            gv = self._view(self._flat_grad, t)
            self._grad_views.append(gv)
        for i, (t, node) in enumerate(self._bn_nodes):
            C, ro = t["shape"][0], t["running_offset"]
            node._buffers["running_mean"] = self._running[ro:ro + C]
            node._buffers["running_var"] = self._running[ro + C:ro + 2 * C]
            node._buffers["num_batches_tracked"] = self._nbt[i]

    def _apply(self, fn, *args, **kwargs):
        # move / cast the flat buffers, then re-create every view (module.cuda(), .to(), .float())
        new_flat = fn(self._flat)
        if new_flat.dtype != torch.float32:
            raise RuntimeError("instaorder_amd: parameters are stored in fp32 only")
        self._flat = new_flat.contiguous()
        self._flat_grad = fn(self._flat_grad).contiguous()
        self._running = fn(self._running).contiguous()
        self._nbt = fn(self._nbt).contiguous()
        self._bind_views()
        self._eval_ws = None
        self._pool = _WorkspacePool()
        return self

    def reset_parameters(self):
        """Construction-time init of the reference (resnet_cls.py:162-167): He-normal fan-out filters,
        BN weight 1 / bias 0, default nn.Linear init; running_mean 0 / running_var 1."""
        with torch.no_grad():
            for t, p in self._param_list:
                if t["kind"] == 0:
                    nn.init.kaiming_normal_(p, mode="fan_out", nonlinearity="relu")
                elif t["kind"] == 1:
                    p.fill_(1.0)
                elif t["kind"] == 2:
                    p.zero_()
                elif t["kind"] == 3:
                    nn.init.kaiming_uniform_(p, a=5 ** 0.5)
                else:
                    bound = 1.0 / (2048 ** 0.5)
                    nn.init.uniform_(p, -bound, bound)
            for t, node in self._bn_nodes:
                node._buffers["running_mean"].zero_()
                node._buffers["running_var"].fill_(1.0)
            self._nbt.zero_()

    # ---- flat access for the fused optimiser / data-parallel helpers ----------------------------
    @property
    def flat_params(self):
        return self._flat

    @property
    def flat_grads(self):
        return self._flat_grad

    @property
    def flat_running(self):
        return self._running

    def attach_grads(self):
        """Point every ``param.grad`` at its slice of the flat gradient buffer (what the fused step
        produces), so optimisers / all-reduce helpers written against ``param.grad`` see them."""
        for (t, p), gv in zip(self._param_list, self._grad_views):
            p.grad = gv

    def bump_batches_tracked(self, k):
        self._nbt += k

    # ---- execution -------------------------------------------------------------------------------
    def _run_forward(self, x8, N, S, G, training):
        dev = self._flat.device
        if dev.type != "cuda":
            raise RuntimeError("instaorder_amd: the network runs only on an MI355X (module is on %s); "
                               "there is no CPU fallback" % dev)
        nbytes = self.plan.workspace_bytes(N, S, training)
        if training:
            ws = self._pool.take(nbytes, dev)
        else:
            if self._eval_ws is None or self._eval_ws.numel() < nbytes:
                self._eval_ws = None
                self._eval_ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            ws = self._eval_ws
        logits = torch.empty((N, self.plan.num_logits), dtype=torch.float32, device=dev)
        self.plan.forward(self._flat, self._running, x8, N, S, G, training, ws, logits)
        if training:
            self.bump_batches_tracked(G)
        return logits, ws

    def _run_forward_eval_hw(self, x8, N, H, W):
        dev = self._flat.device
        if dev.type != "cuda":
            raise RuntimeError("instaorder_amd: the network runs only on an MI355X (module is on %s); "
                               "there is no CPU fallback" % dev)
        nbytes = self.plan.workspace_bytes_hw(N, H, W)
        if self._eval_ws is None or self._eval_ws.numel() < nbytes:
            self._eval_ws = None
            self._eval_ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        logits = torch.empty((N, self.plan.num_logits), dtype=torch.float32, device=dev)
        self.plan.forward_eval_hw(self._flat, self._running, x8, N, H, W, self._eval_ws, logits)
        return logits

    def _run_backward(self, x8, dlogits, N, S, G, ws, stages=None):
        self.plan.backward(self._flat, self._flat_grad, x8, dlogits, N, S, G, ws, stages)

    def grad_stage_slices(self):
        """[lo, hi) element ranges of the flat gradient buffer, one per backward stage in EXECUTION order (heads +
        layer4, layer3, layer2, layer1 + stem): what becomes final when that stage of the backward pass has run.  The
        parameters lie in creation order (conv1, bn1, layer1 .. layer4, heads), so the ranges are contiguous and tile
        the buffer -- the buckets of the overlapped gradient exchange (distributed_utils.GradientBuckets)."""
        first = {}
        for t in self.plan.tensors:
            stage = t["name"].split(".")[0]
            first.setdefault(stage, t["offset"])
        cuts = [first["layer4"], first["layer3"], first["layer2"], 0]
        his = [self.plan.param_floats] + cuts[:-1]
        return [(lo, hi) for lo, hi in zip(cuts, his)]

    def forward_packed(self, x8, groups=1):
        """x8[N,S,S,8] (already packed NHWC) -> raw logits [N, K].  In training mode the batch is
        normalised in ``groups`` independent BN groups (2 = both mask orders of a pair batch) and the
        result carries autograd history when gradients are enabled."""
        N, S = x8.shape[0], x8.shape[1]
        if x8.dtype != engine.TORCH_DTYPE[self.dtype] or not x8.is_contiguous():
            raise ValueError("packed input must be contiguous %s (the net's storage type), got %s"
                             % (self.dtype, x8.dtype))
        if x8.shape[2] != S:
            # H x W inputs: inference only (the 'orig' mode of inference.py:401-407 feeds whole images at their own
            # aspect ratio; resnet_cls.py:152 pools adaptively, so the network takes them)
            if self.training:
                raise ValueError("training takes square inputs only, got %s" % (tuple(x8.shape),))
            return self._run_forward_eval_hw(x8, N, int(x8.shape[1]), int(x8.shape[2]))
        if self.training:
            if torch.is_grad_enabled():
                params = [p for _, p in self._param_list]
                return _NetFunction.apply(self, x8, N, S, groups, *params)
            logits, ws = self._run_forward(x8, N, S, groups, True)
            self._pool.give(ws)
            return logits
        logits, _ = self._run_forward(x8, N, S, 1, False)
        return logits

    def split_heads(self, logits):
        if self.is_occ_and_depth:
            k = self.head_dims[0]
            return logits[:, :k], logits[:, k:]
        return logits

    def forward(self, x):
        """x[B, in_channels, H, W] NCHW fp32 -> [B,C] or (occ[B,2], depth[B,3]) (resnet_cls.py:203-222)."""
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError("expected [B,%d,H,W], got %s" % (self.in_channels, tuple(x.shape)))
        if not x.is_cuda:
            raise RuntimeError("instaorder_amd: input must be on the GPU; there is no CPU fallback")
        x8 = engine.pack_nchw(x.contiguous().float(), dtype=self.dtype)
        return self.split_heads(self.forward_packed(x8, 1))


def resnet50_cls(pretrained=False, progress=True, **kwargs):
    """Factory with the reference's signature (resnet_cls.py:259-268)."""
    if pretrained:
        raise ValueError("no ImageNet checkpoint is bundled; load one with load_state_dict")
    return ResNet(**kwargs)
