"""Fused momentum-SGD over the network's flat parameter buffer.

Behaves as ``torch.optim.SGD(model.parameters(), lr, momentum=0.9, weight_decay=wd)`` does in the
reference (models/single_stage_model.py:35-38): ONE parameter group (BN affine terms and biases are
decayed too), coupled L2, dampening 0, no Nesterov, momentum buffer = first decayed gradient.  It IS
a ``torch.optim.Optimizer`` (utils/scheduler.py:7-9 type-checks it and rewrites
``param_groups[i]['lr']``), and its ``state_dict`` has torch.optim.SGD's layout (one
``momentum_buffer`` per parameter) so checkpoints interchange with the reference
(single_stage_model.py:66-72).  The update itself is one HIP launch over 23.5 M floats.
"""
import torch

from . import engine


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, module, lr, momentum=0.9, weight_decay=0.0):
        net = getattr(module, "module", module)     # DistModule / FixModule wrap the ResNet
        if not hasattr(net, "flat_params"):
            raise TypeError("FusedSGD needs an instaorder_amd ResNet (flat parameter buffer)")
        self._net = net
        defaults = dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=False)
        super(FusedSGD, self).__init__(list(net.parameters()), defaults)
        if len(self.param_groups) != 1:
            raise ValueError("FusedSGD supports exactly one parameter group")
        self._buf = None
        self._views = None

    def _ensure_buf(self):
        flat = self._net.flat_params
        if self._buf is None or self._buf.device != flat.device:
            old = self._buf
            self._buf = torch.zeros_like(flat)
            if old is not None:
                self._buf.copy_(old)
            self._views = [self._net._view(self._buf, t) for t, _ in self._net._param_list]
        return self._buf

    def zero_grad(self, set_to_none=True):
        # gradients are rewritten in full by every backward of the engine; nothing to clear
        for p in self.param_groups[0]["params"]:
            if set_to_none:
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        if g.get("nesterov") or g.get("dampening", 0) != 0:
            raise ValueError("FusedSGD: nesterov / dampening are not supported")
        buf = self._ensure_buf()
        engine.sgd_momentum(self._net.flat_params, self._net.flat_grads, buf, g["lr"], g["momentum"],
                            g["weight_decay"])
        return None

    # ---- checkpoint interchange with torch.optim.SGD ------------------------------------------------
    def state_dict(self):
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        n = len(self.param_groups[0]["params"])
        group["params"] = list(range(n))
        state = {}
        if self._buf is not None:
            for i, v in enumerate(self._views):
                state[i] = {"momentum_buffer": v.detach().clone().contiguous()}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        grp = sd["param_groups"][0]
        for k, v in grp.items():
            if k != "params":
                self.param_groups[0][k] = v
        if sd["state"]:
            self._ensure_buf()
            with torch.no_grad():
                self._buf.zero_()
                for i, st in sd["state"].items():
                    mb = st.get("momentum_buffer")
                    if mb is not None:
                        self._views[int(i)].copy_(mb)


class FlatSGD(torch.optim.Optimizer):
    """Momentum-SGD for an ordinary module tree (the MiDaS branch): the parameters are moved into ONE flat fp32
    buffer (each ``nn.Parameter`` becomes a view of it), so the update is the same single HIP launch as FusedSGD;
    the gradients autograd produced per tensor are gathered into a flat buffer first (that buffer is also what the
    data-parallel all-reduce runs on).  torch.optim.SGD semantics / state_dict layout as above."""

    def __init__(self, module, lr, momentum=0.9, weight_decay=0.0):
        params = []
        seen = set()
        for p in module.parameters():
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                params.append(p)
        defaults = dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=False)
        super(FlatSGD, self).__init__(params, defaults)
        self._params = params
        n = sum(((p.numel() + 63) // 64) * 64 for p in params)
        dev = params[0].device
        self.flat_params = torch.zeros(n, device=dev)
        self.flat_grads = torch.zeros(n, device=dev)
        self._buf = torch.zeros(n, device=dev)
        self._spans = []
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat_params[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_params[off:off + k].view(p.shape)
                self._spans.append((off, k))
                off += ((k + 63) // 64) * 64

    def gather_grads(self, skip=None, prezeroed=False):
        """Per-tensor autograd gradients -> the flat gradient buffer (missing gradients count as zero).  ``skip``: ids of
        parameters whose slice somebody else fills (ops.WeightPlan: filters unpacked right afterwards, BatchNorm gradients
        written in place by their kernels).  ``prezeroed``: the buffer was cleared before the backward pass (WeightPlan.prepare):
        gradient-less parameters need no fill of their own."""
        with torch.no_grad():
            dst, src = [], []
            for p, (off, k) in zip(self._params, self._spans):
                if p.grad is None:
                    if not prezeroed and (skip is None or id(p) not in skip):
                        self.flat_grads[off:off + k].zero_()
                else:
                    dst.append(self.flat_grads[off:off + k].view(p.shape))
                    src.append(p.grad)
            if dst:
                # one multi-tensor copy instead of a launch per parameter (680 tensors in InstaDepthNet_od)
                torch._foreach_copy_(dst, src)
        return self.flat_grads

    def gather_stage(self, idx, grads, skip=None, attach=False, prezeroed=False):
        """The gradients of the parameters ``idx`` (indices into the flat layout, one stage of a staged backward pass) ->
        their slices of the flat gradient buffer; ``grads``: what torch.autograd.grad returned for them (None = no
        gradient: zeros, unless the parameter is in ``skip`` -- ops.WeightPlan fills those).  attach: ``p.grad`` = the view."""
        with torch.no_grad():
            dst, src = [], []
            for i, g in zip(idx, grads):
                p = self._params[i]
                off, k = self._spans[i]
                view = self.flat_grads[off:off + k].view(p.shape)
                if g is None:
                    if not prezeroed and (skip is None or id(p) not in skip):
                        view.zero_()
                else:
                    dst.append(view)
                    src.append(g)
                if attach:
                    p.grad = view
            if dst:
                torch._foreach_copy_(dst, src)

    def offset_of(self, param):
        for p, (off, _) in zip(self._params, self._spans):
            if p is param:
                return off
        raise KeyError("parameter is not in the flat buffer")

    @torch.no_grad()
    def step(self, closure=None, gathered=False):
        g = self.param_groups[0]
        if not gathered:
            self.gather_grads()
        engine.sgd_momentum(self.flat_params, self.flat_grads, self._buf, g["lr"], g["momentum"], g["weight_decay"])
        from . import ops
        ops.WEIGHTS_EPOCH[0] += 1          # parameters changed in place, invisibly to torch's version counters
        return None

    def state_dict(self):
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self._params)))
        state = {i: {"momentum_buffer": self._buf[off:off + k].view(p.shape).detach().clone()}
                 for i, (p, (off, k)) in enumerate(zip(self._params, self._spans))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = v
        with torch.no_grad():
            for i, st in sd["state"].items():
                mb = st.get("momentum_buffer")
                if mb is not None:
                    off, k = self._spans[int(i)]
                    self._buf[off:off + k].copy_(mb.reshape(-1))
