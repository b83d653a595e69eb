"""Operator layer for graphs the fixed ResNet-50 executor (io_net_*) does not cover -- today the MiDaS branch
(InstaDepthNet_od / _d, midas/midas_net.py:116-212).  Every function is a ``torch.autograd.Function`` whose forward
and backward are launches of libinstaorder_hip.so on **NHWC** tensors, fp32 or bf16 (the element type of the input
selects the kernels; parameters and their gradients are always fp32); torch supplies memory, streams and the autograd
tape only.  Filters are taken in the reference's OIHW layout (what ``state_dict`` holds) and re-laid out to the
kernels' [Cout][taps][Cin] (in the activation type) once per call.

There is no CPU path: inputs must live on an MI355X.
"""
import ctypes as C

import torch

from . import _lib

__all__ = ["conv2d", "conv_bn", "batch_norm", "max_pool_3x3s2", "avgpool_fc", "upsample2x", "bias_act", "relu", "add", "head1",
           "nhwc_from_nchw", "nchw_from_nhwc"]


# bumped by whoever changes parameters behind torch's back (the HIP optimiser kernels update them in place without
# touching torch's version counters): invalidates the folded inference operands cached on the parameters
WEIGHTS_EPOCH = [0]


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _WeightDesc(C.Structure):        # io_weight_desc of include/instaorder_hip.h
    _fields_ = [("src", C.c_long), ("dst_op", C.c_long), ("dst_t", C.c_long), ("dst_g", C.c_long), ("Co", C.c_int),
                ("Ci", C.c_int), ("T", C.c_int), ("Cop", C.c_int), ("Cip", C.c_int)]


class WeightPlan(object):
    """Every dense filter of a module tree re-laid out in ONE launch per training step, instead of three to six tiny
    launches per convolution (OIHW master -> [Cout'][taps][Cin'] operand + its transpose for the data gradient; filter
    gradient back to OIHW): ``prepare()`` before the forward, ``unpack_grads()`` after the backward, and in between
    ``_Conv`` / ``_ConvBn`` pick their operands up as views of the plan's buffers (the filter gradient is written into
    the plan's staging buffer and the node returns no gradient for ``w``).

    Built from a RECORDING of one eager training step (which filters are used, with which stored channel counts), so
    it needs no knowledge of the module tree; a filter used more than once per step stays on the per-call path (its
    gradients must accumulate).  Only for parameters that live in a flat buffer (optim.FlatSGD)."""
    active = None          # the plan convolutions consult (set for the duration of a training step)
    _recording = None      # id(w) -> [w, Cip, Cop, uses] while a step is being recorded

    @classmethod
    def start_recording(cls):
        cls._recording = {}

    @classmethod
    def note(cls, w, Cip, Cop):
        if cls._recording is not None:
            r = cls._recording.setdefault(id(w), [w, Cip, Cop, 0])
            r[3] += 1
            if (r[1], r[2]) != (Cip, Cop):
                r[3] += 100        # used with two layouts: never plan it

    @classmethod
    def note_vec(cls, *ps):
        """the 1-D parameters a node consumes (BatchNorm weight / bias): recorded like the filters -- used exactly once per
        step, their gradient kernels may write straight into the flat gradient buffer (``grad_target``)"""
        if cls._recording is not None:
            for q in ps:
                if q is not None:
                    cls._recording.setdefault(id(q), [q, -1, -1, 0])[3] += 1

    @classmethod
    def grad_target(cls, q):
        """where the gradient of the 1-D parameter ``q`` goes in the step in flight: its slice of optim.flat_grads (the node
        then returns no gradient for it), or None = an ordinary autograd gradient"""
        plan = cls.active
        return plan.vecs.get(id(q)) if (plan is not None and q is not None) else None

    @classmethod
    def stop_recording(cls):
        recs, cls._recording = cls._recording, None
        return [r for r in (recs or {}).values() if r[3] == 1]

    def __init__(self, optim, recs, dtype):
        spans = {id(p): off for p, (off, k) in zip(optim._params, optim._spans)}
        dev = optim.flat_params.device
        self.optim, self.dtype = optim, dtype
        self.entries = {}
        rows, n_op, n_g = [], 0, 0
        # rows in the order of the flat parameter buffer: a contiguous slice of that buffer (one stage of a staged,
        # data-parallel backward pass) is then a contiguous range of rows (unpack_grads(lo, hi))
        for w, Cip, Cop, _ in sorted((r for r in recs if id(r[0]) in spans), key=lambda r: spans[id(r[0])]):
            if id(w) not in spans or w.dim() != 4:
                continue
            Co, Ci, R, S = w.shape
            T = R * S
            sz = Cop * T * Cip
            need_t = Cip != 8                       # the packed 8-channel stems have no data gradient
            rows.append((w, _WeightDesc(spans[id(w)], n_op, n_op + sz if need_t else -1, n_g, Co, Ci, T, Cop, Cip)))
            n_op += sz * (2 if need_t else 1)
            n_g += sz
        # BatchNorm weights / biases used once per step: BatchNorm-backward launches write dgamma / dbeta into these views of
        # the flat gradient buffer -- no per-tensor gradient, no copy into the flat buffer (gather_grads made 286 of them
        # per InstaDepthNet_od step: torch._foreach_copy_ decomposes into one hipMemcpyAsync per tensor)
        self.vecs = {}
        import os
        direct = os.environ.get("IO_DEPTH_DIRECT_GRADS", "1") != "0"      # (0: every BatchNorm gradient through autograd + gather -- A/B runs)
        for q, Cip, _, _ in recs:
            if direct and Cip == -1 and id(q) in spans and q.dim() == 1:
                off = spans[id(q)]
                self.vecs[id(q)] = optim.flat_grads[off:off + q.numel()]
        self.skip_ids = None
        self.n = len(rows)
        self._row_off = [d.src for _, d in rows]
        self.ops = torch.zeros(max(n_op, 1), device=dev, dtype=dtype)
        self.gk = torch.zeros(max(n_g, 1), device=dev, dtype=torch.float32)
        tab = (_WeightDesc * max(self.n, 1))(*[d for _, d in rows])
        self.table = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
        for w, d in rows:
            sz = d.Cop * d.T * d.Cip
            self.entries[id(w)] = (d.Cip, d.Cop,
                                   self.ops[d.dst_op:d.dst_op + sz].view(d.Cop, d.T, d.Cip),
                                   self.ops[d.dst_t:d.dst_t + sz].view(d.Cip, d.T, d.Cop) if d.dst_t >= 0 else None,
                                   self.gk[d.dst_g:d.dst_g + sz].view(d.Cop, d.T, d.Cip))

    @property
    def skip(self):
        """ids of the parameters whose slice of the flat gradient buffer this plan fills (optim.gather_grads leaves them alone)"""
        if self.skip_ids is None:
            self.skip_ids = set(self.entries) | set(self.vecs)
        return self.skip_ids

    def lookup(self, w, Cip, Cop, dtype):
        e = self.entries.get(id(w))
        return e if (e is not None and e[0] == Cip and e[1] == Cop and dtype == self.dtype) else None

    def prepare(self):
        if self.n:
            # A planned filter whose gradient kernel does not run in this step (a loss weight switched to 0 after the plan
            # was recorded) must deliver zero, not the previous step's gradient: unpack_grads() copies every entry of gk.
            self.gk.zero_()
            if self.vecs:
                # ... and so must a directly written BatchNorm gradient, and every parameter no loss reaches (the occlusion
                # branch of the reference's InstaDepthNet_od recipe: occ_order_weight 0) -- ONE fill of the flat buffer
                # instead of one per gradient-less parameter in gather_grads (129 per step)
                self.optim.flat_grads.zero_()
            _lib.check(_L().io_weights_prepare(_p(self.table), self.n, _p(self.optim.flat_params), _p(self.ops),
                                               1 if self.dtype == torch.bfloat16 else 0, _st()), "io_weights_prepare")

    def unpack_grads(self, lo=None, hi=None):
        """after optim.gather_grads(): the planned filters' gradients into their OIHW places in the flat buffer; with
        (lo, hi) only those of the filters that live in [lo, hi) of the flat buffer (rows are in flat order)"""
        if not self.n:
            return
        r0, r1 = 0, self.n
        if lo is not None:
            r0 = next((i for i, o in enumerate(self._row_off) if o >= lo), self.n)
            r1 = next((i for i, o in enumerate(self._row_off) if o >= hi), self.n)
        if r1 > r0:
            tab = C.c_void_p(self.table.data_ptr() + r0 * C.sizeof(_WeightDesc))
            _lib.check(_L().io_weights_unpack_grads(tab, r1 - r0, _p(self.gk), _p(self.optim.flat_grads), _st()),
                       "io_weights_unpack_grads")


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _L():
    _lib.require_gpu()
    return _lib.lib()


def _chk(t, name):
    if not (t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and t.is_contiguous()):
        raise RuntimeError("instaorder_amd.ops: %s must be a contiguous fp32 / bf16 tensor on the GPU (no CPU fallback)"
                           % name)


def _dt(t):
    return 1 if t.dtype == torch.bfloat16 else 0


def nhwc_from_nchw(x, pad_to=None, dtype=None):
    """[N,C,H,W] -> contiguous [N,H,W,C'] (C' = pad_to, zero filled, for the 8-channel stems) of ``dtype``."""
    N, Cc, H, W = x.shape
    dtype = dtype or x.dtype
    if pad_to is not None and pad_to != Cc:
        out = torch.zeros((N, H, W, pad_to), device=x.device, dtype=dtype)
        out[..., :Cc] = x.permute(0, 2, 3, 1)
        return out
    return x.permute(0, 2, 3, 1).contiguous().to(dtype)


def nchw_from_nhwc(x):
    return x.permute(0, 3, 1, 2).contiguous()


# ---- convolution -------------------------------------------------------------------------------------------------------
class _Conv(torch.autograd.Function):
    """Dense convolution (models/backbone/resnet_cls.py:23-31, midas/blocks.py:27-38, 133-139).  x[N,H,W,Ci'] where
    Ci' >= Ci is the stored channel count (8 for the 3- / 2-channel stems); w OIHW [Co,Ci,R,S].  Output channels are
    padded to a multiple of 64 when `co_pad` (the 128 -> 32 convolution of output_conv): the extra channels are 0."""

    @staticmethod
    def forward(ctx, x, w, stride, pad, co_pad):
        _chk(x, "x")
        N, H, W_, Cs = x.shape
        Co, Ci, R, S = w.shape
        Cop = ((Co + 63) // 64) * 64 if co_pad else Co
        WeightPlan.note(w, Cs, Cop)
        ent = WeightPlan.active.lookup(w, Cs, Cop, x.dtype) if WeightPlan.active is not None else None
        if ent is not None:
            wk = ent[2]
        else:
            wk = torch.zeros((Cop, R * S, Cs), device=x.device, dtype=x.dtype)
            wk[:Co, :, :Ci] = w.detach().permute(0, 2, 3, 1).reshape(Co, R * S, Ci)
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W_ + 2 * pad - S) // stride + 1
        y = torch.empty((N, Ho, Wo, Cop), device=x.device, dtype=x.dtype)
        _lib.check(_L().io_conv2d_fwd_dt(_p(x), _p(wk), _p(y), N, H, W_, Cs, Cop, R, S, stride, pad, _dt(x), _dt(x), _st()),
                   "io_conv2d_fwd_dt")
        ctx.save_for_backward(x, wk)
        ctx.ent = ent
        ctx.geom = (N, H, W_, Cs, Cop, Co, Ci, R, S, stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wk = ctx.saved_tensors
        N, H, W_, Cs, Cop, Co, Ci, R, S, stride, pad = ctx.geom
        dy = dy.contiguous()
        L = _L()
        dx = None
        if ctx.needs_input_grad[0]:
            if Cs == 8:
                raise RuntimeError("ops.conv2d: no data gradient for the packed 8-channel stem input")
            ent = ctx.ent
            wt = ent[3] if ent is not None else wk.permute(2, 1, 0).contiguous()     # [Cin][taps][Cout]
            dx = torch.empty_like(x)
            _lib.check(L.io_conv2d_dgrad_dt(_p(dy), _p(wt), _p(dx), None, None, N, H, W_, Cs, Cop, R, S, stride, pad, _dt(x),
                                            _st()), "io_conv2d_dgrad_dt")
        dw = None
        if ctx.needs_input_grad[1]:
            ent = ctx.ent
            nb = int(L.io_conv2d_wgrad_workspace_bytes(N, H, W_, Cs, Cop, R, S, stride, pad))
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=x.device)
            dwk = ent[4] if ent is not None else torch.empty((Cop, R * S, Cs), device=x.device, dtype=torch.float32)
            _lib.check(L.io_conv2d_wgrad_dt(_p(x), _p(dy), _p(dwk), N, H, W_, Cs, Cop, R, S, stride, pad, _p(ws), nb, _dt(x),
                                            _dt(x), _st()), "io_conv2d_wgrad_dt")
            if ent is None:        # (planned filters: WeightPlan.unpack_grads delivers the gradient)
                dw = dwk[:Co, :, :Ci].reshape(Co, R, S, Ci).permute(0, 3, 1, 2).contiguous()
        return dx, dw, None, None, None


class _GroupedConv(torch.autograd.Function):
    """Grouped 3x3 convolution of ResNeXt (resnet_cls.py:23-26 with groups=32): w OIHW [C, C/groups, R, S]."""

    @staticmethod
    def forward(ctx, x, w, stride, pad):
        _chk(x, "x")
        N, H, W_, Cc = x.shape
        Co, cg, R, S = w.shape
        if Co != Cc:
            raise ValueError("grouped conv: Cin must equal Cout")
        L = _L()
        wc = torch.empty((Cc, R * S, 64), device=x.device, dtype=x.dtype)
        wtc = torch.empty_like(wc)
        _lib.check(L.io_gconv_pack(_p(w.detach().contiguous()), Cc, cg, R * S, _p(wc), _p(wtc), _dt(x), _st()),
                   "io_gconv_pack")
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W_ + 2 * pad - S) // stride + 1
        y = torch.empty((N, Ho, Wo, Cc), device=x.device, dtype=x.dtype)
        _lib.check(L.io_gconv2d_fwd(_p(x), _p(wc), _p(y), N, H, W_, Cc, R, S, stride, pad, _dt(x), _st()), "io_gconv2d_fwd")
        ctx.save_for_backward(x, wtc)
        ctx.geom = (N, H, W_, Cc, cg, R, S, stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wtc = ctx.saved_tensors
        N, H, W_, Cc, cg, R, S, stride, pad = ctx.geom
        dy = dy.contiguous()
        L = _L()
        dx = torch.empty_like(x)
        _lib.check(L.io_gconv2d_dgrad(_p(dy), _p(wtc), _p(dx), N, H, W_, Cc, R, S, stride, pad, _dt(x), _st()),
                   "io_gconv2d_dgrad")
        nb = int(L.io_gconv2d_wgrad_workspace_bytes(N, H, W_, Cc, R, S, stride, pad))
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=x.device)
        dwc = torch.empty((Cc, R * S, 64), device=x.device, dtype=torch.float32)
        _lib.check(L.io_gconv2d_wgrad(_p(x), _p(dy), _p(dwc), N, H, W_, Cc, R, S, stride, pad, _p(ws), nb, _dt(x), _st()),
                   "io_gconv2d_wgrad")
        dw = torch.empty((Cc, cg, R, S), device=x.device, dtype=torch.float32)
        _lib.check(L.io_gconv_unpack_grad(_p(dwc), Cc, cg, R * S, _p(dw), _st()), "io_gconv_unpack_grad")
        return dx, dw, None, None


def conv2d(x, w, stride=1, pad=0, groups=1, co_pad=False):
    if groups == 1:
        return _Conv.apply(x, w, stride, pad, co_pad)
    if w.shape[0] // groups != w.shape[1]:
        raise ValueError("grouped conv: expected [C, C/groups, R, S] filters")
    return _GroupedConv.apply(x, w, stride, pad)


# ---- BatchNorm (+ residual add) (+ ReLU) ---------------------------------------------------------------------------------
def _momentum(repeat):
    """R identical module calls advance a running estimate R times with the same statistic:
    r <- (1-m)^R r + (1 - (1-m)^R) s, i.e. ONE update with momentum 1 - (1-m)^R (m = 0.1)."""
    return 1.0 - 0.9 ** int(repeat)


class _BatchNorm(torch.autograd.Function):
    """nn.BatchNorm2d (+ `out += identity`) (+ nn.ReLU) of resnet_cls.py:96-116.  Training: batch statistics, running
    estimates advanced in place (momentum 0.1, unbiased variance); eval: running estimates.
    groups = G: the batch is G consecutive equal parts normalised with separate statistics, running estimates advanced
    part by part -- exactly G sequential module calls on the parts (the two mask orders of a pair batch).
    repeat = R: the module call is accounted R times (running estimates advanced R times with the same statistics):
    the reference runs the shared encoder once per mask order on identical input."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, relu, identity, groups, repeat):
        _chk(x, "x")
        if training:
            WEIGHTS_EPOCH[0] += 1      # running statistics move (behind torch's version counters): folded operands are stale
        L = _L()
        Cc = x.shape[-1]
        M = x.numel() // Cc
        dev = x.device
        G = int(groups) if training else 1
        mean, rstd, scale, shift = (torch.empty(G * Cc, device=dev, dtype=torch.float32) for _ in range(4))
        if training:
            WeightPlan.note_vec(gamma, beta)
            npart = int(L.io_bn_partial_floats(M, Cc, G))
            part = torch.empty(npart, device=dev, dtype=torch.float32)
            _lib.check(L.io_bn_stats_finalize_dt(_p(x), M, Cc, G, _p(gamma.detach()), _p(beta.detach()), _p(running_mean),
                                                 _p(running_var), _momentum(repeat), 1e-5, _p(mean), _p(rstd), _p(scale),
                                                 _p(shift), _p(part), npart, _dt(x), _st()), "io_bn_stats_finalize_dt")
        else:
            _lib.check(L.io_bn_eval_prepare(Cc, _p(gamma.detach()), _p(beta.detach()), _p(running_mean), _p(running_var),
                                            1e-5, _p(mean), _p(scale), _p(shift), _st()), "io_bn_eval_prepare")
            rstd = scale / gamma.detach()
        out = torch.empty_like(x)
        _lib.check(L.io_bn_apply_dt(_p(x), M, Cc, G, 1 if training else 0, _p(mean), _p(scale), _p(shift), _p(identity), None,
                                    None, None, int(relu), _p(out), _dt(x), _st()), "io_bn_apply_dt")
        ctx.save_for_backward(x, out, gamma, mean, rstd)
        ctx.beta = beta
        ctx.cfg = (M, Cc, bool(relu), identity is not None, bool(training), G)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, gamma, mean, rstd = ctx.saved_tensors
        M, Cc, relu, has_id, training, G = ctx.cfg
        if not training:
            raise RuntimeError("ops.batch_norm: backward through eval-mode BatchNorm is not implemented")
        L = _L()
        dout = dout.contiguous()
        dev = x.device
        npart = int(L.io_bn_partial_floats(M, Cc, G))
        part = torch.empty(npart, device=dev, dtype=torch.float32)
        coef = torch.empty(2 * G * Cc, device=dev, dtype=torch.float32)
        tg, tb = WeightPlan.grad_target(gamma), WeightPlan.grad_target(ctx.beta)
        dgamma = tg if tg is not None else torch.empty(Cc, device=dev)
        dbeta = tb if tb is not None else torch.empty(Cc, device=dev)
        dx = torch.empty_like(x)
        dz = torch.empty_like(x) if has_id else None          # gradient of the pre-ReLU sum = gradient of `identity`
        _lib.check(L.io_bn_bwd_dt(_p(dout), _p(out) if relu else None, None, None, _p(x), M, Cc, G, _p(gamma.detach()),
                                  _p(mean), _p(rstd), _p(dgamma), _p(dbeta), _p(dx), _p(dz), _p(part), npart, _p(coef),
                                  _dt(x), _st()), "io_bn_bwd_dt")
        return dx, (None if tg is not None else dgamma), (None if tb is not None else dbeta), None, None, None, None, dz, None, None


def batch_norm(x, gamma, beta, running_mean, running_var, training, relu=False, identity=None, groups=1, repeat=1):
    if repeat > 1 and groups > 1:
        raise ValueError("batch_norm: repeat and groups are exclusive")
    return _BatchNorm.apply(x, gamma, beta, running_mean, running_var, training, relu, identity, groups, repeat)


class _ConvBn(torch.autograd.Function):
    """conv (dense or grouped) -> BatchNorm (+ identity) (+ ReLU) as ONE node: in training mode the batch statistics are
    accumulated in the convolution's epilogue (per 128-row tile, merged with Chan's update) whenever the rows per
    statistics group are a multiple of 128 -- no separate pass over the conv output (resnet_cls.py:99-114)."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, running_mean, running_var, stride, pad, groups, training, relu, identity,
                bn_groups, repeat, prev, fork):
        _chk(x, "x")
        ctx.prev = None
        ctx.link = None
        ctx.fork = bool(fork)
        out = _ConvBn._forward(ctx, x, w, gamma, beta, running_mean, running_var, stride, pad, groups, training, relu,
                               identity, bn_groups, repeat, prev)
        # fork: x comes back as a second output (an alias) -- for the residual connection of the Bottleneck this convolution
        # opens -- so that the gradient of that path arrives HERE and rides in the `add` operand of the data-gradient
        # launch instead of being summed onto dx by a separate ATen kernel of the autograd engine
        return (out, x) if fork else out

    @staticmethod
    def _forward(ctx, x, w, gamma, beta, running_mean, running_var, stride, pad, groups, training, relu, identity,
                 bn_groups, repeat, prev):
        _ConvBn.last_offer = None
        if training:
            WEIGHTS_EPOCH[0] += 1      # running statistics move (behind torch's version counters): folded operands are stale
        L = _L()
        dev, dt = x.device, _dt(x)
        N, H, W_, Cs = x.shape
        Co, Cig, R, S = w.shape
        dense = groups == 1
        wsrc = w.detach()
        cache_key = None
        if not training:
            # inference: the BatchNorm is folded into the filters (scaled per output channel) and a bias; the
            # convolution's epilogue adds bias (+ identity) (+ ReLU) -- no pass over the conv output.  The folded
            # operands are kept on the parameter until the weights change (torch version counters + WEIGHTS_EPOCH,
            # which the HIP-side optimiser bumps): an inference loop re-packs nothing.
            cache_key = (x.dtype, Cs, WEIGHTS_EPOCH[0], w._version, gamma._version, beta._version, running_mean._version,
                         running_var._version)
            hit = getattr(w, "_io_folded", None)
            if hit is not None and hit[0] == cache_key:
                wop, fbias = hit[1], hit[2]
                Ho, Wo = (H + 2 * pad - R) // stride + 1, (W_ + 2 * pad - S) // stride + 1
                y = torch.empty((N, Ho, Wo, Co), device=dev, dtype=x.dtype)
                _lib.check(L.io_conv2d_fwd_bias_dt(_p(x), _p(wop), _p(y), N, H, W_, Cs, Co, R, S, stride, pad, _p(fbias),
                                                   _p(identity), int(relu), dt, 0 if dense else 64, _st()),
                           "io_conv2d_fwd_bias_dt")
                ctx.cfg = (N, H, W_, Cs, Co, Cig, R, S, stride, pad, dense, N * Ho * Wo, 1, bool(relu),
                           identity is not None, False)
                return y
            fscale = gamma.detach() / torch.sqrt(running_var + 1e-5)
            fbias = (beta.detach() - running_mean * fscale).contiguous()
            wsrc = wsrc * fscale.view(-1, 1, 1, 1)
        ent = None
        if dense:
            if training:
                WeightPlan.note(w, Cs, Co)
                ent = WeightPlan.active.lookup(w, Cs, Co, x.dtype) if WeightPlan.active is not None else None
            if ent is not None:
                wop = ent[2]
            else:
                wop = torch.zeros((Co, R * S, Cs), device=dev, dtype=x.dtype)
                wop[:, :, :Cig] = wsrc.permute(0, 2, 3, 1).reshape(Co, R * S, Cig)
            wback = wop
        else:
            if Co != Cs or Co // groups != Cig:
                raise ValueError("grouped conv: expected [C, C/groups, R, S] filters on C input channels")
            wop = torch.empty((Co, R * S, 64), device=dev, dtype=x.dtype)
            wback = torch.empty_like(wop)
            _lib.check(L.io_gconv_pack(_p(wsrc.contiguous()), Co, Cig, R * S, _p(wop), _p(wback), dt, _st()),
                       "io_gconv_pack")
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W_ + 2 * pad - S) // stride + 1
        M = N * Ho * Wo
        G = int(bn_groups) if training else 1
        y = torch.empty((N, Ho, Wo, Co), device=dev, dtype=x.dtype)
        if not training:
            w._io_folded = (cache_key, wop, fbias)
            _lib.check(L.io_conv2d_fwd_bias_dt(_p(x), _p(wop), _p(y), N, H, W_, Cs, Co, R, S, stride, pad, _p(fbias),
                                               _p(identity), int(relu), dt, 0 if dense else 64, _st()),
                       "io_conv2d_fwd_bias_dt")
            ctx.cfg = (N, H, W_, Cs, Co, Cig, R, S, stride, pad, dense, M, G, bool(relu), identity is not None, False)
            return y
        mean, rstd, scale, shift = (torch.empty(G * Co, device=dev, dtype=torch.float32) for _ in range(4))
        gd, bd = gamma.detach(), beta.detach()
        if training:
            WeightPlan.note_vec(gamma, beta)
        ctx.beta = beta
        if training and M % G == 0 and (M // G) % 128 == 0:
            nws = int(L.io_conv2d_bnstats_workspace_floats(N, H, W_, Co, R, S, stride, pad, G))
            ws = torch.empty(nws, device=dev, dtype=torch.float32)
            _lib.check(L.io_conv2d_fwd_bnstats_dt(_p(x), _p(wop), _p(y), N, H, W_, Cs, Co, R, S, stride, pad, G, _p(gd), _p(bd),
                                                  _p(running_mean), _p(running_var), _momentum(repeat), 1e-5, _p(mean),
                                                  _p(rstd), _p(scale), _p(shift), _p(ws), nws, dt, 0 if dense else 64, _st()),
                       "io_conv2d_fwd_bnstats_dt")
        else:
            if dense:
                _lib.check(L.io_conv2d_fwd_dt(_p(x), _p(wop), _p(y), N, H, W_, Cs, Co, R, S, stride, pad, dt, dt, _st()),
                           "io_conv2d_fwd_dt")
            else:
                _lib.check(L.io_gconv2d_fwd(_p(x), _p(wop), _p(y), N, H, W_, Co, R, S, stride, pad, dt, _st()), "io_gconv2d_fwd")
            if training:
                npart = int(L.io_bn_partial_floats(M, Co, G))
                part = torch.empty(npart, device=dev, dtype=torch.float32)
                _lib.check(L.io_bn_stats_finalize_dt(_p(y), M, Co, G, _p(gd), _p(bd), _p(running_mean), _p(running_var),
                                                     _momentum(repeat), 1e-5, _p(mean), _p(rstd), _p(scale), _p(shift),
                                                     _p(part), npart, dt, _st()), "io_bn_stats_finalize_dt")
            else:
                _lib.check(L.io_bn_eval_prepare(Co, _p(gd), _p(bd), _p(running_mean), _p(running_var), 1e-5, _p(mean),
                                                _p(scale), _p(shift), _st()), "io_bn_eval_prepare")
                rstd = scale / gd
        out = torch.empty_like(y)
        _lib.check(L.io_bn_apply_dt(_p(y), M, Co, G, 1 if training else 0, _p(mean), _p(scale), _p(shift), _p(identity), None,
                                    None, None, int(relu), _p(out), dt, _st()), "io_bn_apply_dt")
        ctx.save_for_backward(x, wback, y, out, gamma, mean, rstd)
        ctx.ent = ent
        ctx.cfg = (N, H, W_, Cs, Co, Cig, R, S, stride, pad, dense, M, G, bool(relu), identity is not None, bool(training))
        # Cross-node fusion of the BatchNorm backward (see conv_bn's `sole`): as PRODUCER of relu(bn(y)) this node offers
        # what a consumer's data-gradient epilogue needs; as CONSUMER of such a tensor it keeps the producer's offer.
        if training and relu and identity is None and M % G == 0 and (M // G) % 128 == 0 and Co % 64 == 0:
            ctx.link = {"_gamma": gamma, "_beta": beta}       # (the parameters: a consumer that runs this BatchNorm's backward asks WeightPlan where their gradients go)
            _ConvBn.last_offer = (ctx.link, y, mean, rstd, scale, shift, gamma.detach(), G, M, Co)
        else:
            _ConvBn.last_offer = None
        if prev is not None and stride == 1 and prev[8] == N * H * W_ and prev[9] == Cs and (dense or Co == Cs):
            ctx.prev = prev
        return out

    @staticmethod
    def backward(ctx, dout, dfork=None):
        N, H, W_, Cs, Co, Cig, R, S, stride, pad, dense, M, G, relu, has_id, training = ctx.cfg
        if not training:
            raise RuntimeError("ops.conv_bn: backward through eval-mode BatchNorm is not implemented")
        if dfork is not None:
            dfork = dfork.contiguous()
        x, wback, y, out, gamma, mean, rstd = ctx.saved_tensors
        L = _L()
        dev, dt = x.device, _dt(x)
        dout = dout.contiguous()
        npart = int(L.io_bn_partial_floats(M, Co, G))
        part = torch.empty(npart, device=dev, dtype=torch.float32)
        coef = torch.empty(2 * G * Co, device=dev, dtype=torch.float32)
        tg, tb = WeightPlan.grad_target(gamma), WeightPlan.grad_target(ctx.beta)
        link = ctx.link
        if link is not None and "dy" in link:
            # the sole consumer of this node's output already ran this BatchNorm's backward in its data-gradient launch
            # (and wrote dgamma / dbeta to the same targets)
            dy, dgamma, dbeta = link.pop("dy"), link.pop("dgamma"), link.pop("dbeta")
            dz = None
        else:
            dgamma = tg if tg is not None else torch.empty(Co, device=dev)
            dbeta = tb if tb is not None else torch.empty(Co, device=dev)
            dy = torch.empty_like(y)
            dz = torch.empty_like(y) if has_id else None
            _lib.check(L.io_bn_bwd_dt(_p(dout), _p(out) if relu else None, None, None, _p(y), M, Co, G, _p(gamma.detach()),
                                      _p(mean), _p(rstd), _p(dgamma), _p(dbeta), _p(dy), _p(dz), _p(part), npart, _p(coef),
                                      dt, _st()), "io_bn_bwd_dt")
        dx = None
        ent = getattr(ctx, "ent", None)
        prev = ctx.prev

        def fused_dgrad(wt_op, gw):
            """data gradient + the PRODUCER's BatchNorm backward: ReLU mask recomputed from its y, its two reductions in
            the epilogue of this launch, then only the apply pass -- the producer node finds its results in the link"""
            link, yp, mean_p, rstd_p, scale_p, shift_p, gamma_p, Gp, Mp, Cp = prev
            tiles = Mp // 128
            nws = 2 * ((tiles + tiles // 64 + Gp + 2) * Cp) + 2 * Gp * Cp
            wsf = torch.empty(nws, device=dev, dtype=torch.float32)
            dz = torch.empty_like(x)
            dyb = torch.empty_like(x)
            tgp, tbp = WeightPlan.grad_target(link.get("_gamma")), WeightPlan.grad_target(link.get("_beta"))
            dg = tgp if tgp is not None else torch.empty(Cp, device=dev)
            db = tbp if tbp is not None else torch.empty(Cp, device=dev)
            _lib.check(L.io_conv2d_dgrad_bnbwd_dt(_p(dy), _p(wt_op), _p(dz), N, H, W_, Cs, Co, R, S, pad, _p(yp), Gp,
                                                  _p(gamma_p), _p(mean_p), _p(rstd_p), _p(scale_p), _p(shift_p), _p(dg),
                                                  _p(db), _p(dyb), _p(wsf), nws, dt, gw, _st()), "io_conv2d_dgrad_bnbwd_dt")
            link["dy"], link["dgamma"], link["dbeta"] = dyb, dg, db
            return dz

        if dense:
            if ctx.needs_input_grad[0]:
                if Cs == 8:
                    raise RuntimeError("ops.conv_bn: no data gradient for the packed 8-channel stem input")
                wt = ent[3] if ent is not None else wback.permute(2, 1, 0).contiguous()
                if prev is not None:
                    dx = fused_dgrad(wt, 0)
                else:
                    dx = torch.empty_like(x)
                    _lib.check(L.io_conv2d_dgrad_dt(_p(dy), _p(wt), _p(dx), _p(dfork), None, N, H, W_, Cs, Co, R, S, stride, pad,
                                                    dt, _st()), "io_conv2d_dgrad_dt")
                    dfork = None                 # (summed inside the launch)
            nb = int(L.io_conv2d_wgrad_workspace_bytes(N, H, W_, Cs, Co, R, S, stride, pad))
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
            dwk = ent[4] if ent is not None else torch.empty((Co, R * S, Cs), device=dev, dtype=torch.float32)
            _lib.check(L.io_conv2d_wgrad_dt(_p(x), _p(dy), _p(dwk), N, H, W_, Cs, Co, R, S, stride, pad, _p(ws), nb, dt, dt,
                                            _st()), "io_conv2d_wgrad_dt")
            # (planned filters: WeightPlan.unpack_grads delivers the gradient)
            dw = None if ent is not None else dwk[:, :, :Cig].reshape(Co, R, S, Cig).permute(0, 3, 1, 2).contiguous()
        else:
            if prev is not None:
                dx = fused_dgrad(wback, 64)
            else:
                dx = torch.empty_like(x)
                _lib.check(L.io_gconv2d_dgrad(_p(dy), _p(wback), _p(dx), N, H, W_, Co, R, S, stride, pad, dt, _st()),
                           "io_gconv2d_dgrad")
            nb = int(L.io_gconv2d_wgrad_workspace_bytes(N, H, W_, Co, R, S, stride, pad))
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
            dwc = torch.empty((Co, R * S, 64), device=dev, dtype=torch.float32)
            _lib.check(L.io_gconv2d_wgrad(_p(x), _p(dy), _p(dwc), N, H, W_, Co, R, S, stride, pad, _p(ws), nb, dt, _st()),
                       "io_gconv2d_wgrad")
            dw = torch.empty((Co, Cig, R, S), device=dev, dtype=torch.float32)
            _lib.check(L.io_gconv_unpack_grad(_p(dwc), Co, Cig, R * S, _p(dw), _st()), "io_gconv_unpack_grad")
        if dfork is not None:                    # (forms whose data gradient has no add operand: summed here)
            dx = dfork if dx is None else dx + dfork
        return (dx, dw, (None if tg is not None else dgamma), (None if tb is not None else dbeta), None, None, None, None, None,
                None, None, dz, None, None, None, None)


_ConvBn.last_offer = None


def conv_bn(x, w, gamma, beta, running_mean, running_var, stride, pad, groups, training, relu=False, identity=None,
            bn_groups=1, repeat=1, sole=False, fork=False):
    """``sole=True``: the caller promises that THIS call is the only consumer of ``x`` (as conv2 / conv3 of a Bottleneck
    are of relu(bn1(.)) / relu(bn2(.)), resnet_cls.py:99-111).  If ``x`` came out of a conv_bn with ReLU, the backward of
    that BatchNorm then runs inside this node's data-gradient launch (mask recomputed from the producer's conv output,
    reductions in the epilogue, io_conv2d_dgrad_bnbwd_dt) instead of as separate reduce + apply passes.
    ``fork=True`` (training, x requires a gradient): returns ``(out, x_alias)``; the caller routes every OTHER use of x (the
    residual connection / the downsample branch of resnet_cls.py:96-114) through ``x_alias``, whose gradient then enters this
    node and is added inside its data-gradient launch (io_conv2d_dgrad_dt's `add`) -- one launch instead of the data gradient
    plus the autograd engine's accumulation kernel (65 of them per InstaDepthNet_od step).  Otherwise returns ``out`` alone."""
    if repeat > 1 and bn_groups > 1:
        raise ValueError("conv_bn: repeat and bn_groups are exclusive")
    prev = getattr(x, "_io_offer", None) if (sole and training) else None
    if prev is not None and prev[7] != (int(bn_groups) if training else 1):
        prev = None
    fork = bool(fork and training and torch.is_grad_enabled() and x.requires_grad)
    out = _ConvBn.apply(x, w, gamma, beta, running_mean, running_var, stride, pad, groups, training, relu, identity,
                        bn_groups, repeat, prev, fork)
    alias = None
    if fork:
        out, alias = out
    if _ConvBn.last_offer is not None:
        out._io_offer, _ConvBn.last_offer = _ConvBn.last_offer, None
    return (out, alias) if fork else out


# ---- pooling / heads --------------------------------------------------------------------------------------------------
class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x, "x")
        N, H, W_, Cc = x.shape
        Ho, Wo = (H + 1) // 2, (W_ + 1) // 2
        out = torch.empty((N, Ho, Wo, Cc), device=x.device, dtype=x.dtype)
        idx = torch.empty((N, Ho, Wo, Cc // 4), dtype=torch.int32, device=x.device)
        _lib.check(_L().io_maxpool_fwd_dt(_p(x), N, H, W_, Cc, _p(out), _p(idx), _dt(x), _st()), "io_maxpool_fwd_dt")
        ctx.save_for_backward(idx)
        ctx.geom = (N, H, W_, Cc)
        return out

    @staticmethod
    def backward(ctx, dy):
        idx, = ctx.saved_tensors
        N, H, W_, Cc = ctx.geom
        dx = torch.empty((N, H, W_, Cc), device=dy.device, dtype=dy.dtype)
        _lib.check(_L().io_maxpool_bwd_dt(_p(dy.contiguous()), _p(idx), N, H, W_, Cc, _p(dx), _dt(dy), _st()),
                   "io_maxpool_bwd_dt")
        return dx


def max_pool_3x3s2(x):
    """nn.MaxPool2d(3, 2, 1) (resnet_cls.py:144)."""
    return _MaxPool.apply(x)


class _AvgPoolFc(torch.autograd.Function):
    """AdaptiveAvgPool2d(1) + flatten + nn.Linear (midas_net.py:195-197, 205-207)."""

    @staticmethod
    def forward(ctx, x, w, b):
        _chk(x, "x")
        N, H, W_, Cc = x.shape
        K = w.shape[0]
        pooled = torch.empty((N, Cc), device=x.device, dtype=torch.float32)
        logits = torch.empty((N, K), device=x.device, dtype=torch.float32)
        wd, bd = w.detach().contiguous(), b.detach().contiguous()
        _lib.check(_L().io_avgpool_fc_fwd_dt(_p(x), N, H * W_, Cc, _p(wd), _p(bd), K, None, None, 0, _p(pooled), _p(logits),
                                             _dt(x), _st()), "io_avgpool_fc_fwd_dt")
        ctx.save_for_backward(pooled, wd)
        ctx.geom = (N, H, W_, Cc, K, x.dtype)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        pooled, wd = ctx.saved_tensors
        N, H, W_, Cc, K, xdt = ctx.geom
        dx = torch.empty((N, H, W_, Cc), device=pooled.device, dtype=xdt)
        dw, db = torch.empty((K, Cc), device=pooled.device), torch.empty(K, device=pooled.device)
        _lib.check(_L().io_avgpool_fc_bwd_dt(_p(dlogits.float().contiguous()), _p(pooled), N, H * W_, Cc, _p(wd), K, None, 0,
                                             None, _p(dx), _p(dw), _p(db), None, None, _dt(dx), _st()),
                   "io_avgpool_fc_bwd_dt")
        return dx, dw, db


def avgpool_fc(x, w, b):
    return _AvgPoolFc.apply(x, w, b)


# ---- decoder pieces -----------------------------------------------------------------------------------------------------
class _Upsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, align):
        _chk(x, "x")
        N, H, W_, Cc = x.shape
        out = torch.empty((N, 2 * H, 2 * W_, Cc), device=x.device, dtype=x.dtype)
        _lib.check(_L().io_upsample2x_bilinear_fwd(_p(x), N, H, W_, Cc, int(align), _p(out), _dt(x), _st()),
                   "io_upsample2x_fwd")
        ctx.geom = (N, H, W_, Cc, int(align))
        return out

    @staticmethod
    def backward(ctx, dy):
        N, H, W_, Cc, align = ctx.geom
        dx = torch.empty((N, H, W_, Cc), device=dy.device, dtype=dy.dtype)
        _lib.check(_L().io_upsample2x_bilinear_bwd(_p(dy.contiguous()), N, H, W_, Cc, align, _p(dx), _dt(dy), _st()),
                   "io_upsample2x_bwd")
        return dx, None


def upsample2x(x, align_corners):
    """nn.functional.interpolate(scale_factor=2, mode='bilinear', align_corners=...) (midas/blocks.py:111-113, 186-188)."""
    return _Upsample.apply(x, align_corners)


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, relu):
        _chk(x, "x")
        Cc = x.shape[-1]
        M = x.numel() // Cc
        out = torch.empty_like(x)
        bd = None
        if bias is not None:
            bd = bias.detach()
            if bd.numel() != Cc:                       # padded output channels carry no bias
                bd = torch.cat([bd, torch.zeros(Cc - bd.numel(), device=x.device)])
        _lib.check(_L().io_bias_act(_p(x), _p(bd), M, Cc, int(relu), _p(out), _dt(x), _st()), "io_bias_act")
        ctx.save_for_backward(out)
        ctx.cfg = (M, Cc, bool(relu), None if bias is None else bias.numel())
        return out

    @staticmethod
    def backward(ctx, dy):
        out, = ctx.saved_tensors
        M, Cc, relu, nbias = ctx.cfg
        L = _L()
        dy = dy.contiguous()
        dx = dy
        if relu:
            dx = torch.empty_like(dy)
            _lib.check(L.io_relu_bwd(_p(dy), _p(out), dy.numel(), _p(dx), _dt(dy), _st()), "io_relu_bwd")
        db = None
        if nbias is not None:
            npart = int(L.io_colsum_partial_floats(M, Cc))
            part = torch.empty(npart, device=dy.device, dtype=torch.float32)
            dbf = torch.empty(Cc, device=dy.device, dtype=torch.float32)
            _lib.check(L.io_colsum(_p(dx), M, Cc, _p(dbf), _p(part), npart, _dt(dx), _st()), "io_colsum")
            db = dbf[:nbias].clone()
        return dx, db, None


def bias_act(x, bias=None, relu=False):
    """[relu](x + bias): the bias of a biased nn.Conv2d and / or nn.ReLU (midas/blocks.py:133-160)."""
    return _BiasAct.apply(x, bias, relu)


def relu(x):
    return _BiasAct.apply(x, None, True)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _chk(a, "a")
        _chk(b, "b")
        out = torch.empty_like(a)
        _lib.check(_L().io_add(_p(a), _p(b), a.numel(), _p(out), _dt(a), _st()), "io_add")
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _Add.apply(a, b)


class _AddShared(torch.autograd.Function):
    """a + b where b is a tensor OTHER STREAMS consume too (the encoder features injected into an order branch that runs on a
    side stream, midas/midas_net.py:190-197).  The backward hands b its OWN copy of the gradient.  With `return dy, dy`
    (what _Add does, and torch's own AddBackward) the branch's next backward node and the node behind b -- on the main
    stream, where the contributions of both order branches meet in one autograd input buffer -- hold the SAME tensor; the
    engine accumulates the second contribution IN PLACE as soon as nobody else references the first (use_count == 1), and
    kernels of the branch that are still queued on the side stream hold no reference: they then read a gradient that
    already contains the other branch's.  Found in round 5 (two processes sharing a GPU: one branch's layers below the
    injection point and the encoder stage under it got gradients 10 % off, run to run; tools/depth_eager_race.py)."""

    @staticmethod
    def forward(ctx, a, b):
        _chk(a, "a")
        _chk(b, "b")
        out = torch.empty_like(a)
        _lib.check(_L().io_add(_p(a), _p(b), a.numel(), _p(out), _dt(a), _st()), "io_add")
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy, dy.clone()


def add_shared(a, b):
    return _AddShared.apply(a, b)


class _Head1(torch.autograd.Function):
    """nn.Conv2d(C, 1, 1) [+ nn.ReLU] on an input that may carry padding channels (midas_net.py:139-140)."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        _chk(x, "x")
        N, H, W_, pitch = x.shape
        Cc = w.numel()
        M = N * H * W_
        out = torch.empty((N, H, W_), device=x.device, dtype=torch.float32)
        wd, bd = w.detach().reshape(-1).contiguous(), b.detach().reshape(-1).contiguous()
        _lib.check(_L().io_head1_fwd(_p(x), M, pitch, Cc, _p(wd), _p(bd), int(relu), _p(out), _dt(x), _st()), "io_head1_fwd")
        ctx.save_for_backward(x, out, wd)
        ctx.cfg = (M, pitch, Cc, int(relu), tuple(w.shape))
        return out

    @staticmethod
    def backward(ctx, dy):
        x, out, wd = ctx.saved_tensors
        M, pitch, Cc, relu, wshape = ctx.cfg
        L = _L()
        npart = int(L.io_colsum_partial_floats(M, Cc))
        part = torch.empty(npart, device=x.device, dtype=torch.float32)
        dx = torch.empty_like(x)
        dw, db = torch.empty(Cc, device=x.device), torch.empty(1, device=x.device)
        _lib.check(L.io_head1_bwd(_p(dy.float().contiguous()), _p(out), _p(x), M, pitch, Cc, _p(wd), relu, _p(dx), _p(dw),
                                  _p(db), _p(part), npart, _dt(x), _st()), "io_head1_bwd")
        return dx, dw.view(wshape), db, None


def head1(x, w, b, relu):
    return _Head1.apply(x, w, b, relu)


class _SmoothLoss(torch.autograd.Function):
    """Edge-aware smoothness loss (models/supervised_order.py:214-235) of a disparity map disp[B,1,H,W] against the image
    img[B,3,H,W]: three launches forward, one backward (io_smooth_loss_*), the incoming scalar gradient read on the device.
    ``times``: the loss evaluated that many times on the same map (pair mode: both mask orders share the disparity)."""

    @staticmethod
    def forward(ctx, disp, img, times):
        if not (disp.is_cuda and disp.dtype == torch.float32 and img.dtype == torch.float32):
            raise RuntimeError("instaorder_amd.ops.smooth_loss: fp32 tensors on the GPU (no CPU fallback)")
        B, _, H, W_ = disp.shape
        d, im = disp.detach().contiguous(), img.detach().contiguous()
        L = _L()
        nws = int(L.io_smooth_loss_workspace_floats(B, H, W_))
        ws = torch.empty(nws, device=d.device, dtype=torch.float32)
        g = torch.empty((B, H, W_), device=d.device, dtype=torch.float32)
        loss = torch.empty((), device=d.device, dtype=torch.float32)
        _lib.check(L.io_smooth_loss_fwd(_p(d), _p(im), B, H, W_, float(times), _p(loss), _p(g), _p(ws), nws, _st()),
                   "io_smooth_loss_fwd")
        ctx.save_for_backward(g, ws)
        ctx.cfg = (B, H, W_, float(times), tuple(disp.shape))
        return loss

    @staticmethod
    def backward(ctx, dl):
        g, ws = ctx.saved_tensors
        B, H, W_, times, shape = ctx.cfg
        dd = torch.empty((B, H, W_), device=g.device, dtype=torch.float32)
        _lib.check(_L().io_smooth_loss_bwd(_p(g), _p(ws), _p(dl.float().contiguous()), times, B, H, W_, 0, _p(dd), _st()),
                   "io_smooth_loss_bwd")
        return dd.view(shape), None, None


def smooth_loss(disp, img, times=1):
    return _SmoothLoss.apply(disp, img, times)


def disp_order_count(disp1, disp2, modal1, modal2, depth_order1, is_overlap, le_order=0, scale=1.0):
    """supervised_order.py:152-173 for the whole batch in three launches (io_disp_order_count): erosion, masked max / min,
    comparisons and counts on the device, no host round trip; returns a 0-dim tensor = scale * total / (H * W)."""
    B, _, H, W_ = disp1.shape
    dev = disp1.device
    out = torch.empty((), device=dev, dtype=torch.float32)
    nws = int(_L().io_disp_order_workspace_floats(B, H, W_))
    ws = torch.empty(max(nws, 1), device=dev, dtype=torch.float32)
    f = lambda t: t.detach().float().contiguous()       # noqa: E731
    _lib.check(_L().io_disp_order_count(_p(f(disp1)), _p(f(disp2)), _p(f(modal1)), _p(f(modal2)),
                                        _p(depth_order1.long().contiguous()), _p(is_overlap.long().contiguous()), B, H, W_,
                                        int(le_order), float(scale), _p(out), _p(ws), nws, _st()), "io_disp_order_count")
    return out
