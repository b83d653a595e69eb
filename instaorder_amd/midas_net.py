"""MI355X-native InstaDepthNet_od / InstaDepthNet_d -- host-side mirror of midas/midas_net.py:14-212 and
midas/blocks.py:5-195 (BASELINE configs[4]).

Same constructor surface, same ``forward(img, mask1, mask2) -> (disp[B,H,W], depth_order[B,3], occ_order[B,2] | None)``
and the same ``state_dict`` keys / OIHW shapes as the reference modules (including the duplicate keys the reference
gets from re-using ``conv1`` / ``bn1`` inside ``layer1 = nn.Sequential(conv1, bn1, relu, maxpool, layer1)``), so
checkpoints interchange.  The arithmetic is ``instaorder_amd.ops`` (HIP kernels, NHWC fp32) op by op: the ResNeXt-101
32x8d encoder with its grouped 3x3 convolutions, the RefineNet decoder, and the two ResNet-50 order branches with
feature injection.  The encoder of ``torch.hub``'s ``resnext101_32x8d_wsl`` is torchvision's ResNeXt-101 32x8d, i.e.
the reference's own ``resnext101_32x8d`` (resnet_cls.py:309-320).

Reference quirk kept on purpose: ``ResidualConvUnit`` applies an IN-PLACE ReLU to its input (midas/blocks.py:151), so
its skip connection adds relu(x), not x.
"""
import math

import torch
import torch.nn as nn

from . import common_utils, ops

import os

__all__ = ["InstaDepthNet_od", "InstaDepthNet_d"]


class Conv2d(nn.Module):
    """Parameter holder with nn.Conv2d's names, shapes and default init; computation in ops.conv2d (+ bias)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=False, groups=1, co_pad=False):
        super(Conv2d, self).__init__()
        self.stride, self.padding, self.groups, self.co_pad = stride, padding, groups, co_pad
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(cin // groups * k * k)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, relu=False):
        y = ops.conv2d(x, self.weight, self.stride, self.padding, self.groups, self.co_pad)
        if self.bias is not None or relu:
            y = ops.bias_act(y, self.bias, relu)
        return y


class _BnMode(object):
    """How the BatchNorm layers account the current call (see ops.batch_norm): `groups` independent statistic groups
    (the two mask orders batched into one call of an order branch) or `repeat` identical calls folded into one (the
    shared encoder / decoder, which the reference runs once per mask order on the same image)."""
    groups = 1
    repeat = 1

    def __init__(self, groups=1, repeat=1):
        self.g, self.r = groups, repeat

    def __enter__(self):
        self.prev = (_BnMode.groups, _BnMode.repeat)
        _BnMode.groups, _BnMode.repeat = self.g, self.r

    def __exit__(self, *a):
        _BnMode.groups, _BnMode.repeat = self.prev


class _Counters(object):
    """num_batches_tracked of every BatchNorm layer a forward pass touches advances in ONE multi-tensor launch at the
    end of the pass instead of one tiny launch per layer (there are 210 of them)."""
    pending = None

    def __enter__(self):
        _Counters.pending = []

    def __exit__(self, *a):
        todo, _Counters.pending = _Counters.pending, None
        for k in sorted(set(k for _, k in todo)):
            torch._foreach_add_([t for t, kk in todo if kk == k], k)

    @staticmethod
    def bump(t, k):
        if _Counters.pending is None:
            t += k
        else:
            _Counters.pending.append((t, k))


class BatchNorm2d(nn.Module):
    def __init__(self, c):
        super(BatchNorm2d, self).__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long))

    def forward(self, x, relu=False, identity=None):
        if self.training:
            _Counters.bump(self.num_batches_tracked, _BnMode.groups * _BnMode.repeat)
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, self.training, relu,
                              identity, _BnMode.groups, _BnMode.repeat)

    def after(self, conv, x, relu=False, identity=None, sole=False, fork=False):
        """bn(conv(x)) (+ identity) (+ ReLU) as one fused node (statistics in the convolution's epilogue).  sole: this is
        the only use of x (ops.conv_bn then runs the backward of x's own BatchNorm inside its data-gradient launch)."""
        if self.training:
            _Counters.bump(self.num_batches_tracked, _BnMode.groups * _BnMode.repeat)
        return ops.conv_bn(x, conv.weight, self.weight, self.bias, self.running_mean, self.running_var, conv.stride,
                           conv.padding, conv.groups, self.training, relu, identity, _BnMode.groups, _BnMode.repeat,
                           sole=sole, fork=fork)


class _Fn(nn.Module):
    """Parameter-free stage of an nn.Sequential (keeps the reference's child indices)."""

    def __init__(self, fn):
        super(_Fn, self).__init__()
        self.fn = fn

    def forward(self, x):
        return self.fn(x)


_FORK = os.environ.get("IO_DEPTH_FORK", "1") != "0"      # (0: the residual gradient summed by the autograd engine -- A/B runs)


class Bottleneck(nn.Module):
    """resnet_cls.py:75-116 with groups / base_width."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64):
        super(Bottleneck, self).__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = Conv2d(inplanes, width, 1)
        self.bn1 = BatchNorm2d(width)
        self.conv2 = Conv2d(width, width, 3, stride, 1, groups=groups)
        self.bn2 = BatchNorm2d(width)
        self.conv3 = Conv2d(width, planes * 4, 1)
        self.bn3 = BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        # fork: the block input feeds conv1 AND the residual path; the alias conv1 hands back carries the residual path, so its
        # gradient is added inside conv1's data-gradient launch (ops.conv_bn) instead of by an accumulation kernel
        out = self.bn1.after(self.conv1, x, relu=True, fork=_FORK)
        if isinstance(out, tuple):
            out, x = out
        out = self.bn2.after(self.conv2, out, relu=True, sole=True)          # relu(bn1(.)) feeds conv2 only
        identity = x
        if self.downsample is not None:
            identity = self.downsample[1].after(self.downsample[0], x)
        return self.bn3.after(self.conv3, out, relu=True, identity=identity, sole=True)   # relu(bn2(.)) feeds conv3 only


def _make_layer(inplanes, planes, blocks, stride, groups, base_width):
    downsample = None
    if stride != 1 or inplanes != planes * 4:
        downsample = nn.Sequential(Conv2d(inplanes, planes * 4, 1, stride), BatchNorm2d(planes * 4))
    layers = [Bottleneck(inplanes, planes, stride, downsample, groups, base_width)]
    for _ in range(1, blocks):
        layers.append(Bottleneck(planes * 4, planes, 1, None, groups, base_width))
    return nn.Sequential(*layers)


class _Trunk(nn.Module):
    """The four stages of a (grouped) ResNet as midas/blocks.py:71-82 arranges them:
    layer1 = Sequential(conv1, bn1, relu, maxpool, layer1), layer2..4.  ``conv1`` / ``bn1`` / ``fc`` are also
    registered at top level when ``keep_stem_names`` (the reference's do_net / oo_net / gdo_net keep them)."""

    def __init__(self, in_channels, layers, groups, base_width, keep_stem_names, num_classes=None):
        super(_Trunk, self).__init__()
        conv1 = Conv2d(in_channels, 64, 7, 2, 3)
        bn1 = BatchNorm2d(64)
        if keep_stem_names:
            self.conv1, self.bn1 = conv1, bn1
        l1 = _make_layer(64, 64, layers[0], 1, groups, base_width)
        self.layer1 = nn.Sequential(conv1, bn1, _Fn(lambda x: x), _Fn(ops.max_pool_3x3s2), l1)
        self.layer2 = _make_layer(256, 128, layers[1], 2, groups, base_width)
        self.layer3 = _make_layer(512, 256, layers[2], 2, groups, base_width)
        self.layer4 = _make_layer(1024, 512, layers[3], 2, groups, base_width)
        if num_classes is not None:
            self.fc = nn.Linear(2048, num_classes)      # present (unused) in the reference's order branches
        for m in self.modules():                          # resnet_cls.py:162-167
            if isinstance(m, Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def run_layer1(self, x8):
        seq = self.layer1
        y = seq[1].after(seq[0], x8, relu=True)
        return seq[4](seq[3](y))


class ResidualConvUnit(nn.Module):
    """midas/blocks.py:121-160 (skip connection carries relu(x): the reference's ReLU is in-place)."""

    def __init__(self, features):
        super(ResidualConvUnit, self).__init__()
        self.conv1 = Conv2d(features, features, 3, 1, 1, bias=True)
        self.conv2 = Conv2d(features, features, 3, 1, 1, bias=True)

    def forward(self, x):
        r = ops.relu(x)
        out = self.conv1(r, relu=True)
        out = self.conv2(out)
        return ops.add(out, r)


class FeatureFusionBlock(nn.Module):
    """midas/blocks.py:163-195."""

    def __init__(self, features):
        super(FeatureFusionBlock, self).__init__()
        self.resConfUnit1 = ResidualConvUnit(features)
        self.resConfUnit2 = ResidualConvUnit(features)

    def forward(self, *xs):
        output = xs[0]
        if len(xs) == 2:
            output = ops.add(output, self.resConfUnit1(xs[1]))
        output = self.resConfUnit2(output)
        return ops.upsample2x(output, True)


class _Scratch(nn.Module):
    pass


def _device_identity():
    """What names the physical GPU this rank computes on: host + device UUID (PCI address where the build has no UUID).
    Index-free on purpose: with HIP_VISIBLE_DEVICES isolation every rank calls its GPU "device 0"."""
    import socket
    host = socket.gethostname()
    if not torch.cuda.is_available():
        return (host, "cpu")
    prop = torch.cuda.get_device_properties(torch.cuda.current_device())
    uid = getattr(prop, "uuid", None)
    if uid is None:
        uid = tuple(getattr(prop, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        if all(v is None for v in uid):
            uid = ("index", torch.cuda.current_device(), os.environ.get("HIP_VISIBLE_DEVICES"))
    return (host, str(uid))


def _ranks_share_a_device():
    """Do two ranks of the process group compute on the SAME physical GPU?  Decided from what the ranks themselves report
    (one all-gather of (hostname, device identity)), not from world_size vs device_count(): that comparison is also true for a
    multi-node job (16 ranks, 8 GPUs per node) and for launchers that isolate one GPU per process with HIP_VISIBLE_DEVICES
    (device_count() == 1) -- exactly the one-process-per-GPU deployment the side streams are for.  COLLECTIVE: every rank
    must call it at the same point (it is called from the first forward of the data-parallel step)."""
    import torch.distributed as dist
    try:
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return False
        mine = _device_identity()
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, mine)
        return sum(1 for e in everyone if tuple(e) == tuple(mine)) > 1
    except Exception:          # noqa: BLE001 -- an undecidable topology keeps the safe single-stream form
        return True


class _InstaDepthBase(nn.Module):
    def __init__(self, path=None, features=256, non_negative=True):
        super(_InstaDepthBase, self).__init__()
        # encoder: ResNeXt-101 32x8d (midas/blocks.py:5-8, 85-87)
        self.pretrained = _Trunk(3, [3, 4, 23, 3], 32, 8, keep_stem_names=False)
        self.scratch = _Scratch()
        for i, c in enumerate([256, 512, 1024, 2048]):                       # midas/blocks.py:19-45
            setattr(self.scratch, "layer%d_rn" % (i + 1), Conv2d(c, features, 3, 1, 1))
        for i in (4, 3, 2, 1):
            setattr(self.scratch, "refinenet%d" % i, FeatureFusionBlock(features))
        self.non_negative = non_negative
        # Side streams (decoder || order branches, see _fork_join) are for a process that has its GPU to itself -- the
        # deployment: one process per GPU.  When several ranks SHARE a device (the two-rank gloo runs on a one-GPU box) the
        # streams of the processes are time-sliced against each other and the staged step ran 0.27 .. 8.7 s per step instead
        # of 0.27 (profiles/r05_bench_config4_*_2ranks_gloo_one_gpu.json was taken with this rule).  IO_DEPTH_STREAMS: 0 =
        # one stream, force = side streams whatever the ranks report, anything else = decide at the FIRST forward (the
        # process group may be initialised after the model is built) from the ranks' device identities.  Read per model, so
        # a test can build both forms in one process; assigning True / False to multi_stream overrides the rule.
        mode = os.environ.get("IO_DEPTH_STREAMS", "1")
        self._multi_stream = False if mode == "0" else (True if mode == "force" else None)      # None: not decided yet
        self.dtype = "fp32"       # 'bf16': activations / GEMM operands in bf16 (set by SingleStageModel from params['dtype'])
        self.scratch.output_conv = nn.Sequential(                              # midas_net.py:134-141
            Conv2d(features, 128, 3, 1, 1, bias=True),
            _Fn(lambda x: ops.upsample2x(x, False)),
            Conv2d(128, 32, 3, 1, 1, bias=True, co_pad=True),
            _Fn(lambda x: x),
            Conv2d(32, 1, 1, 1, 0, bias=True),
            _Fn(lambda x: x),
        )
        if path:
            self.load(path)

    @property
    def multi_stream(self):
        if self._multi_stream is None:
            shared = _ranks_share_a_device()
            self._multi_stream = not shared
            if shared:
                import logging
                logging.getLogger("instaorder_amd").warning(
                    "InstaDepthNet: several ranks compute on one device (%s) -- decoder and order branches stay on ONE "
                    "stream (IO_DEPTH_STREAMS=force overrides)", _device_identity()[1])
        return self._multi_stream

    @multi_stream.setter
    def multi_stream(self, on):
        self._multi_stream = bool(on)

    def load(self, path):
        """midas/base_model.py:5-15."""
        parameters = torch.load(path, map_location=torch.device("cpu"))
        if "optimizer" in parameters:
            parameters = parameters["model"]
        self.load_state_dict(parameters, strict=False)

    @staticmethod
    def _cut(t):
        return t.detach().requires_grad_(True)

    def _act_dtype(self):
        return torch.bfloat16 if self.dtype == "bf16" else torch.float32

    def _order_branch(self, net, fc, x8m, l1, l2, l3, side=False):
        # l1..l3 feed both branches and the decoder.  ``side``: this branch runs on a side stream, so the shared operand's
        # gradient must be its own copy (ops._AddShared: three encoder-feature-sized copies per backward); on ONE stream
        # everything is ordered and the plain add (one gradient tensor handed to both inputs) is safe and cheaper.
        add = ops.add_shared if side else ops.add
        f1 = net.run_layer1(x8m)
        f2 = net.layer2(add(f1, l1))
        f3 = net.layer3(add(f2, l2))
        f4 = net.layer4(add(f3, l3))
        return ops.avgpool_fc(f4, fc.weight, fc.bias)

    def _encode(self, img):
        if img.dim() != 4 or img.shape[1] != 3:
            raise ValueError("expected img [B,3,H,W], got %s" % (tuple(img.shape),))
        if not img.is_cuda:
            raise RuntimeError("instaorder_amd: input must be on the GPU; there is no CPU fallback")
        p, s = self.pretrained, self.scratch
        x8 = ops.nhwc_from_nchw(img.float(), pad_to=8, dtype=self._act_dtype())
        # The encoder's stage outputs are where a data-parallel step cuts its backward pass into stages (supervised_order.
        # _DepthBase._staged_steps: heads + order branches + decoder, then encoder layer4, layer3, layer2 + layer1).  With
        # `_stage_cut` every consumer of a stage output -- the next encoder stage, the decoder, the order branches -- reads a
        # detached alias of it (same storage, its own autograd leaf), so each stage is an autograd graph of its own and the
        # gradient a boundary collects is handed to the stage below by the caller; without it this is the plain graph.
        staged = getattr(self, "_stage_cut", False) and torch.is_grad_enabled()
        cut = self._cut if staged else (lambda t: t)
        r1 = p.run_layer1(x8)
        l1 = cut(r1)
        r2 = p.layer2(l1)
        l2 = cut(r2)
        r3 = p.layer3(l2)
        l3 = cut(r3)
        r4 = p.layer4(l3)
        l4 = cut(r4)
        # (stage output, what its consumers read) -- only for the staged caller, which drops it again: a reference held here
        # would keep the step's autograd graph alive into the next step
        self._feats = ((r1, l1), (r2, l2), (r3, l3), (r4, l4)) if staged else None
        return l1, l2, l3, l4

    def _decode(self, l1, l2, l3, l4):
        s = self.scratch
        path4 = s.refinenet4(s.layer4_rn(l4))
        path3 = s.refinenet3(path4, s.layer3_rn(l3))
        path2 = s.refinenet2(path3, s.layer2_rn(l2))
        path1 = s.refinenet1(path2, s.layer1_rn(l1))
        oc = s.output_conv
        y = oc[0](path1)
        y = oc[1](y)
        y = oc[2](y, relu=True)                         # 32 real + 32 zero channels
        return ops.head1(y, oc[4].weight, oc[4].bias, self.non_negative)

    def _encode_decode(self, img):
        l1, l2, l3, l4 = self._encode(img)
        return self._decode(l1, l2, l3, l4), (l1, l2, l3)

    # The decoder and the order branches all hang off the encoder's stage outputs and do not depend on one another; at the
    # 16-pair batch of BASELINE configs[4] their launches (24 x 24 and 12 x 12 maps: 36-150 tiles) fill a fraction of the 256
    # CUs.  IO_DEPTH_STREAMS=1: the branches run on side streams next to the decoder (fork after the encoder, join before
    # the losses); autograd runs each backward node on its forward stream, so the backward pass overlaps the same way, and
    # a hipGraph capture records the fork / join as parallel branches of the graph.  Same kernels, same arithmetic.
    _side = {}           # device index -> side streams (shared by every model on that device)

    def _side_streams(self, n):
        dev = torch.cuda.current_device()
        cur = _InstaDepthBase._side.get(dev)
        if cur is None or len(cur) < n:
            cur = _InstaDepthBase._side[dev] = [torch.cuda.Stream(device=dev) for _ in range(n)]
        return cur[:n]

    def _fork_join(self, jobs, main_job, shared=()):
        """jobs: callables for the side streams; main_job runs on the current stream.  Returns ([side results], main result).
        ``shared``: tensors made on the current stream that the side jobs read.  The caching allocator hands a freed block
        back to the stream it was allocated on at once, and the backward nodes of a side job may still be waiting in their
        queue when the last reference to such a tensor goes (the packed masks: released by the last stem's node, while the
        encoder's backward is already allocating on the main stream) -- record_stream makes the block wait for them."""
        main = torch.cuda.current_stream()
        sides = self._side_streams(len(jobs))
        for t in shared:
            for st in sides:
                t.record_stream(st)
        ev = torch.cuda.Event()
        ev.record(main)
        out = []
        try:
            for st, job in zip(sides, jobs):
                st.wait_event(ev)
                with torch.cuda.stream(st):
                    out.append(job())
            res = main_job()
        finally:
            # always joined, also when a job raised: a side stream left un-joined would leave a stream capture open-ended
            for st in sides:
                main.wait_stream(st)
        # only a forward that recorded a tape has a backward whose side-stream tail must be joined later
        self._forked = len(sides) if torch.is_grad_enabled() else 0
        return out, res

    _forked = 0

    def join_side_streams(self):
        """After the backward pass of a forked forward: the current stream waits for the side streams.  Autograd runs
        each backward node on its forward stream and orders a node's OUTPUTS before their consumers, but the last node of
        a branch (its stem convolution: the masks need no gradient) has none when ops.WeightPlan is active -- its filter
        gradient goes into the plan's buffer as a side effect -- so nothing in the tape orders that kernel before what the
        caller enqueues next (WeightPlan.unpack_grads, the optimiser).  Legal under hipGraph capture: the side streams
        joined this capture at the fork."""
        if self._forked:
            main = torch.cuda.current_stream()
            for st in (_InstaDepthBase._side.get(torch.cuda.current_device()) or [])[:self._forked]:
                main.wait_stream(st)
            self._forked = 0


def _pair_inputs(mask1, mask2, feats, dtype):
    """Both mask orders as one 2B batch: rows [0,B) = (mask1, mask2), rows [B,2B) = (mask2, mask1); the injected
    encoder features are the same for both."""
    m = torch.cat([torch.cat([mask1, mask2], 1), torch.cat([mask2, mask1], 1)], 0).float()
    return ops.nhwc_from_nchw(m, pad_to=8, dtype=dtype), tuple(torch.cat([f, f], 0) for f in feats)


class InstaDepthNet_od(_InstaDepthBase):
    """midas_net.py:116-212: disparity + depth-order head + occlusion-order head."""

    def __init__(self, path=None, features=256, depth_num_classes=3, occ_num_classes=2, non_negative=True):
        super(InstaDepthNet_od, self).__init__(path, features, non_negative)
        self.do_net = _Trunk(2, [3, 4, 6, 3], 1, 64, keep_stem_names=True, num_classes=depth_num_classes)
        self.depth_fc = nn.Linear(2048, depth_num_classes)
        self.oo_net = _Trunk(2, [3, 4, 6, 3], 1, 64, keep_stem_names=True, num_classes=occ_num_classes)
        self.occ_fc = nn.Linear(2048, occ_num_classes)
        common_utils.init_weights(self.do_net, init_type="xavier")
        common_utils.init_weights(self.oo_net, init_type="xavier")

    def forward(self, img, mask1, mask2):
        with _Counters():
            disp, (l1, l2, l3) = self._encode_decode(img)
            x8m = ops.nhwc_from_nchw(torch.cat([mask1, mask2], 1).float(), pad_to=8, dtype=self._act_dtype())
            depth_order = self._order_branch(self.do_net, self.depth_fc, x8m, l1, l2, l3)
            occ_order = self._order_branch(self.oo_net, self.occ_fc, x8m, l1, l2, l3)
        return disp, depth_order, occ_order

    def forward_pair(self, img, mask1, mask2):
        """Both directional calls of supervised_order.py:187-188 / 198-199 at once: the image-only encoder + decoder
        run ONCE (their result is the same for both mask orders), the order branches see the two orders as one 2B
        batch with per-order BatchNorm statistics.  Returns (disp[B,H,W], depth[2B,3], occ[2B,2]), rows [0,B) =
        call (mask1, mask2), rows [B,2B) = call (mask2, mask1).  Same values, gradients and running statistics as the
        two separate calls."""
        with _Counters():
            if self.multi_stream and img.is_cuda:
                with _BnMode(repeat=2):
                    e1, e2, e3, e4 = self._encode(img)
                x8m, (l1, l2, l3) = _pair_inputs(mask1, mask2, (e1, e2, e3), self._act_dtype())

                def branch(net, fc):
                    def run():
                        with _BnMode(groups=2):
                            return self._order_branch(net, fc, x8m, l1, l2, l3, side=True)
                    return run

                def decode():
                    with _BnMode(repeat=2):
                        return self._decode(e1, e2, e3, e4)
                (depth_order, occ_order), disp = self._fork_join([branch(self.do_net, self.depth_fc),
                                                                  branch(self.oo_net, self.occ_fc)], decode,
                                                                 shared=(x8m, l1, l2, l3))
                return disp, depth_order, occ_order
            with _BnMode(repeat=2):
                disp, feats = self._encode_decode(img)
            x8m, (l1, l2, l3) = _pair_inputs(mask1, mask2, feats, self._act_dtype())
            with _BnMode(groups=2):
                depth_order = self._order_branch(self.do_net, self.depth_fc, x8m, l1, l2, l3)
                occ_order = self._order_branch(self.oo_net, self.occ_fc, x8m, l1, l2, l3)
        return disp, depth_order, occ_order


class InstaDepthNet_d(_InstaDepthBase):
    """midas_net.py:14-113: disparity + geometric depth-order head."""

    def __init__(self, path=None, features=256, depth_num_classes=3, occ_num_classes=2, non_negative=True):
        super(InstaDepthNet_d, self).__init__(path, features, non_negative)
        self.gdo_net = _Trunk(2, [3, 4, 6, 3], 1, 64, keep_stem_names=True, num_classes=3)
        self.fc = nn.Linear(2048, depth_num_classes)
        common_utils.init_weights(self.gdo_net, init_type="xavier")

    def forward(self, img, mask1, mask2):
        with _Counters():
            disp, (l1, l2, l3) = self._encode_decode(img)
            x8m = ops.nhwc_from_nchw(torch.cat([mask1, mask2], 1).float(), pad_to=8, dtype=self._act_dtype())
            depth_order = self._order_branch(self.gdo_net, self.fc, x8m, l1, l2, l3)
        return disp, depth_order, None

    def forward_pair(self, img, mask1, mask2):
        """See InstaDepthNet_od.forward_pair."""
        with _Counters():
            if self.multi_stream and img.is_cuda:          # the order branch on a side stream next to the decoder (see _fork_join)
                with _BnMode(repeat=2):
                    e1, e2, e3, e4 = self._encode(img)
                x8m, (l1, l2, l3) = _pair_inputs(mask1, mask2, (e1, e2, e3), self._act_dtype())

                def branch():
                    with _BnMode(groups=2):
                        return self._order_branch(self.gdo_net, self.fc, x8m, l1, l2, l3, side=True)

                def decode():
                    with _BnMode(repeat=2):
                        return self._decode(e1, e2, e3, e4)
                (depth_order,), disp = self._fork_join([branch], decode, shared=(x8m, l1, l2, l3))
                return disp, depth_order, None
            with _BnMode(repeat=2):
                disp, feats = self._encode_decode(img)
            x8m, (l1, l2, l3) = _pair_inputs(mask1, mask2, feats, self._act_dtype())
            with _BnMode(groups=2):
                depth_order = self._order_branch(self.gdo_net, self.fc, x8m, l1, l2, l3)
        return disp, depth_order, None


class MidasNet(_InstaDepthBase):
    """midas_net.py:215-277: MiDaS v2.1 alone -- ResNeXt-101 32x8d encoder + RefineNet decoder + output head;
    ``forward(x[B,3,H,W])`` -> disparity [B,H,W].  Same ``pretrained.*`` / ``scratch.*`` state_dict keys as the reference
    (``model-f6b98070.pt`` loads through ``load``); the method 'midas_pretrained' of tools/test.py:139-146 and
    inference.py:583-590."""

    def __init__(self, path=None, features=256, num_classes=3, non_negative=True):
        super(MidasNet, self).__init__(path, features, non_negative)

    def forward(self, x):
        with _Counters():
            disp, _ = self._encode_decode(x)
        return disp
