"""Order-prediction model wrappers -- host-side mirror of models/supervised_order.py.

``InstaOrderNet_o`` (:496-548), ``InstaOrderNet_od`` (:18-95), ``InstaOrderNet_d`` (:370-438) and
``OrderNet`` (:442-493) keep the reference API consumed by trainer.py / tools/test.py / inference.py:
``set_input(*batch)``, ``step()``, ``forward_only()``, ``switch_to()``, ``load_state()``,
``save_state()``, attributes ``.model .optim .world_size``.

What differs is how a step executes.  The reference runs the backbone twice (mask order a,b then
b,a), builds the loss with torch ops and lets autograd walk back.  Here both mask orders are packed
into ONE batch of 2B samples that the engine normalises as two independent BatchNorm groups
(identical statistics and running-stat updates to two sequential calls), the loss + its logit
gradient come from one HIP kernel, the backward is ``io_net_backward``, the gradient exchange is one
flat all-reduce and the update one fused SGD launch -- with no host synchronisation in between.
"""
import os

import torch
import torch.distributed as dist

from . import distributed_utils, engine
from .single_stage_model import SingleStageModel


def _dev(t, dtype=None):
    t = t.cuda(non_blocking=True)
    return t.to(dtype) if dtype is not None and t.dtype != dtype else t



def _force_overlap_requested():
    """IO_COMM_OVERLAP=force: the staged data-parallel step on ONE rank.  It issues the backend's all-reduce between the
    stage graphs, so it needs an initialised process group -- asked for without one, fail HERE with a clear message
    instead of deep inside the first step, after the stage graphs have been enqueued."""
    if os.environ.get("IO_COMM_OVERLAP", "1") != "force":
        return False
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("IO_COMM_OVERLAP=force runs the staged gradient exchange (dist.all_reduce per backward stage): "
                           "initialise a process group first (torch.distributed.init_process_group, world_size >= 1) or "
                           "unset IO_COMM_OVERLAP")
    return True


class _OrderBase(SingleStageModel):
    """Shared mechanics of the four wrappers; subclasses define labels and loss configuration."""

    KOCC = 0          # occlusion logits (sigmoid + BCE)
    KDEP = 0          # depth / order-class logits (softmax + CE on the probabilities)

    def __init__(self, params, load_pretrain=None, dist_model=False):
        super(_OrderBase, self).__init__(params, dist_model)
        self.params = params
        self.use_rgb = params.get("use_rgb", False)
        want = 5 if self.use_rgb else 2         # (mask_a, mask_b[, R, G, B])  supervised_order.py:521-526
        if self.net.in_channels != want:
            raise ValueError("use_rgb=%s needs backbone_param.in_channels=%d, got %d"
                             % (self.use_rgb, want, self.net.in_channels))
        self._x8 = None
        # hipGraph replay of forward + loss + backward (the ~800 launches of a step): pays off when the step
        # is launch-bound (small per-GPU batches, e.g. the reference recipe's 32 pairs/GPU); IO_NO_GRAPH=1 disables
        self._use_graph = os.environ.get("IO_NO_GRAPH", "0") != "1"
        self._graph = None
        self._graph_ws = None
        self._dp_keep = None
        self._graph_key = None
        self._seen_key = None
        self._static = {}
        # world_size > 1: the gradient all-reduce in stage buckets under the backward pass (IO_COMM_OVERLAP=0: one flat
        # all-reduce after it, the round-1/2 behaviour)
        self._overlap_comm = os.environ.get("IO_COMM_OVERLAP", "1") != "0"
        # IO_COMM_OVERLAP=force: the staged path on ONE rank too (an initialised process group of size 1) -- the per-stage
        # hipGraphs with the backend's asynchronous all-reduce in between, exercised where a second GPU is not to be had
        self._force_overlap = _force_overlap_requested()
        self._buckets = None
        self._dp_graphs = None          # world_size > 1: one hipGraph per backward stage (the collectives sit between them)
        self._dp_key = None
        if load_pretrain is not None:
            self.load_pretrain(load_pretrain)

    # -- inputs ------------------------------------------------------------------------------------
    def _static_copy(self, name, t):
        """Labels live in persistent device tensors so that a captured step can be replayed."""
        cur = self._static.get(name)
        if cur is None or cur.shape != t.shape or cur.dtype != t.dtype or cur.device != t.device:
            cur = torch.empty_like(t)
            self._static[name] = cur
            self._graph = None
            self._dp_graphs = None
        cur.copy_(t)
        return cur

    def _set_images(self, rgb, modal1, modal2):
        self.rgb = _dev(rgb, torch.float32).contiguous()
        self.modal1 = _dev(modal1, torch.float32).contiguous()
        self.modal2 = _dev(modal2, torch.float32).contiguous()
        B = self.rgb.shape[0]
        if self._x8 is None or self._x8.shape[0] != 2 * B or self._x8.shape[1:3] != self.rgb.shape[2:]:
            self._x8 = None
            self._graph = None
            self._dp_graphs = None
        self._x8 = engine.pack_pair_directions(self.rgb if self.use_rgb else None, self.modal1, self.modal2,
                                               self._x8, dtype=self.net.dtype)
        self.B = B

    # -- loss plumbing -------------------------------------------------------------------------------
    def _loss_args(self, training):
        raise NotImplementedError

    def _loss(self, logits, training, want_grad):
        kw = self._loss_args(training)
        return engine.order_loss(logits, self.B, self.KOCC, self.KDEP, inv_world=1.0 / self.world_size,
                                 want_grad=want_grad, **kw)

    def _logs(self, losses):
        return None

    def _pack_return(self, losses):
        logs = self._logs(losses)
        out = {"loss": losses[0]}
        return (logs, out) if logs is not None else out

    # -- API -------------------------------------------------------------------------------------------
    def forward_only(self, ret_loss=True):
        with torch.no_grad():
            if self.net.training:
                logits = self.net.forward_packed(self._x8, groups=2)
            else:
                logits = self.net.forward_packed(self._x8, groups=1)
            self.last_logits = logits
            if not ret_loss:
                return {}
            losses, _ = self._loss(logits, False, False)
        logs = self._logs(losses)
        return (logs if logs is not None else {}), {"loss": losses[0]}

    def _fwd_loss_bwd(self, N, S):
        net = self.net
        logits, ws = net._run_forward(self._x8, N, S, 2, True)
        losses, dlogits = self._loss(logits, True, True)
        net._run_backward(self._x8, dlogits, N, S, 2, ws)
        return logits, losses, ws

    def _step_overlapped(self, N, S):
        """world_size > 1: forward, loss, then the backward stage by stage with the all-reduce of each stage's slice of
        the flat gradient buffer launched as soon as that stage is enqueued (distributed_utils.GradientBuckets) -- the
        exchange of the reference (average_gradients after loss.backward(), supervised_order.py:545-546) overlapped
        with the rest of the backward pass.  The collectives sit between the stages, so the launch-bound case (small
        per-GPU batches) is served by ONE hipGraph PER STAGE -- forward + loss + stage 0, then stages 1..3 -- replayed with
        the all-reduces launched in between (second step with a shape: capture; the first ran eagerly and warmed every
        kernel); identical kernels and order, so bit-identical to the eager form."""
        net = self.net
        if self._buckets is None:
            self._buckets = distributed_utils.GradientBuckets(net)
        bk = self._buckets
        ns = bk.num_stages
        key = (N, S, self._x8.data_ptr(), net.flat_params.data_ptr())
        graphs_ok = self._use_graph and not engine.prof_active()
        if graphs_ok and self._dp_graphs is not None and self._dp_key == key:
            for s, g in enumerate(self._dp_graphs):
                g.replay()
                bk.launch(s)
            bk.finish()
            logits, losses = (t.clone() for t in self._dp_out)
            return logits, losses
        if graphs_ok and self._seen_key == key:
            # Only the CAPTURE sits in the try-block.  Once a collective of this step has been launched there is no falling
            # back to another path: re-running the forward would advance the BatchNorm running statistics twice and post
            # more all-reduces than the peer ranks do.
            graphs = None
            try:
                torch.cuda.synchronize()
                g0 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g0, capture_error_mode="thread_local"):
                    logits, ws = net._run_forward(self._x8, N, S, 2, True)
                    losses, dlogits = self._loss(logits, True, True)
                    net._run_backward(self._x8, dlogits, N, S, 2, ws, stages=(0, 1))
                graphs = [g0]
                for s in range(1, ns):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=g0.pool(), capture_error_mode="thread_local"):
                        net._run_backward(self._x8, dlogits, N, S, 2, ws, stages=(s, s + 1))
                    graphs.append(g)
            except Exception as ex:   # noqa: BLE001 -- capture unsupported here: stay eager (nothing has run yet)
                graphs = None
                self._use_graph = False
                self._dp_graphs = None
                print("instaorder_amd: per-stage hipGraph capture disabled (%s)" % ex)
            if graphs is not None:
                self._dp_graphs, self._dp_key, self._dp_out, self._dp_keep = graphs, key, (logits, losses), (ws, dlogits)
                for s, g in enumerate(graphs):
                    g.replay()
                    bk.launch(s)
                bk.finish()
                return logits.clone(), losses.clone()
        lent = self._dp_keep[0] if (self._dp_graphs is not None and not net._pool.free) else None
        if lent is not None:          # (see step(): the stage graphs own their arena; lend it to this eager step)
            net._pool.give(lent)
        logits, ws = net._run_forward(self._x8, N, S, 2, True)
        losses, dlogits = self._loss(logits, True, True)
        for s in range(ns):
            net._run_backward(self._x8, dlogits, N, S, 2, ws, stages=(s, s + 1))
            bk.launch(s)
        bk.finish()
        if ws is not lent:
            net._pool.give(ws)
        self._seen_key = key
        return logits, losses

    def step(self):
        net = self.net
        if not net.training:
            raise RuntimeError("step() needs switch_to('train')")
        N = 2 * self.B
        S = self._x8.shape[1]
        key = (N, S, self._x8.data_ptr(), net.flat_params.data_ptr())
        if (self.world_size > 1 or self._force_overlap) and self._overlap_comm:
            logits, losses = self._step_overlapped(N, S)
            self.last_logits = logits
            net.attach_grads()
            self.optim.step()
            return self._pack_return(losses)
        if self._use_graph and self._graph is not None and self._graph_key == key and not engine.prof_active():
            self._graph.replay()
            # the graph writes the same output tensors on every replay: hand out copies, so that a caller who keeps the
            # previous step's loss (to average later) does not see it change under them, as in the eager path
            logits, losses = (t.clone() for t in self._graph_out)
        elif self._use_graph and self._seen_key == key and not engine.prof_active():
            # second step with this shape (the first ran eagerly and warmed every kernel): capture, then run it
            try:
                ws_keep = None
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # thread_local: a data-pipeline thread (datasets.BatchPrefetcher) may allocate / launch on its own
                # stream while this thread captures
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    logits, losses, ws_keep = self._fwd_loss_bwd(N, S)
                self._graph, self._graph_key, self._graph_out, self._graph_ws = g, key, (logits, losses), ws_keep
                g.replay()
                logits, losses = logits.clone(), losses.clone()
            except Exception as ex:   # noqa: BLE001 -- capture unsupported here: stay eager
                self._use_graph = False
                self._graph = None
                print("instaorder_amd: hipGraph capture disabled (%s)" % ex)
                logits, losses, ws = self._fwd_loss_bwd(N, S)
                net._pool.give(ws)
        else:
            # An eager step while a captured graph exists (a profiled pass, a changed batch shape): the graph owns the
            # workspace arena it was captured on -- lend it to this step instead of allocating a second one (55 GiB at the
            # bench batch; nothing of the graph is in flight on this stream)
            lent = self._graph_ws if (self._graph is not None and not net._pool.free) else None
            if lent is not None:
                net._pool.give(lent)
            logits, losses, ws = self._fwd_loss_bwd(N, S)
            if ws is not lent:
                net._pool.give(ws)
            self._seen_key = key
        self.last_logits = logits
        if self.world_size > 1:
            distributed_utils.average_gradients(self.model)
        net.attach_grads()
        self.optim.step()
        return self._pack_return(losses)


def _check_labels(t, lo, hi, what, is_overlap=None):
    """Labels that arrive on the host (the DataLoader's tensors) are validated here; the loss kernel poisons the loss
    with NaN for an out-of-range class id that only shows up on the device.  Only rows that take part in the loss are
    looked at: with ``is_overlap`` given, those whose flag is 0 or 1 -- the reference keeps -1 for overlapped pairs under
    ``remove_depth_overlap`` (datasets/reader.py:363-380) and its boolean masks drop them (supervised_order.py:62-73)."""
    if t is not None and torch.is_tensor(t) and not t.is_cuda and t.numel():
        if is_overlap is not None and torch.is_tensor(is_overlap):
            if is_overlap.is_cuda:        # labels on the host, flags already on the device: bring the flags over
                is_overlap = is_overlap.cpu()
            live = (is_overlap.reshape(-1) == 0) | (is_overlap.reshape(-1) == 1)
            t = t.reshape(-1)[live]
            if not t.numel():
                return
        mn, mx = int(t.min()), int(t.max())
        if mn < lo or mx > hi:
            raise ValueError("%s: values must lie in [%d, %d], got [%d, %d]" % (what, lo, hi, mn, mx))


def _mirror_occ(occ_order):
    return occ_order[:, [1, 0]]


def _mirror_classes(order):
    # 0 <-> 1, anything >= 2 unchanged  (supervised_order.py:39-42, 456-460)
    return torch.where(order >= 2, order, 1 - order)


class InstaOrderNet_o(_OrderBase):
    """Occlusion order: 2 logits, sigmoid + BCELoss (supervised_order.py:496-548)."""
    KOCC = 2

    def set_input(self, rgb=None, modal1=None, modal2=None, occ_order=None):
        self._set_images(rgb, modal1, modal2)
        self.occ_order1 = _dev(occ_order, torch.float32)
        self.occ_order2 = _mirror_occ(self.occ_order1)
        self._occ_t = self._static_copy("occ", torch.cat([self.occ_order1, self.occ_order2], 0))

    def _loss_args(self, training):
        return dict(occ_target=self._occ_t)


class InstaOrderNet_od(_OrderBase):
    """Joint occlusion (2, BCE) + depth (3, CE on softmax, overlap/distinct weighted) heads
    (supervised_order.py:18-95); ``step`` returns ``(losses_to_log, {'loss'})``."""
    KOCC = 2
    KDEP = 3

    def set_input(self, rgb=None, modal1=None, modal2=None, depth_order=None, count=None, is_overlap=None,
                  occ_order=None):
        _check_labels(depth_order, 0, self.KDEP - 1, "depth_order", is_overlap)
        self._set_images(rgb, modal1, modal2)
        self.depth_order1 = _dev(depth_order, torch.long)
        self.depth_order2 = _mirror_classes(self.depth_order1)
        self.count = _dev(count)
        self.is_overlap = self._static_copy("ovl", _dev(is_overlap, torch.long).contiguous())
        self.occ_order1 = _dev(occ_order, torch.float32)
        self.occ_order2 = _mirror_occ(self.occ_order1)
        self._occ_t = self._static_copy("occ", torch.cat([self.occ_order1, self.occ_order2], 0))
        self._dep_t = self._static_copy("dep", torch.cat([self.depth_order1, self.depth_order2], 0))

    def _loss_args(self, training):
        return dict(occ_target=self._occ_t, depth_target=self._dep_t, is_overlap=self.is_overlap,
                    overlap_weight=self.params["overlap_weight"], distinct_weight=self.params["distinct_weight"])

    def _logs(self, losses):
        return {"loss_occ": losses[1], "loss_depth": losses[2]}


class InstaOrderNet_d(_OrderBase):
    """Depth order only: 3 logits, CE on softmax; the TRAINING loss is overlap/distinct weighted but
    forward_only's is the plain mean (supervised_order.py:370-438 -- reference quirk, kept)."""
    KDEP = 3

    def __init__(self, params, load_pretrain=None, dist_model=False):
        super(InstaOrderNet_d, self).__init__(params, load_pretrain, dist_model)
        self.KDEP = int(params["backbone_param"]["num_classes"])

    def set_input(self, rgb=None, modal1=None, modal2=None, depth_order=None, count=None, is_overlap=None):
        _check_labels(depth_order, 0, self.KDEP - 1, "depth_order", is_overlap)
        self._set_images(rgb, modal1, modal2)
        self.depth_order1 = _dev(depth_order, torch.long)
        self.depth_order2 = _mirror_classes(self.depth_order1)
        self.count = _dev(count)
        self.is_overlap = self._static_copy("ovl", _dev(is_overlap, torch.long).contiguous())
        self._dep_t = self._static_copy("dep", torch.cat([self.depth_order1, self.depth_order2], 0))

    def _loss_args(self, training):
        if training:
            return dict(depth_target=self._dep_t, is_overlap=self.is_overlap,
                        overlap_weight=self.params["overlap_weight"],
                        distinct_weight=self.params["distinct_weight"])
        return dict(depth_target=self._dep_t)


class OrderNet(_OrderBase):
    """PCNet-style order classifier: 3 (or 4, OrderNet_ext) classes, CE on softmax
    (supervised_order.py:442-493); labels arrive as class ids in ``occ_order``."""
    KDEP = 3

    def __init__(self, params, load_pretrain=None, dist_model=False):
        super(OrderNet, self).__init__(params, load_pretrain, dist_model)
        self.KDEP = int(params["backbone_param"]["num_classes"])

    def set_input(self, rgb=None, modal1=None, modal2=None, occ_order=None):
        _check_labels(occ_order, 0, self.KDEP - 1, "occ_order (class id)")
        self._set_images(rgb, modal1, modal2)
        self.occ_order1 = _dev(occ_order, torch.long)
        self.occ_order2 = _mirror_classes(self.occ_order1)
        self._dep_t = self._static_copy("dep", torch.cat([self.occ_order1, self.occ_order2], 0))

    def _loss_args(self, training):
        return dict(depth_target=self._dep_t)


# ---- MiDaS-based nets (supervised_order.py:97-367 of the reference) --------------------------------------------------
class _DepthBase(SingleStageModel):
    """Shared machinery of InstaDepthNet_od / InstaDepthNet_d: two directional passes of the MiDaS-based net
    (HIP operators, instaorder_amd.ops), order-head losses through io_order_loss, and the two disparity losses --
    edge-aware smoothness (differentiable) and the disparity-order count (a pure count: no gradient, exactly as in
    the reference) -- as small device-side tensor programs on the [B,1,H,W] disparity maps."""
    HAS_OCC = False

    def __init__(self, params, load_pretrain=None, dist_model=False):
        super(_DepthBase, self).__init__(params, dist_model)
        self.params = params
        self.use_rgb = params.get("use_rgb", False)
        # hipGraph replay of forward + losses + backward + gradient gathering: the MiDaS-based nets are ~3000 launches of
        # mostly small kernels per step, built op by op (instaorder_amd.ops under the autograd tape), so an eager step is
        # bound by the host; the second step of a shape is captured, later ones are replayed (IO_NO_GRAPH=1: always eager)
        self._use_graph = os.environ.get("IO_NO_GRAPH", "0") != "1"
        self._graph = None
        self._graph_key = None
        self._seen_key = None
        self._static = {}
        self._wplan = None          # ops.WeightPlan, built from a recording of the first training step (False: none)
        # world_size > 1: the backward pass in four stages, the all-reduce of each stage's slice of the flat gradient
        # buffer launched as soon as the stage is enqueued (IO_COMM_OVERLAP=0: one flat all-reduce after the backward)
        self._overlap_comm = os.environ.get("IO_COMM_OVERLAP", "1") != "0"
        self._force_overlap = _force_overlap_requested()                            # (as in _OrderBase: one-rank staging)
        self._stages = None
        self._buckets = None
        self._dp_graphs = None
        self._dp_key = None

    # inputs ----------------------------------------------------------------------------------------------------------
    def _keep(self, name, t):
        """Inputs and labels live in persistent device tensors so that a captured step can be replayed on new data."""
        cur = self._static.get(name)
        if cur is None or cur.shape != t.shape or cur.dtype != t.dtype or cur.device != t.device:
            cur = torch.empty_like(t)
            self._static[name] = cur
            self._graph = None
        cur.copy_(t)
        return cur

    def _set_common(self, rgb, modal1, modal2, depth_order, count, is_overlap):
        _check_labels(depth_order, 0, 2, "depth_order", is_overlap)
        self.rgb = self._keep("rgb", _dev(rgb, torch.float32).contiguous())
        self.modal1 = self._keep("modal1", _dev(modal1, torch.float32).contiguous())
        self.modal2 = self._keep("modal2", _dev(modal2, torch.float32).contiguous())
        self.depth_order1 = self._keep("depth_order1", _dev(depth_order, torch.long))
        self.depth_order2 = _mirror_classes(self.depth_order1)
        self.count = _dev(count)
        self.is_overlap = self._keep("is_overlap", _dev(is_overlap, torch.long).contiguous())
        self._dep_t = self._keep("dep_t", torch.cat([self.depth_order1, self.depth_order2], 0))
        self.B = self.rgb.shape[0]

    # losses ----------------------------------------------------------------------------------------------------------
    def get_smooth_loss(self, disp, img, times=1):
        """supervised_order.py:214-235 (min-max normalisation, division by the mean, edge-aware gradient penalty): HIP
        kernels behind an autograd function (ops.smooth_loss); ``times`` = how many of the reference's two evaluations
        fall on this very map."""
        from . import ops
        return ops.smooth_loss(disp, img, times)

    def _disp_order_count(self, disp1, disp2, scale=1.0):
        """supervised_order.py:152-173 (no gradient flows through these counts in the reference either), for the whole
        batch in two launches: erosion of both masks, per-sample masked max / min, comparisons, masked counts."""
        from . import ops
        return ops.disp_order_count(disp1, disp2, self.modal1, self.modal2, self.depth_order1, self.is_overlap,
                                    self._LE_ORDER, scale)

    _LE_ORDER = 0     # both reference classes use `<=` on disp1 when depth_order1 == 0 (supervised_order.py:158-162, 289-293)

    PAIR_MODE = True      # False: two separate model calls, literally as supervised_order.py:187-188

    def _run(self, training):
        net = self.model
        B = self.B
        if self.PAIR_MODE:
            disp, dep, occ = net.module.forward_pair(self.rgb, self.modal1, self.modal2)
            d = disp.unsqueeze(1)
            return d, d, dep[:B], dep[B:], (occ[:B] if occ is not None else None), (occ[B:] if occ is not None else None)
        disp1, dep1, occ1 = net(self.rgb, self.modal1, self.modal2)
        disp2, dep2, occ2 = net(self.rgb, self.modal2, self.modal1)
        return disp1.unsqueeze(1), disp2.unsqueeze(1), dep1, dep2, occ1, occ2

    def _losses(self, outs, want_grad):
        disp1, disp2, dep1, dep2, occ1, occ2 = outs
        inv = 1.0 / self.world_size
        p = self.params
        B = self.B
        dep = torch.cat([dep1, dep2], 0).contiguous()
        l_ov, g_ov = engine.order_loss(dep, B, 0, 3, depth_target=self._dep_t, is_overlap=self.is_overlap,
                                       overlap_weight=p["overlap_weight"], distinct_weight=0.0, inv_world=inv,
                                       want_grad=want_grad)
        l_di, g_di = engine.order_loss(dep, B, 0, 3, depth_target=self._dep_t, is_overlap=self.is_overlap,
                                       overlap_weight=0.0, distinct_weight=p["distinct_weight"], inv_world=inv,
                                       want_grad=want_grad)
        loss_overlap, loss_distinct = l_ov[0], l_di[0]
        heads = [(dep, (g_ov + g_di) if want_grad else None)]
        loss_occ = 0
        if self.HAS_OCC and p["occ_order_weight"] != 0:
            occ = torch.cat([occ1, occ2], 0).contiguous()
            l_oc, g_oc = engine.order_loss(occ, B, 2, 0, occ_target=self._occ_t, inv_world=inv, want_grad=want_grad)
            loss_occ = l_oc[0]
            heads.append((occ, g_oc))
        loss_smooth = 0
        if p["smooth_weight"] != 0:
            sw = p["smooth_weight"] * inv            # (the weight rides in the kernel's output scale)
            if disp1 is disp2:       # pair mode: one disparity map serves both mask orders -- the same term twice
                loss_smooth = self.get_smooth_loss(disp1, self.rgb, 2 * sw)
            else:
                loss_smooth = self.get_smooth_loss(disp1, self.rgb, sw) + self.get_smooth_loss(disp2, self.rgb, sw)
        loss_disp_order = 0
        if p["dorder_weight"] != 0:
            loss_disp_order = self._disp_order_count(disp1, disp2, p["dorder_weight"] * inv)
        loss = loss_overlap + loss_distinct + loss_occ + loss_smooth + loss_disp_order
        logs = {"loss_overlap": loss_overlap, "loss_distinct": loss_distinct}
        if self.HAS_OCC:
            logs["loss_occ"] = loss_occ
        logs["loss_smooth"] = loss_smooth
        logs["loss_disp_order"] = loss_disp_order
        return logs, loss, heads, loss_smooth

    # API -------------------------------------------------------------------------------------------------------------
    def forward_only(self, ret_loss=True):
        with torch.no_grad():
            outs = self._run(False)
            logs, loss, _, _ = self._losses(outs, False)
        return logs, {"loss": loss}

    def _fwd_loss_bwd(self):
        """forward (both mask orders) + the five loss terms + backward + gradients gathered into the flat buffer"""
        from . import ops
        flat = hasattr(self.optim, "gather_grads")          # FlatSGD; torch.optim.Adam keeps per-tensor gradients
        plan = self._wplan if (self._wplan and self.PAIR_MODE and self._wplan.dtype == self.net._act_dtype()) else None
        recording = self._wplan is None and self.PAIR_MODE and hasattr(self.optim, "_spans")
        if recording:
            ops.WeightPlan.start_recording()
        if plan is not None:
            plan.prepare()                       # every dense filter of the net, one launch
        ops.WeightPlan.active = plan
        try:
            outs = self._run(True)
            logs, loss, heads, loss_smooth = self._losses(outs, True)
            self.optim.zero_grad()
            roots, grads = [h for h, _ in heads], [g for _, g in heads]
            if torch.is_tensor(loss_smooth) and loss_smooth.requires_grad:
                roots.append(loss_smooth)
                grads.append(torch.ones_like(loss_smooth))
            torch.autograd.backward(roots, grads)
            self.net.join_side_streams()
        finally:
            ops.WeightPlan.active = None
            recs = ops.WeightPlan.stop_recording() if recording else None
        if flat:
            self.optim.gather_grads(skip=plan.skip if plan is not None else None,
                                    prezeroed=bool(plan is not None and plan.vecs))
        if plan is not None:
            plan.unpack_grads()                  # ... and their gradients back, one launch
        if recording:
            self._wplan = ops.WeightPlan(self.optim, recs, self.net._act_dtype()) if recs else False
        logs = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in logs.items()}
        return logs, (loss.detach() if torch.is_tensor(loss) else loss)

    # ---- data-parallel step: staged backward + bucketed gradient exchange ----------------------------------------------
    STAGE_NAMES = ("heads+order branches+decoder", "encoder layer4", "encoder layer3", "encoder layer2+layer1")

    def grad_stage_slices(self):
        """[lo, hi) of the flat gradient buffer (optim.FlatSGD) that each backward stage finalises, in execution order.
        The parameters lie in creation order -- encoder layer1..layer4, decoder (scratch), order branches, heads -- and
        the backward pass walks them back to front, so a stage is one contiguous slice: everything behind the encoder
        (the order branches hang off l1..l3, the decoder off l1..l4), then encoder layer4, layer3, layer2 + layer1."""
        return [sl for sl, _ in self._stage_plan()]

    @property
    def flat_grads(self):
        return self.optim.flat_grads

    def _stage_plan(self):
        if self._stages is None:
            opt = self.optim
            names = {id(p): n for n, p in self.net.named_parameters()}
            offs = [off for off, _ in opt._spans]
            total = opt.flat_grads.numel()

            def first(prefix_ok):
                for i, p in enumerate(opt._params):
                    if prefix_ok(names[id(p)]):
                        return i
                raise RuntimeError("staged backward: parameter group not found")
            self._i2 = first(lambda n: n.startswith("pretrained.layer2."))
            i3 = first(lambda n: n.startswith("pretrained.layer3."))
            i4 = first(lambda n: n.startswith("pretrained.layer4."))
            it = first(lambda n: not n.startswith("pretrained."))
            if not (0 < i3 < i4 < it) or any(not names[id(p)].startswith("pretrained.") for p in opt._params[:it]) \
                    or any(names[id(p)].startswith("pretrained.") for p in opt._params[it:]):
                raise RuntimeError("staged backward: the flat buffer is not in encoder / decoder / branches order")
            n = len(opt._params)
            bounds = [(it, n), (i4, it), (i3, i4), (0, i3)]
            self._stages = [((offs[a], offs[b] if b < n else total), list(range(a, b))) for a, b in bounds]
        return self._stages

    def _staged_steps(self, plan):
        """Generator: forward (both mask orders) + the loss terms, then the backward pass stage by stage; yields the stage
        index each time that stage's slice of the flat gradient buffer is final (gathered, planned filters unpacked).
        The cut points are the encoder's stage outputs l1..l4 (midas_net._encode_decode): torch.autograd.grad from the
        losses to (parameters behind the encoder, l1..l4), then layer4 from dl4 to (its parameters, l3), and so on -- the
        gradients a boundary collects from several consumers are summed exactly as one backward() call would.  The
        result lands in ``self._staged_out``."""
        opt = self.optim
        stages = self._stage_plan()
        P = [[opt._params[i] for i in idx] for _, idx in stages]
        skip = plan.skip if plan is not None else None
        self.net._stage_cut = True            # every consumer of an encoder stage output reads a detached alias of it
        try:
            outs = self._run(True)
        finally:
            self.net._stage_cut = False
        logs, loss, heads, loss_smooth = self._losses(outs, True)
        roots, grads = [h for h, _ in heads], [g for _, g in heads]
        if torch.is_tensor(loss_smooth) and loss_smooth.requires_grad:
            roots.append(loss_smooth)
            grads.append(torch.ones_like(loss_smooth))
        raw = [r for r, _ in self.net._feats]                # l1..l4 as the encoder stages produced them
        cuts = [c for _, c in self.net._feats]               # ... and the leaves their consumers read
        # (nothing may keep this step's autograd graph alive into the next one: its AccumulateGrad nodes carry the stream
        # they were created on, and a node of an eager step re-used under a later stream capture breaks the capture)
        self.net._feats = None
        self._staged_out = ({k: (v.detach() if torch.is_tensor(v) else v) for k, v in logs.items()},
                            loss.detach() if torch.is_tensor(loss) else loss)

        def finish(si, idx, got):
            opt.gather_stage(idx, got, skip=skip, attach=True, prezeroed=bool(plan is not None and plan.vecs))
            if plan is not None and idx:
                lo = opt._spans[idx[0]][0]
                hi = opt._spans[idx[-1]][0] + opt._spans[idx[-1]][1]
                plan.unpack_grads(lo, hi)

        def add(a, b):
            return b if a is None else (a if b is None else a + b)

        def stage(si, out, g_out, idx, below):
            """backward of one encoder stage: d(out) -> its parameters `idx` (+ the gradient of the leaf `below` it reads)"""
            params = [opt._params[i] for i in idx]
            if g_out is None:                                # no loss reaches this stage: zero gradients
                finish(si, idx, [None] * len(idx))
                return None
            res = torch.autograd.grad([out], params + ([below] if below is not None else []), [g_out], allow_unused=True)
            finish(si, idx, res[:len(idx)])
            return res[-1] if below is not None else None

        # stage 0: heads, order branches, decoder -> their parameters and d(l1..l4) at the cuts
        res = torch.autograd.grad(roots, P[0] + cuts, grads, allow_unused=True)
        self.net.join_side_streams()          # (the branches' last filter-gradient launches: see midas_net.join_side_streams)
        finish(0, stages[0][1], res[:len(P[0])])
        g1, g2, g3, g4 = res[len(P[0]):]
        yield 0
        g3 = add(g3, stage(1, raw[3], g4, stages[1][1], cuts[2]))       # encoder layer4
        yield 1
        g2 = add(g2, stage(2, raw[2], g3, stages[2][1], cuts[1]))       # encoder layer3
        yield 2
        idx3 = stages[3][1]
        i2 = idx3.index(self._i2)
        g1 = add(g1, stage(3, raw[1], g2, idx3[i2:], cuts[0]))          # encoder layer2 ...
        stage(3, raw[0], g1, idx3[:i2], None)                           # ... and layer1 (incl. the stem)
        yield 3

    def _step_overlapped(self):
        """world_size > 1 (FlatSGD): the step of supervised_order.py:198-209 with the gradient exchange
        (utils/distributed_utils.py:27-31) cut into the four stage buckets of grad_stage_slices() and overlapped with the
        rest of the backward pass -- the 610 MB of InstaDepthNet_od's gradients are on the wire while the encoder's
        backward (most of the pass) still runs.  Second step of a shape: one hipGraph PER STAGE, replayed afterwards with
        the collectives launched in between (as the ResNet nets do); identical kernels and order, bit-identical results."""
        from . import ops
        if self._buckets is None:
            self._buckets = distributed_utils.GradientBuckets(self)
        bk = self._buckets
        key = (tuple(self.rgb.shape), self.rgb.data_ptr(), self.PAIR_MODE, self.optim.flat_params.data_ptr())
        graphs_ok = self._use_graph and not engine.prof_active()
        if graphs_ok and self._dp_graphs is not None and self._dp_key == key:
            for si, g in enumerate(self._dp_graphs):
                g.replay()
                bk.launch(si)
            bk.finish()
            logs, loss = self._dp_out
            return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in logs.items()}, loss.clone()
        plan = self._wplan if (self._wplan and self.PAIR_MODE and self._wplan.dtype == self.net._act_dtype()) else None
        recording = self._wplan is None and self.PAIR_MODE and hasattr(self.optim, "_spans")
        capture = graphs_ok and self._seen_key == key and not recording
        graphs = None
        if capture:
            # only the capture sits in the try-block: no collective has been launched yet, so falling back is safe
            try:
                torch.cuda.synchronize()
                ops.WeightPlan.active = plan
                gen = self._staged_steps(plan)
                graphs = []
                for si in range(bk.num_stages):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=graphs[0].pool() if graphs else None, capture_error_mode="thread_local"):
                        if si == 0 and plan is not None:
                            plan.prepare()
                        assert next(gen) == si
                    graphs.append(g)
                gen.close()
            except Exception as ex:   # noqa: BLE001 -- capture unsupported here: stay eager
                graphs = None
                self._use_graph = False
                print("instaorder_amd: per-stage hipGraph capture of the MiDaS step disabled (%s)" % ex)
            finally:
                ops.WeightPlan.active = None
        if graphs is not None:
            self._dp_graphs, self._dp_key, self._dp_out = graphs, key, self._staged_out
            for si, g in enumerate(graphs):
                g.replay()
                bk.launch(si)
            bk.finish()
            logs, loss = self._dp_out
            return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in logs.items()}, loss.clone()
        # eager (first step of a shape, profiling, graphs disabled)
        out = self._staged_fwd_bwd(bk.launch)
        bk.finish()
        self._seen_key = key
        return out

    def _staged_fwd_bwd(self, on_stage=None):
        """One pass of ``_staged_steps`` with the filter plan prepared / recorded around it; ``on_stage(si)`` is called as
        soon as stage si's slice of the flat gradient buffer is final (the data-parallel step launches its all-reduce)."""
        from . import ops
        plan = self._wplan if (self._wplan and self.PAIR_MODE and self._wplan.dtype == self.net._act_dtype()) else None
        recording = self._wplan is None and self.PAIR_MODE and hasattr(self.optim, "_spans")
        if recording:
            ops.WeightPlan.start_recording()
        if plan is not None:
            plan.prepare()
        ops.WeightPlan.active = plan
        try:
            for si in self._staged_steps(plan):
                if on_stage is not None:
                    on_stage(si)
        finally:
            ops.WeightPlan.active = None
            recs = ops.WeightPlan.stop_recording() if recording else None
        if recording:
            self._wplan = ops.WeightPlan(self.optim, recs, self.net._act_dtype()) if recs else False
        return self._staged_out

    def step(self):
        if not self.model.training:
            raise RuntimeError("step() needs switch_to('train')")
        if not hasattr(self.optim, "gather_grads"):
            # `optim: Adam` (single_stage_model.py:39-41): plain torch optimiser over the per-tensor gradients
            logs, loss = self._fwd_loss_bwd()
            if self.world_size > 1:
                distributed_utils.average_gradients(self.model)
            self.optim.step()
            from . import ops
            ops.WEIGHTS_EPOCH[0] += 1
            return logs, {"loss": loss}
        # world_size > 1: the backward in four stages, each stage's slice of the flat gradient buffer all-reduced as soon as
        # the stage is enqueued, one hipGraph per stage (the literal two-call mode keeps the flat exchange)
        staged = self.PAIR_MODE and (self.world_size > 1 or self._force_overlap) and self._overlap_comm
        if staged:
            logs, loss = self._step_overlapped()
            self.optim.step(gathered=True)
            return logs, {"loss": loss}
        key = (tuple(self.rgb.shape), self.rgb.data_ptr(), self.PAIR_MODE, self.optim.flat_params.data_ptr())
        if self._use_graph and self._graph is not None and self._graph_key == key and not engine.prof_active():
            self._graph.replay()
            logs, loss = self._graph_out
            logs, loss = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in logs.items()}, loss.clone()
        elif self._use_graph and self._seen_key == key and not engine.prof_active():
            # second step with this shape (the first ran eagerly and warmed every kernel): capture, then run it
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                self.optim.zero_grad()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    logs, loss = self._fwd_loss_bwd()
                self._graph, self._graph_key, self._graph_out = g, key, (logs, loss)
                g.replay()
                logs, loss = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in logs.items()}, loss.clone()
            except Exception as ex:   # noqa: BLE001 -- capture unsupported here: stay eager
                self._use_graph = False
                self._graph = None
                print("instaorder_amd: hipGraph capture of the MiDaS step disabled (%s)" % ex)
                logs, loss = self._fwd_loss_bwd()
        else:
            logs, loss = self._fwd_loss_bwd()
            self._seen_key = key
        if self.world_size > 1:
            distributed_utils.allreduce_flat(self.optim.flat_grads)
        self.optim.step(gathered=True)
        return logs, {"loss": loss}


class InstaDepthNet_od(_DepthBase):
    """supervised_order.py:97-235: disparity + depth order + occlusion order."""
    HAS_OCC = True

    def set_input(self, rgb=None, modal1=None, modal2=None, depth_order=None, count=None, is_overlap=None,
                  occ_order=None):
        self._set_common(rgb, modal1, modal2, depth_order, count, is_overlap)
        self.occ_order1 = _dev(occ_order, torch.float32)
        self.occ_order2 = _mirror_occ(self.occ_order1)
        self._occ_t = self._keep("occ_t", torch.cat([self.occ_order1, self.occ_order2], 0).contiguous())


class InstaDepthNet_d(_DepthBase):
    """supervised_order.py:239-367: disparity + depth order."""

    def set_input(self, rgb=None, modal1=None, modal2=None, depth_order=None, count=None, is_overlap=None):
        self._set_common(rgb, modal1, modal2, depth_order, count, is_overlap)
