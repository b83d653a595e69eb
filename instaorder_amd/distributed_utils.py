"""Data-parallel layer: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

Mirrors the public names of the reference's utils/distributed_utils.py (DistModule,
average_gradients, broadcast_params, reduce_tensors, dist_init_, the index samplers) with the same
observable semantics -- loss pre-divided by world_size, gradient SUM all-reduce, parameters and
buffers broadcast from rank 0 at construction, rank-local BN statistics -- but the collectives are
re-designed for xGMI: the reference issues one all-reduce per parameter tensor (161 launches of 2 ..
2.4 M floats, distributed_utils.py:27-31) and 320 broadcasts (:34-37); here the whole model is ONE
94 MB flat bucket, i.e. one ring all-reduce whose 7 point-to-point links all stay busy, and three
broadcasts.  Everything is device-agnostic (gloo on CPU tensors in the tests, RCCL on the GPU).
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist
from torch.nn import Module
from torch.utils.data.sampler import Sampler


def _net_of(model):
    m = model
    while hasattr(m, "module"):
        m = m.module
    return m


def broadcast_flat(tensors, src=0):
    for t in tensors:
        dist.broadcast(t, src)


def broadcast_params(model):
    """Rank 0's parameters, BN running statistics and counters to every rank
    (distributed_utils.py:34-37 broadcasts each state_dict entry)."""
    net = _net_of(model)
    if hasattr(net, "flat_params"):
        broadcast_flat([net.flat_params, net.flat_running, net._nbt])
    else:
        for p in model.state_dict().values():
            dist.broadcast(p, 0)


def allreduce_flat(flat):
    """SUM all-reduce of one flat gradient bucket (the semantics of distributed_utils.py:27-31)."""
    dist.all_reduce(flat)
    return flat


class GradientBuckets(object):
    """The gradient exchange of the reference (utils/distributed_utils.py:27-31: SUM all-reduce of every parameter
    gradient, after the whole backward) cut into the stages of the backward pass and overlapped with it.

    The backward runs heads + layer4 -> layer3 -> layer2 -> layer1 + stem (csrc/net.hip: io_net_backward_stages) and the
    parameters lie in creation order in one flat buffer, so what a stage finalises is one contiguous slice of the flat
    gradient buffer (``net.grad_stage_slices()``): 60 / 28 / 5 / 1 MB.  ``launch(stage)`` is called right after a stage
    has been enqueued: the collective is issued asynchronously -- RCCL's stream takes a dependency on everything the
    compute stream holds at that moment, i.e. on that stage, and the compute stream is NOT made to wait -- so layer4's
    60 MB ride the xGMI ring under the ~80 % of the backward that is still to run.  ``finish()`` joins them before the
    optimiser.  Per element the arithmetic is that of the flat call: one SUM over ranks of the same numbers (bit-identical
    on two ranks, where the sum is commutative; with more ranks a ring may associate differently per chunking, as any
    two bucketings of the reference's 161 calls would)."""

    def __init__(self, net):
        self.net = _net_of(net)
        self.slices = self.net.grad_stage_slices()
        self.pending = []

    @property
    def num_stages(self):
        return len(self.slices)

    def launch(self, stage):
        lo, hi = self.slices[stage]
        self.pending.append(dist.all_reduce(self.net.flat_grads[lo:hi], async_op=True))

    def finish(self):
        for w in self.pending:
            w.wait()
        self.pending = []

    def all_at_once(self):
        """every bucket, in backward order, after the fact (the arithmetic of the overlapped form without the overlap)"""
        for s in range(self.num_stages):
            self.launch(s)
        self.finish()


def average_gradients(model, bucketed=False):
    """Gradient exchange of the reference (SUM; the loss was already divided by world_size,
    supervised_order.py:543).  One collective for the whole model, or (``bucketed``) the four stage buckets of
    GradientBuckets -- what the training step overlaps with its backward pass."""
    net = _net_of(model)
    if hasattr(net, "flat_grads") and bucketed:
        GradientBuckets(net).all_at_once()
    elif hasattr(net, "flat_grads"):
        allreduce_flat(net.flat_grads)
    else:
        for param in model.parameters():
            if param.requires_grad and param.grad is not None:
                dist.all_reduce(param.grad.data)


class DistModule(Module):
    """Data-parallel wrapper (distributed_utils.py:13-24): forwards to ``module`` and makes every rank
    start from rank 0's state.  state_dict keys get the ``module.`` prefix, as in the reference."""

    def __init__(self, module):
        super(DistModule, self).__init__()
        self.module = module
        broadcast_params(self.module)

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)

    def train(self, mode=True):
        super(DistModule, self).train(mode)
        self.module.train(mode)
        return self


def reduce_tensors(tensor):
    """Cross-rank SUM of a (cloned) scalar such as the logged loss (distributed_utils.py:133-136)."""
    reduced = tensor.clone()
    dist.all_reduce(reduced)
    return reduced


def dist_init_(launcher="pytorch", backend="nccl", dist_url=None):
    """Process-group bring-up for one process per GPU on one node (distributed_utils.py:53-60).
    RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* come from the launcher (torch.distributed.run)."""
    if launcher != "pytorch":
        raise ValueError("Invalid launcher type: {}".format(launcher))
    rank = int(os.environ.get("RANK", "0"))
    if backend == "nccl":
        ngpu = torch.cuda.device_count()
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank % max(ngpu, 1))))
    world = int(os.environ.get("WORLD_SIZE", torch.cuda.device_count() if backend == "nccl" else 1))
    if dist_url is None:
        dist_url = "tcp://{}:{}".format(os.environ.get("MASTER_ADDR", "127.0.0.1"),
                                        os.environ.get("MASTER_PORT", "1234"))
    dist.init_process_group(backend=backend, init_method=dist_url, world_size=world, rank=rank)
    return rank, world


def shard_range(total, world_size, rank):
    """Contiguous slice [beg, end) of ``ceil(total/world)`` items for ``rank``; indices past ``total``
    wrap around to the start (the padding rule of DistributedSequentialSampler,
    distributed_utils.py:149-153).  Used to shard the O(n^2) pair list across GPUs."""
    sub = int(math.ceil(total * 1.0 / world_size))
    beg = sub * rank
    return beg, beg + sub, sub


class DistributedSequentialSampler(Sampler):
    def __init__(self, dataset, world_size=None, rank=None):
        self.world_size = dist.get_world_size() if world_size is None else world_size
        self.rank = dist.get_rank() if rank is None else rank
        self.dataset = dataset
        n = len(dataset)
        assert n >= self.world_size, "{} vs {}".format(n, self.world_size)
        self.beg, self.end, sub = shard_range(n, self.world_size, self.rank)
        self.padded_ind = list(range(n)) + list(range(sub * self.world_size - n))

    def __iter__(self):
        return iter(self.padded_ind[self.beg:self.end])

    def __len__(self):
        return self.end - self.beg


class DistributedGivenIterationSampler(Sampler):
    """Fixed-length index stream for iteration-based training (distributed_utils.py:203-254): all ranks
    shuffle the same tiled index list with numpy seed 0 and take consecutive slices of
    ``total_iter * batch_size``; resuming skips ``(last_iter + 1) * batch_size`` entries."""

    def __init__(self, dataset, total_iter, batch_size, world_size=None, rank=None, last_iter=-1):
        self.world_size = dist.get_world_size() if world_size is None else world_size
        self.rank = dist.get_rank() if rank is None else rank
        assert self.rank < self.world_size
        self.dataset = dataset
        self.total_iter = total_iter
        self.batch_size = batch_size
        self.last_iter = last_iter
        self.total_size = total_iter * batch_size
        self.indices = self._make_indices()
        self._used = False

    def _make_indices(self):
        np.random.seed(0)
        want = self.total_size * self.world_size
        base = np.arange(len(self.dataset))[:want]
        reps = (want - 1) // base.shape[0] + 1
        idx = np.tile(base, reps)[:want]
        np.random.shuffle(idx)
        beg = self.total_size * self.rank
        idx = idx[beg:beg + self.total_size]
        assert len(idx) == self.total_size
        return idx

    def __iter__(self):
        if self._used:
            raise RuntimeError("this sampler is not designed to be called more than once!!")
        self._used = True
        return iter(self.indices[(self.last_iter + 1) * self.batch_size:])

    def __len__(self):
        return self.total_size


# ---- the remaining public names of utils/distributed_utils.py ------------------------------------------------------------
def dist_init(launcher, backend="nccl", **kwargs):
    """distributed_utils.py:40-50; only the 'pytorch' launcher (torch.distributed.run) exists on this platform -- the
    'mpi' / 'slurm' branches of the reference have no counterpart here."""
    if launcher != "pytorch":
        raise ValueError("Invalid launcher type: {}".format(launcher))
    return dist_init_(launcher, backend, kwargs.get("dist_url"))


class GivenIterationSampler(DistributedGivenIterationSampler):
    """The single-process form (distributed_utils.py:163-200): the same tiled, seed-0-shuffled index stream."""

    def __init__(self, dataset, total_iter, batch_size, last_iter=-1):
        super(GivenIterationSampler, self).__init__(dataset, total_iter, batch_size, world_size=1, rank=0,
                                                    last_iter=last_iter)
