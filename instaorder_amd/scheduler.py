"""Iteration-indexed step learning-rate schedule with optional piecewise-linear warm-up.

Same constructor and behaviour as ``utils.StepLRScheduler`` of the reference
(utils/scheduler.py:84-109 on top of :5-33 and :58-81): ``step(it)`` writes
``lr = initial_lr * scale(it)`` into every param group, where during warm-up the lr is interpolated
linearly through ``(warmup_steps[i], warmup_lr[i])`` starting from ``base_lr`` at iteration 0, and
afterwards ``scale = prod(lr_mults[:k])`` with k = number of milestones <= it (times
``warmup_lr[-1] / base_lr`` if a warm-up was configured).
"""
from bisect import bisect_right

import torch


class StepLRScheduler(object):
    def __init__(self, optimizer, milestones, lr_mults, base_lr, warmup_lr, warmup_steps, last_iter=-1):
        if not isinstance(optimizer, torch.optim.Optimizer):
            raise TypeError("{} is not an Optimizer".format(type(optimizer).__name__))
        if len(milestones) != len(lr_mults):
            raise AssertionError("{} vs {}".format(milestones, lr_mults))
        if not (isinstance(warmup_lr, list) and isinstance(warmup_steps, list)
                and len(warmup_lr) == len(warmup_steps)):
            raise AssertionError("warmup_lr / warmup_steps must be lists of equal length")
        if any(not isinstance(m, int) for m in milestones) or list(milestones) != sorted(milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}".format(milestones))
        self.optimizer = optimizer
        for i, group in enumerate(optimizer.param_groups):
            if last_iter == -1:
                group.setdefault("initial_lr", group["lr"])
            elif "initial_lr" not in group:
                raise KeyError("param 'initial_lr' is not specified in param_groups[{}] "
                               "when resuming an optimizer".format(i))
        self.base_lrs = [group["initial_lr"] for group in optimizer.param_groups]
        self.last_iter = last_iter
        self.base_lr = base_lr
        self.warmup_lr = warmup_lr
        self.warmup_steps = warmup_steps
        self.milestones = milestones
        self.lr_mults = [1.0]
        for m in lr_mults:
            self.lr_mults.append(self.lr_mults[-1] * m)

    def _scale(self):
        it = self.last_iter
        seg = bisect_right(self.warmup_steps, it)
        if seg < len(self.warmup_steps):
            if seg == 0:
                lr0, it0 = self.base_lr, 0
            else:
                lr0, it0 = self.warmup_lr[seg - 1], self.warmup_steps[seg - 1]
            cur = lr0 + (it - it0) * (self.warmup_lr[seg] - lr0) / (self.warmup_steps[seg] - it0)
            return cur / self.base_lr
        k = bisect_right(self.milestones, it)
        if len(self.warmup_lr) == 0:
            return self.lr_mults[k]
        return self.warmup_lr[-1] * self.lr_mults[k] / self.base_lr

    def get_lr(self):
        return [group["lr"] for group in self.optimizer.param_groups]

    def step(self, this_iter=None):
        self.last_iter = self.last_iter + 1 if this_iter is None else this_iter
        scale = self._scale()
        for group, base in zip(self.optimizer.param_groups, self.base_lrs):
            group["lr"] = scale * base
