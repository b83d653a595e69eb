"""Weight initialisation and checkpoint loading used on the hot path
(host-side mirror of utils/common_utils.py:35-65 and :128-149)."""
import os

import torch
from torch.nn import init


def init_weights(net, init_type="normal", init_gain=0.02):
    """Same rule as the reference's ``init_weights`` (common_utils.py:35-65): conv / linear weights from
    ``init_type`` (xavier-normal with gain 0.02 is what SingleStageModel uses), their biases to zero,
    BatchNorm weight ~ N(1, gain) and bias 0.  Works on the flat-buffer ResNet by parameter kind."""
    if not hasattr(net, "_param_list"):
        return _init_weights_generic(net, init_type, init_gain)
    with torch.no_grad():
        for t, p in net._param_list:
            kind = t["kind"]
            if kind in (0, 3):
                if init_type == "normal":
                    init.normal_(p, 0.0, init_gain)
                elif init_type == "xavier":
                    init.xavier_normal_(p, gain=init_gain)
                elif init_type == "kaiming":
                    init.kaiming_normal_(p, a=0, mode="fan_in")
                elif init_type == "orthogonal":
                    init.orthogonal_(p, gain=init_gain)
                else:
                    raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            elif kind == 4:
                p.zero_()
            elif kind == 1:
                init.normal_(p, 1.0, init_gain)
            elif kind == 2:
                p.zero_()
    return net


def _init_weights_generic(net, init_type, init_gain):
    """The reference's rule by class name (common_utils.py:37-60) for ordinary module trees (the order branches of
    InstaDepthNet_*: midas_net.py:156-157)."""
    def init_func(m):
        classname = m.__class__.__name__
        w = getattr(m, "weight", None)
        if w is not None and (classname.find("Conv") != -1 or classname.find("Linear") != -1):
            if init_type == "normal":
                init.normal_(w.data, 0.0, init_gain)
            elif init_type == "xavier":
                init.xavier_normal_(w.data, gain=init_gain)
            elif init_type == "kaiming":
                init.kaiming_normal_(w.data, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                init.orthogonal_(w.data, gain=init_gain)
            else:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            if getattr(m, "bias", None) is not None:
                init.constant_(m.bias.data, 0.0)
        elif classname.find("BatchNorm2d") != -1:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)

    net.apply(init_func)
    return net


def load_state(path, model, optimizer=None):
    """Load ``{'step','state_dict','optimizer'}`` written by save_state / by the reference
    (single_stage_model.py:66-72); non-strict, reports missing keys, returns the stored iteration
    (common_utils.py:128-149)."""
    if not os.path.isfile(path):
        raise Exception("=> no checkpoint found at '{}'".format(path))
    print("=> loading checkpoint '{}'".format(path))
    dev = next(model.parameters()).device
    checkpoint = torch.load(path, map_location=dev, weights_only=False)
    model.load_state_dict(checkpoint["state_dict"], strict=False)
    missing = set(model.state_dict().keys()) - set(checkpoint["state_dict"].keys())
    for k in missing:
        print("caution: missing keys from checkpoint {}: {}".format(path, k))
    last_iter = checkpoint["step"]
    if optimizer is not None:
        optimizer.load_state_dict(checkpoint["optimizer"])
        print("=> also loaded optimizer from checkpoint '{}' (iter {})".format(path, last_iter))
    return last_iter


class FixModule(torch.nn.Module):
    """Single-process stand-in for DistModule: same ``module.`` key prefix (models/backbone/others.py:3-10)."""

    def __init__(self, m):
        super(FixModule, self).__init__()
        self.module = m

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


# ---- harness helpers (utils/common_utils.py:66-127): used by trainer.py / tools/test.py around the hot path -----------------
def create_logger(name, log_file, level=None):
    """File + stream logger with the '[time] message' format (utils/common_utils.py:66-77)."""
    import logging
    lg = logging.getLogger(name)
    fmt = logging.Formatter("[%(asctime)s] %(message)s")
    for h in (logging.FileHandler(log_file), logging.StreamHandler()):
        h.setFormatter(fmt)
        lg.addHandler(h)
    lg.setLevel(logging.INFO if level is None else level)
    return lg


class AverageMeter(object):
    """Running (length = 0) or windowed (last ``length`` values) mean with ``.val`` / ``.avg``
    (utils/common_utils.py:80-109)."""

    def __init__(self, length=0):
        self.length = length
        self.reset()

    def reset(self):
        if self.length > 0:
            self.history = []
        else:
            self.count = 0
            self.sum = 0.0
        self.val = 0.0
        self.avg = 0.0

    def update(self, val):
        if self.length > 0:
            self.history.append(val)
            if len(self.history) > self.length:
                del self.history[0]
            self.val = self.history[-1]
            self.avg = sum(self.history) / float(len(self.history))
        else:
            self.val = val
            self.sum += val
            self.count += 1
            self.avg = self.sum / self.count


def accuracy(output, target, topk=(1,)):
    """precision@k in percent for logits [B,K] and class ids [B] (utils/common_utils.py:112-126)."""
    maxk = max(topk)
    pred = output.topk(maxk, 1, True, True)[1].t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0, keepdim=True) * (100.0 / target.size(0)) for k in topk]


def disp_to_depth(disp, min_depth, max_depth):
    """utils/common_utils.py:9-14: sigmoid disparity -> (scaled disparity, depth) between the two depth bounds."""
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    return scaled_disp, 1 / scaled_disp


class UnNormalize(object):
    """Inverse of the ImageNet normalisation, in place on a [C,H,W] tensor (utils/common_utils.py:17-32)."""

    def __init__(self):
        self.mean = [0.485, 0.456, 0.406]
        self.std = [0.229, 0.224, 0.225]

    def __call__(self, tensor):
        for t, m, s in zip(tensor, self.mean, self.std):
            t.mul_(s).add_(m)
        return tensor


def load_weights(path, model):
    """utils/common_utils.py:152-166: a bare state_dict file into ``model`` (strict=False), with the reference's
    warnings for keys the file lacks."""
    import os
    if not os.path.isfile(path):
        raise Exception("File not exist: {}".format(path))
    print("=> loading checkpoint '{}'".format(path))
    weights = torch.load(path, map_location="cpu")
    model.load_state_dict(weights, strict=False)
    missing = set(model.state_dict().keys()) - set(weights.keys())
    for k in missing:
        if "num_batches_tracked" not in k:
            print("caution: missing keys from checkpoint {}: {}".format(path, k))
