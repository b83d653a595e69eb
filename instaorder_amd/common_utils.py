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
