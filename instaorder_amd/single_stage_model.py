"""SingleStageModel -- host-side mirror of models/single_stage_model.py:11-78.

Builds the backbone named by ``params['backbone_arch']`` with ``params['backbone_param']``, applies
the xavier(0.02) initialisation, places it on the GPU, wraps it (DistModule / FixModule), creates the
optimiser (``.optim`` is a torch.optim.Optimizer whose lr the scheduler rewrites) and provides the
checkpoint I/O.  ``InstaDepthNet_od`` / ``InstaDepthNet_d`` build the MiDaS-based nets of midas_net.py.
"""
import os

import torch
import torch.distributed as dist

from . import common_utils, distributed_utils, midas_net, resnet_cls
from .optim import FlatSGD, FusedSGD

_BACKBONES = {"resnet50_cls": resnet_cls.resnet50_cls}


class SingleStageModel(object):
    def __init__(self, params, dist_model=False):
        if params["algo"] in ("InstaDepthNet_od", "InstaDepthNet_d"):
            # single_stage_model.py:17-22: the MiDaS-based nets take only the (optional) pretrained MiDaS weights
            cls = midas_net.InstaDepthNet_od if params["algo"] == "InstaDepthNet_od" else midas_net.InstaDepthNet_d
            net = cls(params.get("pretrained_weight"), non_negative=True)
            net.dtype = params.get("dtype", "fp32")
        else:
            arch = params["backbone_arch"]
            if arch not in _BACKBONES:
                raise KeyError("unknown backbone_arch '{}' (have: {})".format(arch, sorted(_BACKBONES)))
            # `dtype` is this package's one extension of the config surface: 'fp32' (reference behaviour) | 'bf16'
            net = _BACKBONES[arch](dtype=params.get("dtype", "fp32"), **params["backbone_param"])
            common_utils.init_weights(net, init_type="xavier")
        if torch.cuda.is_available():
            net.cuda()
        if dist_model:
            self.model = distributed_utils.DistModule(net)
            self.world_size = dist.get_world_size()
        else:
            self.model = common_utils.FixModule(net)
            self.world_size = 1
        self.net = net

        if params["optim"] == "SGD" and not hasattr(net, "flat_params"):
            self.optim = FlatSGD(self.model, lr=params["lr"], momentum=0.9, weight_decay=params["weight_decay"])
        elif params["optim"] == "SGD":
            self.optim = FusedSGD(self.model, lr=params["lr"], momentum=0.9,
                                  weight_decay=params["weight_decay"])
        elif params["optim"] == "Adam":
            # not on the measured path: plain torch Adam over the strided parameter views
            self.optim = torch.optim.Adam(self.model.parameters(), lr=params["lr"],
                                          betas=(params["beta1"], 0.999))
        else:
            raise Exception("No such optimizer: {}".format(params["optim"]))

    def forward_only(self, ret_loss=True):
        pass

    def step(self):
        pass

    def load_state(self, path, Iter=None, resume=False):
        if Iter is not None:
            path = os.path.join(path, "ckpt_iter_{}.pth.tar".format(Iter))
        if resume:
            common_utils.load_state(path, self.model, self.optim)
        else:
            common_utils.load_state(path, self.model)

    def load_pretrain(self, load_path):
        common_utils.load_state(load_path, self.model)

    def save_state(self, path, Iter):
        path = os.path.join(path, "ckpt_iter_{}.pth.tar".format(Iter))
        sd = {k: v.detach().clone().contiguous() for k, v in self.model.state_dict().items()}
        torch.save({"step": Iter, "state_dict": sd, "optimizer": self.optim.state_dict()}, path)

    def switch_to(self, phase):
        if phase == "train":
            self.model.train()
        else:
            self.model.eval()
