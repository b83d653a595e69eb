"""instaorder_amd -- MI355X-native pairwise order prediction (InstaOrder hot path).

Public surface mirrors the reference's ``models`` / ``utils`` packages for this path:

    from instaorder_amd import InstaOrderNet_o, InstaOrderNet_od, InstaOrderNet_d, OrderNet
    from instaorder_amd import InstaDepthNet_od, InstaDepthNet_d       # MiDaS-based nets (midas_net.py; MidasNet there too)
    from instaorder_amd import backbone            # backbone.resnet50_cls
    from instaorder_amd import utils               # DistModule, average_gradients, StepLRScheduler, ...
    from instaorder_amd import evaluate            # tools/test.py Tester loops (P / R / F1, WHDR) over the batched drivers
    from instaorder_amd import datasets            # SupOcclusionOrderBatches, SupDepthOccOrderBatches, SupDepthOrderBatches, PairRenderer

Importing the package does not load the HIP library; the first op does, and fails loudly when
it is missing or no gfx950 device is visible (there is no CPU fallback).
"""
import types as _types

from . import synthetic  # noqa: F401  (numpy only)


def __getattr__(name):
    # lazy: keeps `import instaorder_amd.synthetic` usable where torch/ctypes setup is unwanted
    if name in ("InstaOrderNet_o", "InstaOrderNet_od", "InstaOrderNet_d", "OrderNet", "InstaDepthNet_od",
                "InstaDepthNet_d"):
        from . import supervised_order
        return getattr(supervised_order, name)
    if name == "SingleStageModel":
        from .single_stage_model import SingleStageModel
        return SingleStageModel
    if name == "backbone":
        from . import common_utils, resnet_cls
        ns = _types.SimpleNamespace(resnet50_cls=resnet_cls.resnet50_cls, ResNet=resnet_cls.ResNet,
                                    FixModule=common_utils.FixModule)
        return ns
    if name == "utils":
        from . import common_utils, distributed_utils, scheduler
        ns = _types.SimpleNamespace()
        for mod in (common_utils, distributed_utils, scheduler):
            for k, v in vars(mod).items():
                if not k.startswith("_"):
                    setattr(ns, k, v)
        return ns
    raise AttributeError("module 'instaorder_amd' has no attribute %r" % name)
