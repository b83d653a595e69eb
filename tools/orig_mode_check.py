#!/usr/bin/env python3
"""The 'orig' inference mode at a COCO-sized image (480 x 640, the size class of InstaOrder's images): all pairs of a
12-instance scene through InstaOrderNet_od on the H x W input, timed, and the logits of a few pairs against the CPU oracle
on the same planes.   usage: python tools/orig_mode_check.py [H=480] [W=640] [instances=12]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import instaorder_amd as ia  # noqa: E402
from instaorder_amd import inference, synthetic  # noqa: E402
from oracle import resnet_oracle as orc  # noqa: E402      (checker only)

H = int(sys.argv[1]) if len(sys.argv) > 1 else 480
W = int(sys.argv[2]) if len(sys.argv) > 2 else 640
NI = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rng = np.random.RandomState(5)
image = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
modal = np.zeros((NI, H, W), np.uint8)
for i in range(NI):
    y0, x0 = rng.randint(0, H - 80), rng.randint(0, W - 80)
    modal[i, y0:y0 + rng.randint(40, 80), x0:x0 + rng.randint(40, 80)] = 1
cfg = dict(algo="InstaOrderNet_od", lr=1e-3, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls",
           backbone_param=dict(in_channels=5, num_classes=[2, 3]), use_rgb=True, overlap_weight=0.1, distinct_weight=0.9)
m = ia.InstaOrderNet_od(cfg, dist_model=False)
sd = synthetic.make_state_dict(3, 5, [2, 3], prefix="module.", style="kaiming")
m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
m.switch_to("eval")
hh, ww = inference.get_closest_int_multiple_of(H, 32), inference.get_closest_int_multiple_of(W, 32)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    occ, dep = inference.infer_order_sup_occ_depth(m, image, modal, None, "all", "InstaOrderNet_od", "orig", 256)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
pairs = inference.upper_pairs(NI)
print("'orig' mode: %d x %d image -> %d x %d network input, %d instances, %d pairs: %.1f ms (%.0f pairs/s incl. the "
      "device pre-processing)" % (H, W, hh, ww, NI, len(pairs), dt * 1e3, len(pairs) / dt))
rgb, masks = inference.orig_mode_inputs("cuda:0", image, modal)
res = inference.infer_order_batched(m, rgb, masks, "InstaOrderNet_od", pairs=pairs, return_logits=True)
state = orc.state_from_numpy({k[len("module."):]: v.detach().cpu().numpy() for k, v in m.model.state_dict().items()})
sel = [0, len(pairs) // 2, len(pairs) - 1]
mk, im = masks.cpu(), rgb.cpu()
worst = 0.0
for k in sel:
    i, j = pairs[k]
    with torch.no_grad():
        z1 = torch.cat(orc.resnet_forward(state, torch.cat([mk[i][None, None], mk[j][None, None], im], 1), False), 1)
        z2 = torch.cat(orc.resnet_forward(state, torch.cat([mk[j][None, None], mk[i][None, None], im], 1), False), 1)
    ref = torch.cat([z1, z2], 1).numpy()[0]
    err = float(np.abs(res["pair_logits"][k] - ref).max() / max(np.abs(ref).max(), 1e-6))
    worst = max(worst, err)
print("logits of pairs %s against the CPU oracle on the same planes: max relative error %.2e" % (sel, worst))
assert worst < 1e-3
