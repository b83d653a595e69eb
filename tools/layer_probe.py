"""Layer-by-layer comparison of the bf16 network against the fp32 one on the same weights and inputs (GPU)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from instaorder_amd import engine, synthetic
from instaorder_amd.resnet_cls import ResNet


def acts(net, x8, N, S, G):
    net.train()
    logits, ws = net._run_forward(x8, N, S, G, True)
    out = []
    dims = [(S // 2, 64), (S // 2, 64), (S // 4, 64)]
    H = S // 4
    for planes, nblk, stride in [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]:
        for j in range(nblk):
            if j == 0:
                H //= stride
            dims.append((H, planes * 4))
    el = 2 if net.dtype == "bf16" else 4
    for i, (h, c) in enumerate(dims):
        if i == 1:
            continue                  # relu(bn1(.)) lives only inside the pooling kernel
        off = net.plan.activation_offset(N, S, i)
        n = N * h * h * c
        raw = ws[off:off + n * el]
        t = raw.view(torch.bfloat16 if el == 2 else torch.float32).float().view(N, h, h, c)
        out.append(t.clone())
    return logits.clone(), out


def main():
    N, S, G = 8, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 1
    sd = synthetic.make_state_dict(97, 5, 2, prefix="", style="kaiming")
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    g3 = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * g3
    b = synthetic.make_pair_batch(980, N, S)
    x = torch.cat([torch.from_numpy(b[k]) for k in ("modal1", "modal2", "rgb")], 1).cuda()
    res = {}
    for dt in ("fp32", "bf16"):
        net = ResNet(5, 2, dtype=dt).cuda()
        net.load_state_dict(sd)
        res[dt] = acts(net, engine.pack_nchw(x, dtype=dt), N, S, G)
    la, a = res["fp32"]
    lb, bb = res["bf16"]
    for i, (p, q) in enumerate(zip(a, bb)):
        print("act %2d  %s  rel %.3e" % (i, tuple(p.shape), float((p - q).norm() / p.norm())))
    print("logits rel %.3e" % float((la - lb).norm() / la.norm()), la[0].tolist(), lb[0].tolist())


main()
