"""MiDaS-based nets, CPU side: the oracle restatement against the goldens the real reference
produced (tests/golden/make_golden.py depthnet_*), and the host module's state_dict layout against the reference's."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, synthetic
from oracle import midas_oracle as mo

CASES = [("InstaDepthNet_od", "depthnet_od_S64_B2", "od"), ("InstaDepthNet_d", "depthnet_d_S64_B2", "d")]
WEIGHTS = dict(overlap_weight=0.1, distinct_weight=0.9, dorder_weight=1.0, smooth_weight=0.1, occ_order_weight=1.0)


def load(tag):
    g = np.load(os.path.join(GOLDEN, tag + ".npz"), allow_pickle=False)
    spec = [(str(k), tuple(int(d) for d in str(s).split(",") if d), (str(a) or None))
            for k, s, a in zip(g["keys"], g["shapes"], g["aliases"])]
    return g, spec


@pytest.mark.parametrize("algo,tag,variant", CASES)
def test_module_state_dict_matches_reference(algo, tag, variant):
    from instaorder_amd import midas_net
    g, spec = load(tag)
    net = getattr(midas_net, algo)(None)
    sd = net.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in spec]
    for k, shape, _ in spec:
        assert tuple(sd[k].shape) == shape, k
    # aliases: the same keys share storage in both
    for k, _, alias in spec:
        if alias:
            assert sd[k].data_ptr() == sd[alias].data_ptr(), (k, alias)
    assert len(list(net.parameters())) == len(g["names"])
    assert [n for n, _ in net.named_parameters()] == [str(n) for n in g["names"]]


@pytest.mark.parametrize("algo,tag,variant", CASES)
def test_oracle_matches_reference(algo, tag, variant):
    g, spec = load(tag)
    S, B, seed = (int(v) for v in g["meta"])
    sd = synthetic.make_spec_state_dict(seed, spec)
    st = mo.state_from_numpy(sd, prefix="")
    batch = {k: torch.from_numpy(v.copy()) for k, v in synthetic.make_depth_batch(seed + 100, B, S).items()}
    with torch.no_grad():
        d, dep, occ = mo.forward(st, batch["rgb"], batch["modal1"], batch["modal2"], False, variant)
    assert np.allclose(d.numpy(), g["eval_disp"], rtol=1e-5, atol=1e-6)
    assert np.allclose(dep.numpy(), g["eval_dep"], rtol=1e-5, atol=1e-6)
    if variant == "od":
        assert np.allclose(occ.numpy(), g["eval_occ"], rtol=1e-5, atol=1e-6)
    # one training step: both mask orders, losses, gradients
    o1 = mo.forward(st, batch["rgb"], batch["modal1"], batch["modal2"], True, variant)
    o2 = mo.forward(st, batch["rgb"], batch["modal2"], batch["modal1"], True, variant)
    logs, total = mo.losses(o1, o2, batch, WEIGHTS, 1, variant)
    for k, v in logs.items():
        assert abs(float(v) - float(g["step_" + k])) <= 1e-5 * max(1.0, abs(float(g["step_" + k]))), k
    assert abs(float(total) - float(g["step_loss"])) <= 1e-5 * abs(float(g["step_loss"]))
    names = [str(n) for n in g["names"]]
    params = [st[n] for n in names]
    grads = torch.autograd.grad(total, params, allow_unused=True)
    worst = 0.0
    for n, gr, ref in zip(names, grads, g["grad_norms"]):
        got = 0.0 if gr is None else float(gr.double().norm())
        worst = max(worst, abs(got - float(ref)) / max(float(ref), 1e-12)) if ref > 1e-10 else worst
    assert worst < 1e-3, worst
    assert np.allclose(torch.cat([st[k].reshape(-1) for k, _, a in spec if k.endswith("running_mean") and not a]).numpy(),
                       g["step_running_mean"], rtol=1e-5, atol=1e-6)
