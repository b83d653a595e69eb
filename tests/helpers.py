"""Shared helpers for the parity tests (test-only; may import oracle/)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from instaorder_amd import synthetic  # noqa: E402
from oracle import resnet_oracle as orc  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
NSAMP = 64

ALGO_CLASSES = {"InstaOrderNet_o": 2, "InstaOrderNet_od": [2, 3], "InstaOrderNet_d": 3, "OrderNet": 3}
ALGO_LR = {"InstaOrderNet_o": 1e-3, "InstaOrderNet_od": 1e-4, "InstaOrderNet_d": 1e-4, "OrderNet": 1e-3}


def sample_idx(n):
    return (np.arange(NSAMP, dtype=np.int64) * 2654435761) % max(n, 1)


def load_golden(tag):
    return np.load(os.path.join(GOLDEN, tag + ".npz"), allow_pickle=False)


def oracle_state(seed, algo, style="xavier"):
    return orc.state_from_numpy(synthetic.make_state_dict(seed, 5, ALGO_CLASSES[algo], style=style))


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def norms_and_samples(tensors):
    """tensors: iterable of torch tensors in OIHW logical order."""
    norms, samp = [], []
    for t in tensors:
        a = t.detach().contiguous().reshape(-1).double().cpu().numpy()
        norms.append(np.sqrt((a * a).sum()))
        samp.append(a[sample_idx(a.size)].astype(np.float32))
    return np.array(norms), np.stack(samp)


def bn_vectors(state):
    rm = np.concatenate([v.reshape(-1).cpu().numpy() for k, v in state.items() if k.endswith("running_mean")])
    rv = np.concatenate([v.reshape(-1).cpu().numpy() for k, v in state.items() if k.endswith("running_var")])
    nb = np.array([int(v) for k, v in state.items() if k.endswith("num_batches_tracked")])
    return rm, rv, nb


def eval_logits_oracle(state, batch):
    tb = {k: torch.as_tensor(v) for k, v in batch.items()}
    with torch.no_grad():
        o = orc.resnet_forward(state, torch.cat([tb["modal1"], tb["modal2"], tb["rgb"]], 1), False)
    if isinstance(o, tuple):
        return torch.cat(o, 1).numpy()
    return o.numpy()
