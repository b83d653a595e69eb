"""BASELINE.json configs[3] and configs[4] at their own shapes, and the N > 1 data-parallel path on the HIP engine.

* configs[3]: InstaOrderNet_od, 20-instance images (190 pairs each), the pair list sharded by rank for 8 ranks.  One
  GPU plays the 8 ranks in turn: (a) inference -- every rank's shard through ``infer_order_batched``, the union must be
  the unsharded result bit for bit; (b) training -- every rank's shard through forward + loss/8 + backward, the SUM of
  the 8 rank gradients (what the RCCL all-reduce delivers, utils/distributed_utils.py:27-31) against the CPU oracle
  doing the same, in fp32 (tight) and in bf16 (the dtype configs[3] names; the bf16 bar of tests/test_gpu_bf16.py).
* two ranks for real: tests/dp_worker.py under torch.distributed.run -- RCCL when the box has two GPUs, otherwise gloo
  with both ranks on GPU 0 -- against the golden the REAL reference produced on two ranks.
* configs[4]: InstaDepthNet_od at 384 x 384, fp32 and bf16, against oracle/midas_oracle.py (pinned at 64 x 64 by the
  reference goldens of tests/test_gpu_midas.py; the oracle is size-agnostic)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import ALGO_CLASSES, GOLDEN, ROOT, orc, rel_err, synthetic

pytestmark = pytest.mark.gpu
WORLD8 = 8


def _cfg(algo, dtype="fp32"):
    return dict(algo=algo, lr=1e-4, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls", dtype=dtype,
                backbone_param=dict(in_channels=5, num_classes=ALGO_CLASSES[algo]), use_rgb=True, overlap_weight=0.1,
                distinct_weight=0.9)


def _build(algo, seed, dtype="fp32", damp=False):
    import instaorder_amd as ia
    m = getattr(ia, algo)(_cfg(algo, dtype), dist_model=False)
    sd = synthetic.make_state_dict(seed, 5, ALGO_CLASSES[algo], prefix="module.", style="kaiming")
    if damp:      # well-conditioned net for the bf16 comparison (tests/test_gpu_bf16.py explains why)
        for k in sd:
            if k.endswith("bn3.weight"):
                sd[k] = (sd[k] * 0.1).astype(np.float32)
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    return m, sd


def _image20(seed, S):
    item = synthetic.make_images(seed, 1, 20, S)[0]
    rgb, masks = synthetic.image_mode_inputs(item["image"], item["modal"], S)
    return torch.from_numpy(rgb), torch.from_numpy(masks)


def test_config3_pair_list_sharded_over_8_ranks_inference():
    from instaorder_amd import distributed_utils as du, inference
    algo, S = "InstaOrderNet_od", 256
    m, _ = _build(algo, 301)
    m.switch_to("eval")
    rgb, masks = _image20(4242, S)
    pairs = inference.upper_pairs(20)
    assert len(pairs) == 190
    whole = inference.infer_order_batched(m, rgb, masks, algo, return_logits=True)
    got = []
    for r in range(WORLD8):
        beg, end, sub = du.shard_range(len(pairs), WORLD8, r)
        assert sub == 24 and end - beg == 24
        mine = [pairs[k % len(pairs)] for k in range(beg, end)]          # rank 7 wraps around: 2 padding pairs
        res = inference.infer_order_batched(m, rgb, masks, algo, pairs=mine, return_logits=True)
        got.append(res["pair_logits"])
    union = np.concatenate(got, 0)
    assert union.shape[0] == 192
    # eval mode has no cross-sample coupling: a pair's logits do not depend on which batch it rides in
    assert np.array_equal(union[:190], whole["pair_logits"])
    assert np.array_equal(union[190:], whole["pair_logits"][:2])         # the wrap-around duplicates
    # matrices rebuilt from the gathered logits = the unsharded matrices
    l = torch.from_numpy(union[:190])
    dec = inference.decide(l[:, :5], l[:, 5:], 2, 3)
    occ, dep = np.zeros((20, 20), np.int64), np.zeros((20, 20), np.int64)
    for k, (i, j) in enumerate(pairs):
        occ[i, j], occ[j, i] = int(dec["i_over_j"][k]), int(dec["j_over_i"][k])
        d = int(dec["depth"][k])
        dep[i, j], dep[j, i] = {0: (1, 0), 1: (0, 1), 2: (2, 2)}[d]
    assert (occ == whole["occ_order"]).all() and (dep == whole["depth_order"]).all()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_config3_sharded_training_step_equals_oracle_sum(dtype):
    """8 ranks x 24 pairs of one 20-instance image: per-rank loss / 8, gradients SUMMED over ranks (the all-reduce),
    rank-local BatchNorm statistics -- HIP path vs the CPU oracle on the same shards."""
    from instaorder_amd import distributed_utils as du, engine, inference
    algo, S = "InstaOrderNet_od", 64
    m, sd = _build(algo, 302, dtype, damp=(dtype == "bf16"))
    m.world_size = WORLD8                         # loss / world_size, as on a real 8-rank job
    m.switch_to("train")
    m.optim.param_groups[0]["lr"] = 0.0
    rgb, masks = _image20(4243, S)
    pairs = inference.upper_pairs(20)
    rng = np.random.RandomState(5)
    st0 = {k: v.clone() for k, v in m.model.state_dict().items()}
    gsum = torch.zeros_like(m.net.flat_grads)
    osum, losses = None, []
    for r in range(WORLD8):
        beg, end, _ = du.shard_range(len(pairs), WORLD8, r)
        mine = [pairs[k % len(pairs)] for k in range(beg, end)]
        ii, jj = [a for a, _ in mine], [b for _, b in mine]
        B = len(mine)
        batch = dict(rgb=rgb.expand(B, 3, S, S).contiguous().numpy(), modal1=masks[ii][:, None].numpy(),
                     modal2=masks[jj][:, None].numpy(), occ_order=(rng.rand(B, 2) < 0.3).astype(np.float32),
                     depth_order=rng.randint(0, 3, B).astype(np.int64), count=np.full(B, 2, np.float32),
                     is_overlap=(rng.rand(B) < 0.5).astype(np.int64))
        t = {k: torch.from_numpy(v.copy()) for k, v in batch.items()}
        m.model.load_state_dict(st0)              # every rank starts the step from the same (broadcast) state
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"], t["occ_order"])
        _, hl, ws = m._fwd_loss_bwd(2 * B, S)       # forward (two BN groups) + loss / 8 + backward, no collective
        m.net._pool.give(ws)
        out = {"loss": hl[0]}
        gsum += m.net.flat_grads
        state = orc.state_from_numpy(sd, prefix="module.")
        ologs, grads = orc.train_step(state, {}, batch, algo, 0.0, 0.0, world_size=WORLD8)
        losses.append((float(out["loss"]), float(ologs["loss"])))
        osum = grads if osum is None else {k: osum[k] + grads[k] for k in grads}
    tol = 1e-3 if dtype == "fp32" else 1e-2
    for got, ref in losses:
        assert abs(got - ref) < tol * abs(ref), (got, ref)
    # summed gradient: per-tensor views of the flat buffer against the oracle's sum
    m.net.flat_grads.copy_(gsum)
    m.net.attach_grads()
    names = orc.param_names(orc.state_from_numpy(sd, prefix="module."))
    num = da = db = 0.0
    errs = []
    for n, p in zip(names, m.net.parameters()):
        a, b = p.grad.detach().cpu().double().reshape(-1), osum[n].double().reshape(-1)
        num += float(a @ b)
        da += float(a @ a)
        db += float(b @ b)
        errs.append(abs(float(a.norm()) - float(b.norm())) / max(float(b.norm()), 1e-30))
    cos, ratio = num / (da * db) ** 0.5, (da / db) ** 0.5
    print("config3 %s: summed-gradient cosine %.5f, norm ratio %.4f, median per-tensor norm err %.2e"
          % (dtype, cos, ratio, float(np.median(errs))))
    if dtype == "fp32":
        assert cos > 0.999 and abs(ratio - 1) < 0.01 and np.median(errs) < 0.02
    else:
        assert cos > 0.97 and abs(ratio - 1) < 0.05


def _run_two_ranks(out_dir, extra_env=None, worker="dp_worker.py"):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    os.makedirs(out_dir, exist_ok=True)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", worker), str(out_dir)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    res = []
    for r in range(2):
        f = os.path.join(out_dir, "rank%d.json" % r)
        assert os.path.exists(f), "rank %d wrote no result\n%s\n%s" % (r, p.stdout[-2000:], p.stderr[-4000:])
        res.append(json.load(open(f)))
    for r in res:
        assert r["ok"], r.get("error")
    assert p.returncode == 0, p.stderr[-4000:]
    return res


def test_two_ranks_step_matches_reference_golden(tmp_path):
    """Two real ranks through ``step()``: RCCL over two GPUs when the box has them, gloo with both ranks on GPU 0
    otherwise.  Each rank checks itself against the reference's two-rank golden (tests/dp_worker.py); here: both
    ranks end with bit-identical weights, and those weights moved.  The worker then takes three more steps -- with
    hipGraphs the overlapped step is captured as one graph per backward stage (the bucket all-reduces launched in
    between) and replayed twice; run again with IO_NO_GRAPH=1 the same four steps are eager: the weights after step 4
    must agree bit for bit between the two forms, and between the ranks."""
    a, b = tmp_path / "graphs", tmp_path / "eager"
    res = _run_two_ranks(str(a))
    print("two-rank step over", res[0]["backend"], "on", res[0]["ngpu"], "GPU(s); staged graphs:", res[0]["staged_graphs"])
    p0, p1 = np.load(a / "params_rank0.npy"), np.load(a / "params_rank1.npy")
    g0, g1 = np.load(a / "grads_rank0.npy"), np.load(a / "grads_rank1.npy")
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)
    assert np.abs(g0).sum() > 0
    res_e = _run_two_ranks(str(b), {"IO_NO_GRAPH": "1"})
    assert res[0]["staged_graphs"] and not res_e[0]["staged_graphs"]
    q0, q1 = np.load(a / "params4_rank0.npy"), np.load(a / "params4_rank1.npy")
    e0 = np.load(b / "params4_rank0.npy")
    assert np.array_equal(q0, q1)
    assert np.array_equal(q0, e0), float(np.abs(q0 - e0).max())
    assert res[0]["more_losses"] == res_e[0]["more_losses"]
    assert not np.array_equal(q0, p0)


def test_two_ranks_depthnet_step(tmp_path):
    """BASELINE configs[4] names 8 GPUs: the data-parallel step of InstaDepthNet_od on two real ranks (tests/
    dp_worker_depth.py; RCCL on two GPUs, gloo with both ranks on GPU 0 otherwise).  Each rank checks broadcast, loss /
    world_size and the all-reduced gradient against the reference golden.  Here: both ranks hold bit-identical weights
    after every step; the four-stage bucketed exchange overlapped with the backward (one hipGraph per stage from the
    third step on) agrees with the flat all-reduce after the whole backward (IO_COMM_OVERLAP=0) -- bit for bit in the
    gradients' exchange arithmetic up to the summation order of the stage-boundary gradients (1e-5 of the update);
    BatchNorm statistics stay rank-local."""
    a, b = tmp_path / "staged", tmp_path / "flat"
    res = _run_two_ranks(str(a), worker="dp_worker_depth.py")
    print("InstaDepthNet_od two-rank step over", res[0]["backend"], "buckets (MB):", res[0]["buckets_mb"],
          "staged graphs:", res[0]["staged_graphs"])
    assert res[0]["overlap"] and res[0]["staged_graphs"]
    assert len(res[0]["buckets_mb"]) == 4 and abs(sum(res[0]["buckets_mb"]) - 610) < 15
    for tag in ("paramsA", "gradsA", "paramsD"):
        x0, x1 = np.load(a / ("%s_rank0.npy" % tag)), np.load(a / ("%s_rank1.npy" % tag))
        assert np.array_equal(x0, x1), tag
    assert not np.array_equal(np.load(a / "rmD_rank0.npy"), np.load(a / "rmD_rank1.npy"))
    res_f = _run_two_ranks(str(b), {"IO_COMM_OVERLAP": "0"}, worker="dp_worker_depth.py")
    assert not res_f[0]["overlap"] and not res_f[0]["staged_graphs"]
    gs, gf = np.load(a / "gradsA_rank0.npy").astype(np.float64), np.load(b / "gradsA_rank0.npy").astype(np.float64)
    assert np.sqrt(((gs - gf) ** 2).sum()) <= 1e-5 * np.sqrt((gf ** 2).sum())
    f0, f1 = np.load(b / "paramsD_rank0.npy"), np.load(b / "paramsD_rank1.npy")
    assert np.array_equal(f0, f1)
    assert not np.array_equal(np.load(a / "paramsA_rank0.npy"), np.load(a / "paramsD_rank0.npy"))
    # the VALUES of the captured / replayed staged steps (a stale slice or a wrong unpack range would be the same on both
    # ranks and pass every rank-against-rank check above): steps E..G of the worker -- replays from rank 0's initial state
    # with lr = 0 -- leave the same all-reduced gradient whatever the exchange form, and the staged form run eagerly
    # (IO_NO_GRAPH=1) launches the same kernels in the same order as the stage graphs
    gG, fG = np.load(a / "gradsG_rank0.npy").astype(np.float64), np.load(b / "gradsG_rank0.npy").astype(np.float64)
    assert np.array_equal(np.load(a / "gradsG_rank0.npy"), np.load(a / "gradsG_rank1.npy"))
    assert np.sqrt(((gG - fG) ** 2).sum()) <= 1e-5 * np.sqrt((fG ** 2).sum()), "staged graph replays vs flat exchange"
    c = tmp_path / "staged_eager"
    res_e = _run_two_ranks(str(c), {"IO_NO_GRAPH": "1"}, worker="dp_worker_depth.py")
    assert res_e[0]["overlap"] and not res_e[0]["staged_graphs"]
    eG = np.load(c / "gradsG_rank0.npy").astype(np.float64)
    assert np.sqrt(((gG - eG) ** 2).sum()) <= 1e-6 * np.sqrt((eG ** 2).sum()), "graph replay vs eager staging"


# ---- configs[4]: InstaDepthNet_od at 384 x 384 -----------------------------------------------------------------
W4 = dict(overlap_weight=0.1, distinct_weight=0.9, dorder_weight=1.0, smooth_weight=0.1, occ_order_weight=1.0)


def _depthnet(dtype, S, B, seed=11):
    import instaorder_amd as ia
    g = np.load(os.path.join(GOLDEN, "depthnet_od_S64_B2.npz"), allow_pickle=False)
    spec = [(str(k), tuple(int(d) for d in str(s).split(",") if d), (str(a) or None))
            for k, s, a in zip(g["keys"], g["shapes"], g["aliases"])]
    cfg = dict(algo="InstaDepthNet_od", lr=1e-5, weight_decay=1e-4, optim="SGD", pretrained_weight=None, use_rgb=True,
               dtype=dtype, **W4)
    m = ia.InstaDepthNet_od(cfg, dist_model=False)
    sd = synthetic.make_spec_state_dict(seed, spec, prefix="module.")
    m.model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    batch = synthetic.make_depth_batch(seed + 100, B, S)
    return m, sd, batch


def test_depthnet_staged_overlap_path_on_one_nccl_rank(monkeypatch):
    """_DepthBase._step_overlapped -- the MiDaS step's backward in four autograd stages, one hipGraph per stage, RCCL's
    asynchronous all-reduce of each stage's slice of the flat gradient buffer between the replays -- on ONE nccl rank
    (IO_COMM_OVERLAP=force): eager step, capture, two replays, against the flat path (no staging: one backward, one
    graph) at the summation-order bar of the two-rank test, and bit for bit against the staged form run eagerly."""
    import socket
    import torch.distributed as dist
    from instaorder_amd import distributed_utils as du
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    du.dist_init_("pytorch", backend="nccl")
    try:
        res = {}
        for tag, env in (("staged", {"IO_COMM_OVERLAP": "force"}), ("flat", {"IO_COMM_OVERLAP": "0"}),
                         ("staged_eager", {"IO_COMM_OVERLAP": "force", "IO_NO_GRAPH": "1"}),
                         ("unstaged_eager", {"IO_COMM_OVERLAP": "0", "IO_NO_GRAPH": "1", "IO_DEPTH_STREAMS": "0"})):
            for k in ("IO_COMM_OVERLAP", "IO_NO_GRAPH", "IO_DEPTH_STREAMS"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            m, sd, _ = _depthnet("fp32", 64, 2)
            # lr = 0: every step starts from the same weights, so the forms are compared on what a step COMPUTES (with an
            # update in between, the tiny-batch BatchNorms of this random net turn a 1e-8 weight perturbation into 1e-3 of
            # gradient: tools/depth_staged_check.py with CHECK_LR=1e-3)
            m.optim.param_groups[0]["lr"] = 0.0
            m.switch_to("train")
            grads, losses = [], []
            for i in range(5):                                   # eager, capture + replay, three more replays
                t = {k: torch.from_numpy(v.copy()) for k, v in synthetic.make_depth_batch(500 + i, 2, 64).items()}
                m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"], t["occ_order"])
                losses.append(float(m.step()[1]["loss"]))
                torch.cuda.synchronize()
                grads.append(m.optim.flat_grads.clone())
            res[tag] = (losses, grads, bool(m._dp_graphs))
        assert res["staged"][2] and not res["flat"][2] and not res["staged_eager"][2] and not res["unstaged_eager"][2]
        for i in range(5):
            a, e, b = res["staged"][1][i].double(), res["staged_eager"][1][i].double(), res["flat"][1][i].double()
            u = res["unstaged_eager"][1][i].double()
            # per-stage graphs with the bucket all-reduces in between launch the kernels of the eager staged form in the
            # same order: bit for bit
            assert torch.equal(res["staged"][1][i], res["staged_eager"][1][i]), ("graph replay vs eager staging", i)
            # ... and ONE torch.autograd.backward over the uncut graph (the reference's loss.backward()), eager on one stream
            # or captured as one hipGraph on three, agrees up to the summation order of the stage-boundary gradients
            assert float((a - u).norm()) <= 1e-5 * float(u.norm()), ("staged vs one backward", i)
            assert float((a - b).norm()) <= 1e-5 * float(b.norm()), ("staged vs one graph", i)
        assert np.allclose(res["staged"][0], res["unstaged_eager"][0], rtol=1e-5)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_config4_depthnet_384(dtype):
    from oracle import midas_oracle as mo
    S, B = 384, 2
    m, sd, batch = _depthnet(dtype, S, B)
    t = {k: torch.from_numpy(v.copy()) for k, v in batch.items()}
    st = mo.state_from_numpy(sd, prefix="module.")
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    # eval forward, both mask orders
    m.switch_to("eval")
    with torch.no_grad():
        d, dep, occ = m.model(t["rgb"].cuda(), t["modal1"].cuda(), t["modal2"].cuda())
        od, odep, oocc = mo.forward(st, t["rgb"], t["modal1"], t["modal2"], False)
    assert d.shape == (B, S, S)
    e = (rel_err(d.float().cpu().numpy(), od.numpy()), rel_err(dep.float().cpu().numpy(), odep.numpy()),
         rel_err(occ.float().cpu().numpy(), oocc.numpy()))
    print("config4 %s eval 384^2 rel err (disp, depth logits, occ logits): %.2e %.2e %.2e" % ((dtype,) + e))
    assert max(e) < (1e-3 if dtype == "fp32" else 5e-2)
    # one training step: all five loss terms and the gradient norm against the oracle's two-pass step
    m.switch_to("train")
    m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"], t["occ_order"])
    logs, l = m.step()
    st = mo.state_from_numpy(sd, prefix="module.")
    o1 = mo.forward(st, t["rgb"], t["modal1"], t["modal2"], True)
    o2 = mo.forward(st, t["rgb"], t["modal2"], t["modal1"], True)
    ologs, total = mo.losses(o1, o2, t, W4)
    total.backward()
    tol = 2e-3 if dtype == "fp32" else 3e-2
    for k, v in logs.items():
        ref = float(ologs[k])
        ktol = 5e-3 if (k == "loss_disp_order" and dtype == "fp32") else tol
        if k == "loss_disp_order" and dtype == "bf16":
            continue                  # a count of pixel comparisons between near-equal disparities: not a bf16 target
        assert abs(float(v) - ref) <= ktol * max(1.0, abs(ref)), (k, float(v), ref)
    uniq = {id(p): p for p in st.values() if p.requires_grad and p.grad is not None}      # aliased keys share a tensor
    ref_gn = float(sum(float(p.grad.double().pow(2).sum()) for p in uniq.values()) ** 0.5)
    gn = float(m.optim.flat_grads.double().norm())
    print("config4 %s step 384^2: gradient norm %.5f (oracle %.5f)" % (dtype, gn, ref_gn))
    assert abs(gn - ref_gn) < (0.03 if dtype == "fp32" else 0.1) * ref_gn
    # every parameter tensor's gradient against the oracle's (PyTorch-CPU fp32, the reference's arithmetic) at this size:
    # the decoder's filter gradients run here on 24 / 48 / 96 / 192-wide maps (the Winograd filter-gradient kernels at
    # widths that are not powers of two), which the 64 x 64 anchor test of test_gpu_midas.py does not reach
    g = np.load(os.path.join(GOLDEN, "depthnet_od_S64_B2.npz"), allow_pickle=False)
    names = [str(n) for n in g["names"]]
    errs, num, den = [], 0.0, 0.0
    for n, (off, k) in zip(names, m.optim._spans):
        p_ = st.get(n)
        if p_ is None or p_.grad is None:
            continue
        ref = p_.grad.double().reshape(-1)
        gh = m.optim.flat_grads[off:off + k].double().cpu()
        nr = float(ref.norm())
        if nr < 1e-12 * ref_gn:
            continue
        errs.append((float((gh - ref).norm()) / nr, n))
        num += float((gh - ref).norm() ** 2)
        den += nr ** 2
    errs.sort(reverse=True)
    glob = (num / den) ** 0.5
    # (bf16: reported only -- on this random-weight state single tensors of the deepest encoder layers differ by O(1), the
    # chaos the bf16 tests of test_gpu_bf16.py damp with bn3 x 0.1; the bf16 kernels are held to the oracle there)
    per_bound, glob_bound = (3e-2, 5e-3) if dtype == "fp32" else (2.0, 0.3)       # measured fp32: worst 9.7e-3, global 1.3e-3
    over = [e for e in errs if e[0] > per_bound]
    print("config4 %s step 384^2: %d gradient tensors vs the oracle: global rel L2 %.2e, worst %.2e (%s), median %.2e, %d over %.0e"
          % (dtype, len(errs), glob, errs[0][0], errs[0][1], errs[len(errs) // 2][0], len(over), per_bound))
    assert len(errs) > 300 and glob < glob_bound, glob
    assert len(over) <= len(errs) // 50, over[:10]
