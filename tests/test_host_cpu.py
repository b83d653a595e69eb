"""CPU-only tests: the C-ABI library loads and exports what include/instaorder_hip.h declares, and the
host-side logic of the package (parameter layout, state_dict contract, optimiser/scheduler contract,
samplers, decision rules, metrics, label mirroring).  No kernel is launched here."""
import os
import re

import numpy as np
import pytest
import torch

from helpers import ROOT, load_golden, orc, synthetic
from instaorder_amd import _lib, distributed_utils, inference, optim, resnet_cls, scheduler, supervised_order


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "instaorder_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(io_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), "missing symbol " + name
        assert name in _lib.SIGNATURES, "no ctypes signature for " + name
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.io_abi_version() == 1


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert _lib.lib().io_device_count() == 0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.require_gpu()
    net = resnet_cls.resnet50_cls(in_channels=5, num_classes=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 5, 64, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net.forward_packed(torch.zeros(1, 64, 64, 8))


@pytest.mark.parametrize("nc", [2, 3, 4, [2, 3]])
def test_state_dict_contract(nc):
    """Keys, order, logical shapes = the reference module's (pinned through synthetic.state_specs, whose
    names the reference-generated goldens carry) and the flat storage is KRSC with the stem padded to 8."""
    net = resnet_cls.resnet50_cls(in_channels=5, num_classes=nc)
    specs = synthetic.state_specs(5, nc)
    sd = net.state_dict()
    assert list(sd.keys()) == [n for n, _, _ in specs]
    for (n, shape, kind), v in zip(specs, sd.values()):
        assert tuple(v.shape) == tuple(shape), n
    names = [n for n, _ in net.named_parameters()]
    assert names == [n for n, _, k in specs if not k.startswith("bn_m") and k not in ("bn_var", "bn_count")]
    n_par = sum(p.numel() for p in net.parameters())
    assert n_par == {2: 23518402, 3: 23520451, 4: 23522500}.get(nc if not isinstance(nc, list) else -1, 23524549)
    src = synthetic.make_state_dict(3, 5, nc)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in src.items()}, strict=True)
    for k, v in net.state_dict().items():
        assert np.array_equal(v.numpy(), src[k]), k
    flat = net.flat_params
    for t in net.plan.tensors:
        if t["kind"] == 0:
            O, I, R, S = t["shape"]
            cs = t["cin_storage"]
            st = flat[t["offset"]:t["offset"] + O * R * S * cs].view(O, R, S, cs).numpy()
            w = src[t["name"]]
            assert np.array_equal(st[..., :I], w.transpose(0, 2, 3, 1)), t["name"]
            assert (st[..., I:] == 0).all()
        assert t["offset"] % 64 == 0
    g = load_golden("o_S64_B4" if nc == 2 else "od_S64_B6") if nc in (2, [2, 3]) else None
    if g is not None:
        assert list(g["names"]) == ["module." + n for n in names]


def test_module_moves_keep_flat_sharing():
    net = resnet_cls.resnet50_cls(in_channels=5, num_classes=2)
    net.float()
    p = net.conv1.weight
    assert p.data_ptr() >= net.flat_params.data_ptr()
    with torch.no_grad():
        net.flat_params.zero_()
    assert float(p.abs().sum()) == 0.0
    with pytest.raises(RuntimeError):
        net.half()
    wrapped = distributed_utils.DistModule.__new__(distributed_utils.DistModule)
    torch.nn.Module.__init__(wrapped)
    wrapped.module = net
    assert list(wrapped.state_dict().keys())[0] == "module.conv1.weight"
    wrapped.train(False)
    assert not net.training


def test_fused_sgd_is_a_torch_optimizer_with_sgd_checkpoint_layout():
    net = resnet_cls.resnet50_cls(in_channels=5, num_classes=[2, 3])
    opt = optim.FusedSGD(net, lr=1e-4, momentum=0.9, weight_decay=1e-4)
    assert isinstance(opt, torch.optim.Optimizer) and len(opt.param_groups) == 1
    assert len(opt.param_groups[0]["params"]) == 163
    sch = scheduler.StepLRScheduler(opt, [2, 4], [0.1, 0.1], 1e-4, [], [], -1)
    sch.step(3)
    assert abs(opt.param_groups[0]["lr"] - 1e-5) < 1e-18
    # a torch.optim.SGD checkpoint (the reference's format) loads into the flat momentum buffer
    ref_params = [torch.nn.Parameter(p.detach().clone().contiguous()) for p in net.parameters()]
    ref = torch.optim.SGD(ref_params, lr=1e-4, momentum=0.9, weight_decay=1e-4)
    for p in ref_params:
        p.grad = torch.full_like(p, 0.5)
    ref.step()
    opt.load_state_dict(ref.state_dict())
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == 163
    assert sd["param_groups"][0]["params"] == list(range(163))
    for i, p in enumerate(ref_params):
        assert torch.equal(sd["state"][i]["momentum_buffer"], ref.state[p]["momentum_buffer"])
    assert sd["param_groups"][0]["momentum"] == 0.9 and sd["param_groups"][0]["weight_decay"] == 1e-4


def test_scheduler_matches_reference_golden():
    g = load_golden("scheduler")
    p = [torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.SGD(p, lr=0.001, momentum=0.9)
    s = scheduler.StepLRScheduler(opt, [32000, 48000], [0.1, 0.1], 0.001, [], [], -1)
    for it, lr in zip(g["its"], g["lrs_plain"]):
        s.step(int(it))
        assert opt.param_groups[0]["lr"] == lr
    opt = torch.optim.SGD(p, lr=0.001, momentum=0.9)
    s = scheduler.StepLRScheduler(opt, [300, 600], [0.1, 0.5], 0.001, [0.004, 0.01], [50, 200], -1)
    for it, lr in zip(g["its_warm"], g["lrs_warm"]):
        s.step(int(it))
        assert opt.param_groups[0]["lr"] == lr
    with pytest.raises(TypeError):
        scheduler.StepLRScheduler(object(), [1], [0.1], 0.1, [], [])
    with pytest.raises(KeyError):
        scheduler.StepLRScheduler(torch.optim.SGD(p, lr=0.1), [1], [0.1], 0.1, [], [], last_iter=5)


def test_samplers():
    data = list(range(103))
    parts = [distributed_utils.DistributedGivenIterationSampler(data, 5, 4, world_size=3, rank=r) for r in range(3)]
    idx = [np.asarray(list(iter(p))) for p in parts]
    assert all(len(i) == 20 for i in idx)
    np.random.seed(0)
    full = np.arange(103)[:60]
    np.random.shuffle(full)
    assert np.array_equal(np.concatenate(idx), full)           # same global shuffle, contiguous slices
    with pytest.raises(RuntimeError):
        iter(parts[0])
    resumed = distributed_utils.DistributedGivenIterationSampler(data, 5, 4, world_size=3, rank=1, last_iter=1)
    assert np.array_equal(np.asarray(list(iter(resumed))), idx[1][8:])
    # more samples requested than the dataset holds: tiled
    big = distributed_utils.DistributedGivenIterationSampler(list(range(7)), 3, 4, world_size=2, rank=1)
    assert len(list(iter(big))) == 12 and max(big.indices) < 7
    seq = [distributed_utils.DistributedSequentialSampler(list(range(10)), world_size=4, rank=r) for r in range(4)]
    got = [list(iter(s)) for s in seq]
    assert got == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 0, 1]]
    for total, world in [(190, 8), (3, 2), (12, 8), (1, 4)]:
        cover = []
        for r in range(world):
            b, e, sub = distributed_utils.shard_range(total, world, r)
            cover += [k % total for k in range(b, e)]
        assert set(cover) == set(range(total)) and len(cover) == sub * world


def test_decision_rules_match_reference_golden():
    g = load_golden("decisions")
    l1 = torch.from_numpy(np.concatenate([g["occ1"], g["dep1"]], 1))
    l2 = torch.from_numpy(np.concatenate([g["occ2"], g["dep2"]], 1))
    d = inference.decide(l1, l2, 2, 3)
    assert (d["i_over_j"].numpy().astype(np.int64) == g["res_od"][:, 1]).all()
    assert (d["j_over_i"].numpy().astype(np.int64) == g["res_od"][:, 2]).all()
    assert (d["depth"].numpy() == g["res_od"][:, 0]).all()
    d = inference.decide(l1[:, :2], l2[:, :2], 2, 0)
    assert (d["i_over_j"].numpy().astype(np.int64) == g["res_o"][:, 0]).all()
    d = inference.decide(l1[:, 2:], l2[:, 2:], 0, 3)
    assert (d["depth"].numpy() == g["res_d"]).all()
    m = inference.decision_margins(torch.cat([l1, l2], 1), "InstaOrderNet_od")
    assert m["occ"].shape == (64, 2) and m["depth"].shape == (64,) and m["occ"][1].max() == 0.0


def test_metrics_match_oracle_and_reference_golden():
    rng = np.random.RandomState(0)
    for n in (3, 5, 20):
        gt = rng.randint(-1, 2, (n, n))
        pr = rng.randint(0, 2, (n, n))
        assert np.allclose(inference.eval_order_recall_precision_f1(pr, gt, 0), orc.recall_precision_f1(pr, gt, 0))
        gd, ov, cnt, od = rng.randint(0, 3, (n, n)), rng.randint(0, 2, (n, n)), rng.randint(1, 4, (n, n)), rng.randint(0, 3, (n, n))
        a, b = inference.eval_depth_order_whdr(od, (gd, ov, cnt)), orc.whdr(od, gd, ov, cnt)
        assert all(abs(a[k][0] - b[k]) < 1e-9 for k in b)
    assert inference.eval_order_recall_precision_f1(np.zeros((2, 2), int), np.zeros((2, 2), int), 0) == (0, 0, 0)
    g = load_golden("plumbing_od")
    items = synthetic.make_images(int(g["meta"][3]) + 400, int(g["meta"][1]), int(g["meta"][2]), int(g["meta"][0]))
    for ii, item in enumerate(items):
        assert np.allclose(inference.eval_order_recall_precision_f1(g["occ_%d" % ii], item["gt_occ"], 0), g["prf_%d" % ii])
        w = inference.eval_depth_order_whdr(g["depth_%d" % ii], (item["gt_depth"], item["gt_overlap"], item["gt_count"]))
        assert np.allclose([w[str(k)][0] for k in g["whdr_keys"]], g["whdr_%d" % ii])


def test_label_mirroring():
    occ = torch.tensor([[0., 1.], [1., 0.], [1., 1.]])
    assert torch.equal(supervised_order._mirror_occ(occ), torch.tensor([[1., 0.], [0., 1.], [1., 1.]]))
    cls = torch.tensor([0, 1, 2, 3, 1])
    assert torch.equal(supervised_order._mirror_classes(cls), torch.tensor([1, 0, 2, 3, 0]))
    assert torch.equal(supervised_order._mirror_classes(cls), orc.mirror_depth(cls))


def test_package_surface():
    import instaorder_amd as ia
    for name in ("InstaOrderNet_o", "InstaOrderNet_od", "InstaOrderNet_d", "OrderNet", "SingleStageModel",
                 "InstaDepthNet_od", "InstaDepthNet_d"):
        assert callable(getattr(ia, name))
    assert callable(ia.backbone.resnet50_cls) and callable(ia.utils.average_gradients)
    assert callable(ia.utils.StepLRScheduler) and callable(ia.utils.DistModule) and callable(ia.utils.init_weights)
    with pytest.raises(KeyError):
        ia.SingleStageModel({"algo": "InstaOrderNet_o", "backbone_arch": "no_such_net", "backbone_param": {}})


def test_init_weights_statistics_match_the_reference():
    """utils/common_utils.py:35-65 as single_stage_model.py:24 applies it (xavier-normal, gain 0.02, on every Conv / Linear
    weight; BatchNorm weight ~ N(1, 0.02), every bias 0): per-tensor rms / mean of this package's init_weights on its own
    resnet50_cls against the statistics of the REFERENCE's function on the reference's network (tests/golden/
    init_stats.npz, 3 pooled draws).  A sample rms over n values has relative spread 1 / sqrt(2n); 6 sigma + the golden's
    own spread is the bar -- a wrong fan (e.g. the x8 storage of conv1 counted as fan-in), gain or distribution is 20 %
    to 50x off."""
    import torch
    import instaorder_amd as ia
    from instaorder_amd import resnet_cls
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "init_stats.npz"))
    want = {str(k): (float(m), float(r), int(n)) for k, m, r, n in zip(g["names"], g["mean"], g["rms"], g["numel"])}
    torch.manual_seed(5)
    net = resnet_cls.resnet50_cls(in_channels=5, num_classes=[2, 3])
    ia.utils.init_weights(net, init_type="xavier")
    got = dict(net.named_parameters())
    assert set(got) == set(want)
    reps = int(g["reps"])
    for k, p in got.items():
        a = p.detach().double().reshape(-1)
        m, r, n = want[k]
        assert a.numel() == n, k
        if r == 0.0:                                  # biases: exactly zero
            assert float(a.abs().max()) == 0.0, k
            continue
        if k.endswith("bn1.weight") or k.endswith("bn2.weight") or k.endswith("bn3.weight") or ".downsample.1.weight" in k:
            sd_ref = max(r * r - m * m, 0.0) ** 0.5   # N(1, 0.02)
            assert abs(float(a.mean()) - 1.0) < 6 * 0.02 / n ** 0.5 + 1e-6, k
            assert abs(float(a.std()) - sd_ref) < 6 * sd_ref * (1 + 1 / reps) ** 0.5 / (2 * n) ** 0.5, (k, float(a.std()), sd_ref)
            continue
        rms = float((a * a).mean().sqrt())
        assert abs(rms - r) < 6 * r * (1 + 1 / reps) ** 0.5 / (2 * n) ** 0.5, (k, rms, r)
        assert abs(float(a.mean())) < 6 * r / n ** 0.5, k
    # the survey's two anchor numbers
    assert abs(want["conv1.weight"][1] - 4.8e-4) < 0.3e-4


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "instaorder_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn


def test_header_is_plain_c_and_links(tmp_path):
    """The drop-in boundary is a C ABI: a C translation unit (gcc -std=c99, no HIP / torch headers) includes
    include/instaorder_hip.h, links libinstaorder_hip.so and runs the planning entry points without a GPU."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    lib = os.path.join(ROOT, "instaorder_amd", "libinstaorder_hip.so")
    assert os.path.exists(lib), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    exe = str(tmp_path / "abi_smoke")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe, lib, "-Wl,-rpath," + os.path.dirname(lib),
           "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    out = r.stdout
    assert "tensors 163 convs 53 bns 53 logits 5 param_floats" in out, out


def test_orig_mode_rounding_rule():
    """utils/data_utils.py:13-17 get_closest_int_multiple_of: a remainder of at least half the multiplier rounds up (so
    exactly half rounds UP), anything below rounds down; multiples stay."""
    from instaorder_amd import inference
    f = inference.get_closest_int_multiple_of
    assert [f(v, 32) for v in (32, 47, 48, 49, 63, 64, 79, 80, 131, 102, 15, 16)] == \
        [32, 32, 64, 64, 64, 64, 64, 96, 128, 96, 0, 32]
    assert f(10, 4) == 12 and f(9, 4) == 8 and f(7, 3) == 9 and f(8, 3) == 9 and f(6, 3) == 6     # (3 // 2 == 1)


def test_bordering_and_pair_selection():
    """inference.py:691-696 / :446-447: neighbour test = one cross dilation of the FIRST mask; checked against
    scipy's binary_dilation with the same structuring element."""
    from scipy import ndimage
    from instaorder_amd import inference
    rng = np.random.RandomState(3)
    cross = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool)
    for _ in range(50):
        a = (rng.rand(12, 15) < 0.15).astype(np.uint8)
        b = (rng.rand(12, 15) < 0.15).astype(np.uint8)
        want = bool(np.any(ndimage.binary_dilation(a.astype(bool), structure=cross) & (b != 0)))
        assert inference.bordering(a, b) == want
    m = np.zeros((3, 8, 8), np.uint8)
    m[0, 1:3, 1:3] = 1
    m[1, 3:5, 1:3] = 1          # touches mask 0 from below
    m[2, 6:8, 6:8] = 1          # far away
    assert inference.select_pairs(m, "all") == [(0, 1), (0, 2), (1, 2)]
    assert inference.select_pairs(m, "nbor") == [(0, 1)]
    with pytest.raises(ValueError):
        inference.select_pairs(m, "some")


def test_heuristic_baselines_and_gt_order_match_reference():
    """inference.py:272-347, 719-754 (area / y-axis baselines, infer_gt_order, eval_order) against matrices produced by
    the reference's own functions (tests/golden/heuristics.npz)."""
    from instaorder_amd import inference
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "heuristics.npz"))
    rd = synthetic.SyntheticReader(88, n_images=4, n_inst=6, empty_every=0)
    for k, sc in enumerate(rd.scenes):
        m = sc["modal"]
        got = {"occ_area_s": inference.infer_occ_order_area(m, "smaller"),
               "occ_area_l": inference.infer_occ_order_area(m, "larger"),
               "occ_y_lo": inference.infer_occ_order_yaxis(m, "lower"),
               "occ_y_hi": inference.infer_occ_order_yaxis(m, "higher"),
               "dep_area_s": inference.infer_depth_order_area(m, "smaller"),
               "dep_area_l": inference.infer_depth_order_area(m, "larger"),
               "dep_y_lo": inference.infer_depth_order_yaxis(m, "lower"),
               "dep_y_hi": inference.infer_depth_order_yaxis(m, "higher"),
               "gt": inference.infer_gt_order(m, z["amodal_%d" % k])}
        for name, v in got.items():
            assert np.array_equal(v, z["%s_%d" % (name, k)]), (name, k)
        ev = inference.eval_order(got["occ_area_s"], got["gt"])
        assert np.allclose(np.asarray(ev[:4], np.float64), z["eval_%d" % k])
        assert np.array_equal(ev[4], z["eval_err_%d" % k])


def _tester_golden():
    import json
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "tester.npz"))
    cfg = json.loads(str(z["data_cfg_json"]))
    rows = [str(v).split("|") for v in z["scenarios"]]
    return z, cfg, rows


@pytest.mark.parametrize("form", ["string", "callable"])
@pytest.mark.parametrize("k", range(4))
def test_evaluate_heuristics_equal_reference_tester(k, form):
    """instaorder_amd.evaluate against the reference's own tools/test.py Tester loops (tests/golden/tester.npz) for the
    model-free methods: per-image matrices and the aggregated P / R / F1 / WHDR means."""
    from instaorder_amd import evaluate
    z, cfg, rows = _tester_golden()
    name, kind, method, mode, algo = rows[k]
    assert algo == "None"
    S, seed, rseed, warm = [int(v) for v in z["meta"]]
    rd = synthetic.SyntheticReader(rseed, n_images=4, n_inst=5, empty_every=0)
    from helpers import baseline_rule          # the same baselines handed over as a callable ordering rule
    res = evaluate.evaluate(None, rd, rd.load_image, dict(cfg, trainval_dataset=kind, patch_or_image=mode),
                            method if form == "string" else baseline_rule(kind, method), return_orders=True)
    for i in range(4):
        occ, dep = res["orders"][i]
        assert np.array_equal(occ if dep is None else dep, z["%s_pred_%d" % (name, i)])
    if kind == "SupOcclusionOrderDataset":
        for key in ("recall", "precision", "f1"):
            assert abs(res[key] - float(z["%s_log_val.%s" % (name, key)])) < 1e-9
    else:
        for key in evaluate.WHDR_KEYS:
            ovl, eq = key.split("_")
            assert abs(res["WHDR_" + key] - float(z["%s_log_val_%s.WHDR_%s" % (name, ovl, eq)])) < 1e-9


def test_ordernet_decisions_and_f1_corners_match_reference_golden():
    """net_forward_OrderNet (inference.py:44-76) for 3- and 4-class heads, and sklearn's zero-division corners of
    eval_order_recall_precision_f1 (inference.py:794-802) -- tests/golden/decisions_ordernet.npz holds what the
    reference's own functions returned (make_golden.py::case_decisions_ordernet)."""
    g = load_golden("decisions_ordernet")
    for K in (3, 4):
        d = inference.decide_ordernet(torch.from_numpy(g["l1_%d" % K]), torch.from_numpy(g["l2_%d" % K]))
        assert (d["i_over_j"].numpy().astype(np.int64) == g["res_%d" % K][:, 0]).all()
        assert (d["j_over_i"].numpy().astype(np.int64) == g["res_%d" % K][:, 1]).all()
    assert inference._heads("OrderNet") == (0, 3)
    for gt, pr, sc in zip(g["gt"], g["pred"], g["scores"]):
        for zi, zd in enumerate((0, 1)):
            assert np.allclose(inference.eval_order_recall_precision_f1(pr, gt, zd), sc[zi]), (gt, pr, zd)
            assert np.allclose(orc.recall_precision_f1(pr, gt, zd), sc[zi])
    # the advisor's case: one pair predicted the wrong way round, zd = 1 -> F1 is 0, not 100
    gt, pr = np.array([[-1, 1], [0, -1]]), np.array([[0, 0], [1, 0]])
    assert inference.eval_order_recall_precision_f1(pr, gt, 1) == (0.0, 0.0, 0.0)


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher environment must start two ranks itself (a torch.distributed.run
    child, before any GPU call in the parent) and hand back the child's exit code.  There is no GPU here, so both ranks
    stop at the loud no-GPU error of the product path -- which is the evidence that two ranks were started."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""           # also where a GPU exists: this test is about the launcher
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    err = p.stderr + p.stdout
    assert err.count("no gfx950") >= 2, err[-3000:]          # one loud refusal per rank


def _od_cfg():
    return dict(algo="InstaOrderNet_od", lr=1e-4, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls",
                backbone_param=dict(in_channels=5, num_classes=[2, 3]), use_rgb=True, overlap_weight=0.1,
                distinct_weight=0.9)


def test_reference_written_checkpoint_digest(tmp_path):
    """tests/golden/checkpoint_od.npz is the digest of what ``load_state(..., resume=True)`` put into this package's
    flat buffers from a file the REFERENCE's own save_state wrote (single_stage_model.py:66-72; made in the build
    container by make_golden.py::case_checkpoint, which also loads a package-written file back into the reference).
    Here the same seeded content, in a file with exactly the structure the reference produced, must give that digest."""
    import instaorder_amd as ia
    from helpers import checkpoint_digest, write_reference_layout_checkpoint
    g = load_golden("checkpoint_od")
    seed, step = (int(v) for v in g["meta"])
    sd, mom, lr, step2 = synthetic.make_checkpoint_state(seed, 5, [2, 3])
    assert step2 == step and len(mom) == int(g["n_state"]) == 163
    path = str(tmp_path / ("ckpt_iter_%d.pth.tar" % step))
    write_reference_layout_checkpoint(path, g, sd, mom, lr, step)
    m = ia.InstaOrderNet_od(_od_cfg(), dist_model=False)
    m.load_state(str(tmp_path), step, resume=True)
    dg = checkpoint_digest(m)
    for k in ("sha_params", "sha_running", "sha_nbt", "sha_momentum"):
        assert dg[k] == str(g[k]), k
    assert np.array_equal(dg["param_norms"], g["param_norms"]) and np.array_equal(dg["momentum_norms"], g["momentum_norms"])
    assert dg["lr"] == float(g["lr"]) == lr
    # and what this package writes has the reference's structure, entry for entry
    out = tmp_path / "out"
    out.mkdir()
    m.save_state(str(out), 9)
    ck = torch.load(str(out / "ckpt_iter_9.pth.tar"), map_location="cpu", weights_only=False)
    assert list(ck["state_dict"].keys()) == [str(k) for k in g["keys"]]
    assert [",".join(str(x) for x in v.shape) for v in ck["state_dict"].values()] == [str(s) for s in g["shapes"]]
    assert [str(v.dtype) for v in ck["state_dict"].values()] == [str(d) for d in g["dtypes"]]
    grp = ck["optimizer"]["param_groups"][0]
    for k in ("lr", "momentum", "dampening", "weight_decay", "nesterov"):
        assert k in grp
    assert grp["params"] == list(range(163)) and len(ck["optimizer"]["state"]) == 163
    assert sorted(ck["optimizer"]["state"][0].keys()) == [str(k) for k in g["state_keys"]]


def test_forced_overlap_without_a_process_group_fails_at_construction(monkeypatch):
    """IO_COMM_OVERLAP=force drives dist.all_reduce between the stage graphs of the step: without an initialised process
    group that must be a clear error when the wrapper is built, not a failure deep inside the first step."""
    import torch.distributed as dist
    import instaorder_amd as ia
    assert not dist.is_initialized()
    monkeypatch.setenv("IO_COMM_OVERLAP", "force")
    with pytest.raises(RuntimeError, match="initialise a process group"):
        ia.InstaOrderNet_od(_od_cfg(), dist_model=False)
    monkeypatch.setenv("IO_COMM_OVERLAP", "1")
    m = ia.InstaOrderNet_od(_od_cfg(), dist_model=False)
    assert m._force_overlap is False and m._overlap_comm is True


def test_depthnet_side_streams_are_decided_lazily_from_device_identity(monkeypatch):
    """midas_net: whether decoder and order branches fork onto side streams is decided at the first forward from the ranks'
    device identities (one process: never shared), not from world_size vs device_count() at construction; IO_DEPTH_STREAMS
    = 0 / force override in both directions."""
    from instaorder_amd import midas_net
    host, ident = midas_net._device_identity()
    assert host and ident == "cpu"                       # no GPU in this suite
    assert midas_net._ranks_share_a_device() is False    # no process group: one process, nothing shared
    for mode, first in (("1", None), ("0", False), ("force", True)):
        monkeypatch.setenv("IO_DEPTH_STREAMS", mode)
        net = midas_net.InstaDepthNet_d()
        assert net._multi_stream is first
        assert net.multi_stream is (True if first is None else first)
        net.multi_stream = False
        assert net.multi_stream is False


def test_no_unprotected_16_byte_store_in_the_256_row_kernel(tmp_path):
    """gfx950 + hipcc: a 12- / 16-byte buffer store with an SGPR scalar offset followed within two instructions by a VALU write of
    one of its data registers loses data on part of the lanes (the compiler's hazard table covers immediate offsets only; found in
    round 6 on conv_p256_kernel's side-output store).  The kernel keeps such offsets in the vector offset;
    tools/scan_store_hazard.py is the check on the generated assembly -- here on csrc/conv_p256.hip, whose stores sit in the k loop."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    asm = str(tmp_path / "conv_p256.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "instaorder_amd", "csrc"),
                           "-I" + os.path.join(root, "include"), "-S", "--cuda-device-only",
                           os.path.join(root, "instaorder_amd", "csrc", "conv_p256.hip"), "-o", asm], stderr=subprocess.DEVNULL)
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "scan_store_hazard.py"), asm], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:]
    assert "hazard hits: 0" in p.stdout
