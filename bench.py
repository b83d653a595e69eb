#!/usr/bin/env python3
"""Headline benchmark: instance-pairs/sec, forward + backward + SGD, InstaOrderNet_o (ResNet-50 + order
head) on synthetic 256x256x5 pair batches, fp32 -- BASELINE.json configs[1]: pair-batch 256 per GPU.

One "step" = what the reference's ``InstaOrderNet_o.step()`` does for one batch of B pairs
(supervised_order.py:535-548): both mask orders through the network, BCE loss, backward, gradient
all-reduce (N>1), SGD.  Inputs are resident in HBM before the timed region.  N>1: one process per GPU
(torch.distributed.run), pairs sharded by rank = B pairs per GPU (weak scaling), one flat RCCL
all-reduce of the 94 MB gradient per step.

Prints ONE JSON line (rank 0).  ``roofline`` is for the dominant kernel class (by GPU time), timed with
HIP events on the launch stream inside the timed region; ``cpu_baseline`` times the CPU oracle
(plain-PyTorch restatement of the reference, pinned by tests/golden) on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PAIR_FWD = 21.76e9        # 2 passes x 10.882 GFLOP
FLOP_PER_PAIR_TRAIN = 64.27e9      # 2 x (3 x 10.882 - 0.514) GFLOP, conv+FC MACs x 2
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 chip peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same guide: v_mfma_f32_32x32x16_bf16 dense peak
PEAK_HBM_GBS = 8000.0


def usable_cores():
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_model():
    """CPU model string of the host (core count AND model go into the report)"""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or platform.machine()


# The oracle is plain PyTorch on oneDNN.  Its step on a 12-pair batch stops scaling beyond ~32 threads (the layers of
# the last stages have too little work per thread), and more threads only add wake-up noise -- so the baseline uses up to
# 32 of the usable cores and says so in its line.
CPU_BASELINE_MAX_THREADS = 32


CPU_BASELINE_BUDGET_S = 20.0       # bounded sample: about 10-30 s of CPU work per default bench run (the whole default run stays near 40 s)


def cpu_baseline(sd, S):
    """The CPU oracle (plain-PyTorch restatement of the reference step, pinned by tests/golden) timed on the host as
    SURVEY.md 8(d) prescribes: InstaOrderNet_o fwd+bwd+SGD on 256x256 pairs at B = 12 (BASELINE configs[0]: 4 images x 3
    pairs) and B = 32 (the reference's per-GPU batch, InstaOrderNet_o/config.yaml:49), median of 5 steps after 2 warm-ups
    each -- shortened (and said so in `sample`) only where a slow host would push the sample past its time budget."""
    import torch
    from instaorder_amd import synthetic
    from oracle import resnet_oracle as orc                       # CPU baseline leg only
    cores = min(usable_cores(), CPU_BASELINE_MAX_THREADS)
    torch.set_num_threads(cores)
    state = orc.state_from_numpy(sd, prefix="module.")
    mom = {}
    # calibrate on a 16x cheaper problem (4 pairs at half the side): seconds per full-size pair, pessimistic
    cal = synthetic.make_pair_batch(1999, 4, S // 2)
    for _ in range(2):                                             # (the first call pays oneDNN's primitive set-up)
        t0 = time.perf_counter()
        orc.train_step(state, mom, cal, "InstaOrderNet_o", 1e-3, 1e-4)
        t_pair = time.perf_counter() - t0
    left = CPU_BASELINE_BUDGET_S

    def leg(cb, seed):
        nonlocal left
        plans = [(2, 5), (1, 3), (1, 1)]
        warm, reps = next(((w, r) for w, r in plans if (w + r) * cb * t_pair <= left), (0, 0))
        if reps == 0:
            return None
        cbatch = synthetic.make_pair_batch(seed, cb, S)
        ts = []
        for i in range(warm + reps):
            c0 = time.perf_counter()
            orc.train_step(state, mom, cbatch, "InstaOrderNet_o", 1e-3, 1e-4)
            ts.append(time.perf_counter() - c0)
        left -= sum(ts)
        ts = sorted(ts[warm:])
        med = 0.5 * (ts[(len(ts) - 1) // 2] + ts[len(ts) // 2])
        return {"pairs": cb, "pairs_per_s": cb / med, "s_per_step_median": med, "steps": reps, "warmup": warm}

    small = 12 if 2 * 12 * t_pair <= left else 4               # a host too slow for two 12-pair steps: 4 pairs
    l12 = leg(small, 2000)
    l32 = leg(32, 2001)
    what = lambda l: "%d pairs: median of %d step(s) after %d warm-up(s)" % (l["pairs"], l["steps"], l["warmup"])   # noqa: E731
    return {"value": l12["pairs_per_s"], "unit": "pairs/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "usable_cores": usable_cores(), "b12": l12, "b32": l32,
            "sample": "InstaOrderNet_o fwd+bwd+SGD at %dx%d, PyTorch-CPU fp32 oracle on %s, %d threads of %d usable cores "
                      "(capped at %d: a 12-pair step does not scale further on oneDNN); `value` = %s; b32 = %s "
                      "(SURVEY.md 8(d): median of 5 after 2 at B = 12 and B = 32; fewer steps only where the %.0f s budget "
                      "of this sample would be exceeded)" % (
                          S, S, cpu_model(), cores, usable_cores(), CPU_BASELINE_MAX_THREADS, what(l12),
                          what(l32) if l32 else "skipped (over the time budget on this host)", CPU_BASELINE_BUDGET_S)}


def cpu_baseline_depthnet(algo, S, cfg):
    """CPU oracle of the MiDaS-based nets (oracle/midas_oracle.py, pinned by tests/golden/depthnet_*.npz): one training
    step -- both mask orders forward, the five loss terms, backward -- on a bounded sample of the same workload."""
    import numpy as np
    import torch
    from instaorder_amd import synthetic
    from oracle import midas_oracle as mo                          # CPU baseline leg only
    cores = min(usable_cores(), CPU_BASELINE_MAX_THREADS)
    torch.set_num_threads(cores)
    g = np.load(os.path.join(ROOT, "tests", "golden", "depthnet_od_S64_B2.npz" if algo == "InstaDepthNet_od"
                             else "depthnet_d_S64_B2.npz"), allow_pickle=False)
    spec = [(str(k), tuple(int(d) for d in str(s_).split(",") if d), (str(a) or None))
            for k, s_, a in zip(g["keys"], g["shapes"], g["aliases"])]
    sd = synthetic.make_spec_state_dict(3, spec, prefix="module.")
    variant = "od" if algo == "InstaDepthNet_od" else "d"

    def step(B, S_):
        st = mo.state_from_numpy(sd, prefix="module.")
        t = {k: torch.from_numpy(v.copy()) for k, v in synthetic.make_depth_batch(2000, B, S_).items()}
        o1 = mo.forward(st, t["rgb"], t["modal1"], t["modal2"], True, variant=variant)
        o2 = mo.forward(st, t["rgb"], t["modal2"], t["modal1"], True, variant=variant)
        _, total = mo.losses(o1, o2, t, cfg, variant=variant)
        total.backward()

    t0 = time.perf_counter()
    step(1, S // 2)                                                # calibration on a 4x cheaper problem
    probe = time.perf_counter() - t0
    cb = 2 if probe * 8 < 15.0 else 1
    step(cb, S)                                                    # warm-up
    c0 = time.perf_counter()
    step(cb, S)
    cdt = time.perf_counter() - c0
    return {"value": cb / cdt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "usable_cores": usable_cores(),
            "sample": "%d pair(s) at %dx%d, %s fwd+bwd (two full passes per pair, as the reference), PyTorch-CPU fp32 "
                      "oracle on %s, %d threads of %d usable cores, 1 step after 1 warm-up" % (
                          cb, S, S, algo, cpu_model(), cores, usable_cores())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="pairs per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--algo", default="InstaOrderNet_o")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"], help="fp32 = BASELINE configs[1] (default, "
                    "the headline); bf16 = configs[2]: bf16 activations / GEMM operands, fp32 accumulate and weights")
    ap.add_argument("--mode", default="train", choices=["train", "fwd", "infer"], help="train = fwd+bwd+SGD (the "
                    "BASELINE metric, default); fwd = both directional passes + loss in training mode (batch "
                    "statistics), no backward; infer = the same in eval mode (inference.py's forward)")
    ap.add_argument("--host-inputs", default="", choices=["", "pageable", "pinned", "u8"], help="hand the step HOST "
                    "tensors (as the reference's DataLoader does): the timed region then includes the H2D copies of "
                    "set_input -- the PCIe-inclusive rate quoted in DESIGN.md, never the headline value.  u8 = the "
                    "whole input pipeline inside the timed region: per-item planning on the host (reference dataset "
                    "logic), upload of the uint8 images + masks, crop / resize / normalise on the device "
                    "(instaorder_amd.datasets), over synthetic 480x640 scenes")
    ap.add_argument("--workload", default="pairs", choices=["pairs", "images20"], help="pairs = a synthetic pair "
                    "batch (default); images20 = BASELINE configs[3]: synthetic 20-instance images, all 190 pairs of "
                    "every image enumerated (upper triangle, image-major), the pair list sharded contiguously across "
                    "ranks, each step gathers its pair batch from the instance masks on the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--no-overlap-ab", action="store_true", help="N > 1: skip the extra pass with the flat gradient exchange")
    ap.add_argument("--no-fwd-only", action="store_true", help="skip the forward-only secondary measurement of the default line")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads (bf16 at 256 pairs, fp32 at 32 "
                    "pairs) the default line measures after its timed region")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "exercise the multi-rank path when several ranks must share one GPU)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak (default, the reference's "
                    "semantics: batch_size is per process, InstaOrderNet_o/config.yaml:49) = --batch pairs per GPU; "
                    "strong = --batch is the GLOBAL pair batch, each of the N ranks takes batch / N of it")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, the way the reference starts its
    # own workers from one command (main.py:28-35).  The parent has not touched the GPU (no HIP call, no torch
    # import yet), runs torch.distributed.run as a CHILD process and exits with its code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
            "HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    import numpy as np
    import torch
    import torch.distributed as dist
    import instaorder_amd as ia
    from instaorder_amd import _lib, engine, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    _lib.require_gpu()
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > 1:
        # one process per GPU: RCCL cannot put two ranks on one device, and silently stacking them (local % ndev) would
        # report a "scaling" number measured on fewer GPUs than the line says
        lws = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        assert ndev >= lws, ("bench.py --gpus %d: this node shows %d GPU(s) to rank %d (HIP_VISIBLE_DEVICES=%r); one process "
                             "per GPU needs at least %d.  To exercise the multi-rank path on fewer GPUs use --backend gloo."
                             % (args.gpus, ndev, rank, os.environ.get("HIP_VISIBLE_DEVICES"), lws))
    torch.cuda.set_device(local % ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    elif os.environ.get("IO_COMM_OVERLAP") == "force":
        # the staged data-parallel step on ONE rank (per-stage hipGraphs + the backend's all-reduce of every stage bucket):
        # what a rank of an N-GPU job executes, minus the wire
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=0, world_size=1)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    B, S = args.batch, args.size
    if args.scaling == "strong":
        assert B % world == 0, "--scaling strong: the global batch %d must divide by %d ranks" % (B, world)
        B //= world
    depthnet = args.algo.startswith("InstaDepthNet")
    if depthnet:
        # BASELINE configs[4] (secondary workload): the MiDaS-based net, loss weights of the reference's own
        # experiments/InstaOrder/InstaDepthNet_od/config.yaml, module-default initialisation
        assert args.mode == "train", "InstaDepthNet_*: training step only"
        cfg = dict(algo=args.algo, lr=1e-5, weight_decay=1e-4, optim="SGD", pretrained_weight=None, use_rgb=True,
                   dtype=args.dtype,
                   overlap_weight=0.0, distinct_weight=0.0, dorder_weight=1.0, smooth_weight=0.1, occ_order_weight=0.0)
        model = getattr(ia, args.algo)(cfg, dist_model=world > 1)
        sd = None
    else:
        nc = {"InstaOrderNet_o": 2, "InstaOrderNet_od": [2, 3]}[args.algo]
        cfg = dict(algo=args.algo, lr=1e-3, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls",
                   backbone_param=dict(in_channels=5, num_classes=nc), use_rgb=True, overlap_weight=0.1,
                   distinct_weight=0.9, dtype=args.dtype)
        model = getattr(ia, args.algo)(cfg, dist_model=world > 1)
        sd = synthetic.make_state_dict(1, 5, nc, prefix="module.")          # reference-init statistics
        model.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.switch_to("eval" if args.mode == "infer" else "train")

    # synthetic pair batch: B DISTINCT pairs -- seeded blocks of 32, one seed per block and rank, so that
    # no two samples of the batch (and no two ranks) see the same image or masks; built block by block to bound host memory
    mk = synthetic.make_depth_batch if depthnet else synthetic.make_pair_batch
    blocks = [mk(1000 + 7919 * rank + 31 * q, min(32, B - 32 * q), S) for q in range((B + 31) // 32)]
    dev = {k: torch.cat([torch.from_numpy(b[k]).cuda() for b in blocks]) for k in blocks[0]}
    del blocks
    u8 = None
    if args.host_inputs == "u8":
        assert args.algo == "InstaOrderNet_o" and args.workload == "pairs"
        from instaorder_amd import datasets
        rd = synthetic.SyntheticReader(500 + rank, n_images=32, n_inst=8, max_side=640, min_side=480, empty_every=0)
        dcfg = dict(input_size=S, patch_or_image="patch", data_mean=[0.485, 0.456, 0.406],
                    data_std=[0.229, 0.224, 0.225], load_rgb=True, use_category=False, dataset="InstaOrder",
                    remove_occ_bidirec=0, base_aug=dict(flip=True, shift=[-0.2, 0.2], scale=[0.8, 1.2]))
        u8 = {"ds": datasets.SupOcclusionOrderBatches(dcfg, "train", "InstaOrderNet_o", rd, rd.load_image,
                                                      rng=np.random.RandomState(7 + rank))}
        n_img = len(u8["ds"])
        u8["it"] = datasets.BatchPrefetcher(u8["ds"], ([(s_ * B + k) % n_img for k in range(B)]
                                                      for s_ in range(2 * args.steps + args.warmup + 12)))
    elif args.host_inputs:
        dev = {k: (v.cpu().pin_memory() if args.host_inputs == "pinned" else v.cpu()) for k, v in dev.items()}

    pair_src = None
    if args.workload == "images20":
        assert not depthnet and not args.host_inputs
        from instaorder_amd import distributed_utils, inference
        n_inst = 20
        per_img = n_inst * (n_inst - 1) // 2
        n_img = max(1, (world * B * (args.steps + args.warmup) + per_img - 1) // per_img)
        n_img = min(n_img, 24)                                   # the list wraps around beyond that
        imgs = synthetic.make_images(4242, n_img, n_inst, S)
        rgbs = torch.cat([torch.from_numpy(synthetic.image_mode_inputs(it["image"], it["modal"], S)[0]) for it in imgs]).cuda()
        masks = torch.stack([torch.from_numpy(it["modal"].astype(np.float32)) for it in imgs]).cuda()    # [I,20,S,S]
        plist = [(k, i, j) for k in range(n_img) for (i, j) in inference.upper_pairs(n_inst)]
        beg, end, sub = distributed_utils.shard_range(len(plist), world, rank)
        mine = torch.tensor([plist[q % len(plist)] for q in range(beg, end)], device="cuda")            # [sub, 3]
        pair_src = {"rgbs": rgbs, "masks": masks, "mine": mine, "cursor": 0}

    def gather_pairs():
        p = pair_src
        idx = (torch.arange(B, device="cuda") + p["cursor"]) % p["mine"].shape[0]
        p["cursor"] = (p["cursor"] + B) % p["mine"].shape[0]
        sel = p["mine"][idx]
        dev["rgb"] = p["rgbs"][sel[:, 0]]
        dev["modal1"] = p["masks"][sel[:, 0], sel[:, 1]][:, None]
        dev["modal2"] = p["masks"][sel[:, 0], sel[:, 2]][:, None]

    def one_step():
        if pair_src is not None:
            gather_pairs()
        if u8 is not None:
            model.set_input(*next(u8["it"]))
        elif args.algo == "InstaOrderNet_o":
            model.set_input(dev["rgb"], dev["modal1"], dev["modal2"], dev["occ_order"])
        elif args.algo == "InstaDepthNet_d":
            model.set_input(dev["rgb"], dev["modal1"], dev["modal2"], dev["depth_order"], dev["count"],
                            dev["is_overlap"])
        else:
            model.set_input(dev["rgb"], dev["modal1"], dev["modal2"], dev["depth_order"], dev["count"],
                            dev["is_overlap"], dev["occ_order"])
        if args.mode != "train":
            return model.forward_only()[1]
        return model.step()

    # The timed region runs the PRODUCT path: no profiler, and (training steps) the step replayed from the hipGraph the
    # wrapper captures on its second call with a shape -- what a trainer.py loop gets from its third iteration on.  The
    # capture must not fall into the timed region: with --warmup < 2 the missing calls are made here and reported as
    # config.setup_steps (the driver's command has --warmup 5).
    setup_steps = 0
    for _ in range(args.warmup):
        one_step()
    while args.mode == "train" and args.warmup + setup_steps < 2:
        one_step()
        setup_steps += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # one event per step on the launch stream: per-step times for the median SURVEY.md 8(d) asks for (an event record is
    # a marker packet between two steps, not inside one)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        out = one_step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    ms_median = 0.5 * (step_ms[(len(step_ms) - 1) // 2] + step_ms[len(step_ms) // 2]) if step_ms else None
    hip_graph = bool(getattr(model, "_use_graph", False) and (getattr(model, "_graph", None) is not None
                                                              or getattr(model, "_dp_graphs", None)))
    # Second, separately reported pass for the per-kernel-class times: eager launches (a graph replay has no launch
    # groups to bracket) with one HIP event per launch group on the launch stream.  Its step time is stated next to the
    # timed region's so that "class time <= step time" can be checked against either.
    prof, prof_steps, prof_dt = {}, 0, 0.0
    if not args.no_prof:
        prof_steps = max(1, min(args.steps, 6))
        one_step()                                   # (an eager step after replays: re-warm the allocator pool)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        # the ResNet executor issues nothing but library launches: consecutive groups share an event; the MiDaS nets run
        # op by op with torch kernels in between: own start events, so the class times are kernel times, not wall shares
        engine.prof_begin(share_events=not depthnet)
        p0 = time.perf_counter()
        for _ in range(prof_steps):
            one_step()
        torch.cuda.synchronize()
        prof_dt = time.perf_counter() - p0
        prof = engine.prof_end()
    loss = float(out[1]["loss"] if isinstance(out, tuple) else out["loss"])
    ranks_info = None
    if world > 1:
        # what every rank saw, so that a scaling run explains itself: its view of the group, its own wall time per step,
        # its device -- gathered before the MAX that defines `value`
        mine = {"rank": rank, "world_size": dist.get_world_size(), "backend": dist.get_backend(),
                "device": torch.cuda.current_device(), "device_name": torch.cuda.get_device_name(),
                "ms_per_step": 1e3 * dt / args.steps, "ms_per_step_median": ms_median}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ranks_info = gathered
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # A/B of the exchange form (N > 1, training): the same steps with ONE flat all-reduce after the whole backward
    # (IO_COMM_OVERLAP=0's path) instead of the stage buckets launched under it -- after the timed region, never part of `value`
    overlap_ab = None
    if world > 1 and args.mode == "train" and getattr(model, "_overlap_comm", False) and not args.no_overlap_ab:
        arena = 0 if depthnet else model.net.plan.workspace_bytes(2 * B, S, True)
        if arena < 100 * 2 ** 30:            # the flat path captures its own graph on a second arena: it must fit beside the first
            model._overlap_comm = False
            try:
                for _ in range(3):           # eager, capture, first replay
                    one_step()
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                a0 = time.perf_counter()
                for _ in range(args.steps):
                    one_step()
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                ta = torch.tensor([time.perf_counter() - a0], device="cuda", dtype=torch.float64)
                dist.all_reduce(ta, op=dist.ReduceOp.MAX)
                overlap_ab = {"stage_buckets_under_backward_ms_per_step": 1e3 * dt / args.steps,
                              "flat_after_backward_ms_per_step": 1e3 * float(ta.item()) / args.steps, "steps": args.steps}
            except Exception as ex:          # noqa: BLE001 -- a secondary measurement must not cost the line its `value`
                overlap_ab = {"failed": repr(ex)[:300]}
            finally:
                model._overlap_comm = True
        else:
            overlap_ab = {"skipped": "workspace arena %.0f GiB: no room for the flat path's second graph arena" % (arena / 2 ** 30)}
    pairs_per_s = world * B * args.steps / dt

    flop_per_pair = (FLOP_PER_PAIR_TRAIN if args.mode == "train" else FLOP_PER_PAIR_FWD) * (S / 256.0) ** 2
    if depthnet:
        # 127.2 GMAC per sample and pass at 384^2 in the reference (two full passes per pair).  The
        # pair mode runs the image-only encoder + decoder (127.2 - 2 x 12.24 GMAC of ResNet-50 branches) ONCE per
        # pair and only the order branches twice; the rate below counts the work actually executed.
        branches = 2 * 12.24e9 if args.algo == "InstaDepthNet_od" else 12.24e9
        shared = 127.2e9 - 2 * 12.24e9
        macs = (shared + 2 * branches) if model.PAIR_MODE else 2 * (shared + branches)
        flop_per_pair = 3 * 2 * macs * (S / 384.0) ** 2
    result = {
        "metric": "instance-pairs/sec (fwd+bwd)" if args.mode == "train" else "instance-pairs/sec (%s)" % args.mode,
        "value": pairs_per_s, "unit": "pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "ms_per_step_median": ms_median, "ms_per_step_min": step_ms[0] if step_ms else None,
        "ms_per_step_max": step_ms[-1] if step_ms else None,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32" if args.dtype == "fp32" else "bf16", "data": "synthetic",
        "config": {"workload": "%s, pair-batch %d per GPU at %dx%dx5, %s, %s "
                               "(BASELINE.json configs[%d])" % (args.algo, B, S, S, args.dtype,
                                                                {"train": "fwd+bwd+SGD", "fwd": "forward+loss, train mode",
                                                                 "infer": "forward+loss, eval mode"}[args.mode],
                                                                4 if depthnet else (3 if args.workload == "images20"
                                                                                    else (1 if args.dtype == "fp32" else 2))),
                   "inputs": ("host, " + args.host_inputs) if args.host_inputs else "resident in HBM",
                   "pair_source": "20-instance images, 190 pairs each, sharded by rank" if pair_src else "pair batch",
                   "pairs_per_gpu": B, "input_size": S, "parallelism": "dp%d" % world, "final_loss": loss,
                   "hip_graph": hip_graph, "setup_steps": setup_steps,
                   "inputs_distinct": True,
                   **({"precision_note": "bf16 is not a reference precision: parity bar <= 2e-2 on logits (tests/test_gpu_bf16.py); "
                                         "the 0.1 pp accuracy bar of BASELINE.json is claimed for fp32 only"}
                      if args.dtype == "bf16" else {}),
                   "timed_region": "unprofiled product path (%s); kernel_classes / roofline come from a separate profiled pass "
                                   "of eager steps, see `profiled`" % ("hipGraph replay" if hip_graph else "eager launches"),
                   "collective": None if world == 1 else (
                       "%s all-reduce (SUM) of the %d gradient floats in %d stage buckets (%s MB: heads+layer4, layer3, "
                       "layer2, layer1+stem), each launched when its stage of the backward pass is enqueued" % (
                           args.backend, model.net.flat_grads.numel(), len(model.net.grad_stage_slices()),
                           " / ".join("%.0f" % ((hi - lo) * 4 / 1e6) for lo, hi in model.net.grad_stage_slices()))
                       if (not depthnet and getattr(model, "_overlap_comm", False)) else
                       "%s all-reduce (SUM) of the %d gradient floats in %d stage buckets (%s MB: %s), each launched when its "
                       "stage of the backward pass is enqueued" % (
                           args.backend, model.optim.flat_grads.numel(), len(model.grad_stage_slices()),
                           " / ".join("%.0f" % ((hi - lo) * 4 / 1e6) for lo, hi in model.grad_stage_slices()),
                           ", ".join(model.STAGE_NAMES))
                       if (depthnet and getattr(model, "_overlap_comm", False) and hasattr(model.optim, "gather_grads")) else
                       "%s flat all-reduce, %d floats/step" % (
                           args.backend, (model.optim if depthnet else model.net).flat_grads.numel()))},
        "achieved_tflops_whole_step": pairs_per_s * flop_per_pair / 1e12,
        "mfma_frac_whole_step": pairs_per_s * flop_per_pair / 1e12 / (world * PEAK_FP32_MFMA_TFLOPS),
    }
    if args.dtype == "bf16":
        # every GEMM of the bf16 step multiplies on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): the whole-step figure against
        # the dense bf16 peak, under its own key -- `mfma_frac_whole_step` stays the fp32-peak figure of the headline line
        result["mfma_frac_whole_step"] = None
        result["mfma_frac_whole_step_bf16_peak"] = pairs_per_s * flop_per_pair / 1e12 / (world * PEAK_BF16_MFMA_TFLOPS)
    if world > 1:
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:   # noqa: BLE001
            rccl = None
        result["distributed"] = {"world_size": dist.get_world_size(), "backend": args.backend, "rccl_version": rccl,
                                 "ranks": ranks_info,
                                 "ms_per_step_rank_min": min(r["ms_per_step"] for r in ranks_info),
                                 "ms_per_step_rank_max": max(r["ms_per_step"] for r in ranks_info),
                                 "overlap_ab": overlap_ab}
    # the north-star forward target (BASELINE.json: >= 40 % MFMA on the ResNet-50 pairwise forward at batch 256) as a
    # driver-observed value: the two directional passes + loss in training mode (batch statistics), same inputs, measured
    # AFTER the timed region (rank 0's GPU; every rank runs it so that nobody waits at a collective)
    if args.mode == "train" and not depthnet and not args.no_fwd_only and not args.host_inputs and pair_src is None:
        def fwd_step():
            if args.algo == "InstaOrderNet_o":
                model.set_input(dev["rgb"], dev["modal1"], dev["modal2"], dev["occ_order"])
            else:
                model.set_input(dev["rgb"], dev["modal1"], dev["modal2"], dev["depth_order"], dev["count"],
                                dev["is_overlap"], dev["occ_order"])
            return model.forward_only()
        try:
            for _ in range(2):
                fwd_step()
            torch.cuda.synchronize()
            nf = max(3, min(args.steps, 10))
            f0 = time.perf_counter()
            for _ in range(nf):
                fwd_step()
            torch.cuda.synchronize()
            fdt = (time.perf_counter() - f0) / nf
            fpeak = PEAK_FP32_MFMA_TFLOPS if args.dtype == "fp32" else PEAK_BF16_MFMA_TFLOPS
            ftf = B / fdt * FLOP_PER_PAIR_FWD * (S / 256.0) ** 2 / 1e12
            result["fwd_only"] = {"pairs_per_s": B / fdt, "ms_per_step": 1e3 * fdt, "steps": nf, "tflops": ftf,
                                  "mfma_frac": ftf / fpeak, "peak_tflops": fpeak,
                                  "what": "both directional passes + loss, training mode (batch statistics), no backward; "
                                          "one GPU, measured after the timed region"}
        except Exception as ex:          # noqa: BLE001 -- a secondary measurement must not cost the line its `value`
            result["fwd_only"] = {"failed": repr(ex)[:300]}
    # Secondary workloads on the driver's own command (after the timed region; each leg in its own try: it cannot cost the
    # line its `value`): the bf16 step of BASELINE configs[2] at this batch, and the reference's own per-GPU batch of 32 pairs
    # (experiments/InstaOrder/InstaOrderNet_o/config.yaml:49) in fp32 -- replayed steps of the product path, as the headline.
    headline = (args.algo == "InstaOrderNet_o" and B == 256 and S == 256 and args.mode == "train" and args.dtype == "fp32"
                and world == 1 and not args.host_inputs and pair_src is None)
    if headline and not args.no_secondary:
        def secondary_leg(dtype, b2, nsteps=5):
            cfg2 = dict(cfg, dtype=dtype)
            m2 = ia.InstaOrderNet_o(cfg2, dist_model=False)
            m2.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
            m2.switch_to("train")
            d2 = {k: v[:b2].contiguous() for k, v in dev.items()}

            def st():
                m2.set_input(d2["rgb"], d2["modal1"], d2["modal2"], d2["occ_order"])
                return m2.step()
            for _ in range(3):                       # eager, capture, first replay
                st()
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            for _ in range(nsteps):
                o2 = st()
            torch.cuda.synchronize()
            sdt = (time.perf_counter() - s0) / nsteps
            leg = {"pairs_per_s": b2 / sdt, "ms_per_step": 1e3 * sdt, "steps": nsteps, "pairs_per_gpu": b2, "dtype": dtype,
                   "hip_graph": bool(getattr(m2, "_graph", None) is not None), "final_loss": float(o2["loss"]),
                   "tflops": b2 / sdt * FLOP_PER_PAIR_TRAIN / 1e12}
            # algorithmic HBM bytes of one step (what the launch classes account for, one profiled eager step) over the
            # replayed step time, against the 8 TB/s peak
            try:
                m2._use_graph = False
                st()
                torch.cuda.synchronize()
                engine.prof_begin(share_events=True)
                st()
                torch.cuda.synchronize()
                pr = engine.prof_end()
                by = sum(v["bytes"] for v in pr.values())
                leg["algorithmic_gb_per_step"] = by / 1e9
                leg["hbm_frac"] = by / sdt / 1e9 / PEAK_HBM_GBS
            except Exception as ex:      # noqa: BLE001
                leg["hbm_frac"] = None
                leg["hbm_note"] = repr(ex)[:200]
            del m2
            torch.cuda.empty_cache()
            return leg
        result["secondary"] = {}
        for tag, dtype2, b2 in (("bf16_b256", "bf16", 256), ("fp32_b32", "fp32", 32)):
            try:
                result["secondary"][tag] = secondary_leg(dtype2, b2)
            except Exception as ex:      # noqa: BLE001
                result["secondary"][tag] = {"failed": repr(ex)[:300]}
        if "pairs_per_s" in result["secondary"].get("fp32_b32", {}):
            result["secondary"]["fp32_b32"]["rate_vs_256_pairs"] = result["secondary"]["fp32_b32"]["pairs_per_s"] / pairs_per_s
        result["secondary"]["what"] = ("measured after the timed region on the same GPU: InstaOrderNet_o fwd+bwd+SGD, hipGraph "
                                       "replay; bf16_b256 = the bf16 mode (BASELINE configs[2] arithmetic) at 256 pairs; fp32_b32 = "
                                       "the reference's own per-GPU batch; hbm_frac = algorithmic bytes of the step / step time / 8 TB/s")
    if prof:
        result["profiled"] = {"steps": prof_steps, "ms_per_step": 1e3 * prof_dt / prof_steps, "hip_graph": False,
                              "note": "eager launches + one HIP event per launch group; never the source of `value`"}
        tot_ms = sum(v["total_ms"] for v in prof.values())
        dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        name, d = dom
        avg_ms = d["total_ms"] / d["launches"]
        tfl = d["flops"] / d["launches"] / (avg_ms * 1e-3) / 1e12
        # HBM bytes per launch from the newest committed PMC run of this command (profiles/rNN_pmc_traffic.json, made
        # by tools/collect_traffic.sh + tools/make_profiles.py; not live -- the file names the commit it was measured at)
        # A PMC file is quoted only while the kernel sources are the ones it was measured on (csrc_sha = _lib.csrc_digest()
        # recorded by tools/make_profiles.py); after any change under csrc/ traffic is null until the counters are re-collected.
        traffic, traffic_src = None, None
        digest = _lib.csrc_digest()
        try:
            import glob
            f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
            tj = json.load(open(f))
            headline = args.algo == "InstaOrderNet_o" and B == 256 and S == 256 and args.mode == "train" and args.dtype == "fp32"
            if not headline:
                pass            # the counters were collected on the headline configuration only: no traffic figure for other lines
            elif tj.get("kernel") == name and tj.get("csrc_sha") == digest:
                traffic = tj["bytes_per_launch_corrected"]
                traffic_src = "%s (measured at commit %s, csrc %s)" % (os.path.basename(f), tj.get("commit"), digest)
            elif tj.get("kernel") == name:
                traffic_src = "stale: %s was measured on other kernel sources (csrc %s, now %s)" % (
                    os.path.basename(f), tj.get("csrc_sha", "unrecorded"), digest)
        except Exception:
            pass
        # the wgrad kernel runs the fp32 MFMA in both modes; the NT kernel (fwd / dgrad) follows --dtype
        peak = PEAK_BF16_MFMA_TFLOPS if (args.dtype == "bf16" and "wgrad" not in name) else PEAK_FP32_MFMA_TFLOPS
        if args.dtype == "bf16":
            # the bf16 step has its own PMC file (tools/collect_traffic_bf16.sh); it holds for the headline configuration
            traffic, traffic_src = None, None
            try:
                f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_bf16.json")))[-1]
                tj = json.load(open(f))
                # the class's launches run on four kernel families (conv_p256 / conv_halo3 / stem_halo / conv_nt_kernel) that
                # the launch classes cut differently: the counters give ONE ratio of measured to algorithmic bytes over all
                # forward / data-gradient launches of the step; `traffic` = that ratio x this class's algorithmic bytes per launch
                allf = tj["kernels"].get("_forward_and_data_gradient_launches")
                if allf and name.startswith("conv_nt_kernel") and args.algo == "InstaOrderNet_o" and B == 256 and S == 256 \
                        and args.mode == "train" and tj.get("csrc_sha") == digest:
                    traffic = allf["traffic_ratio"] * d["bytes"] / d["launches"]
                    traffic_src = ("%s (commit %s): measured / algorithmic bytes = %.3f over all %d forward and data-gradient "
                                   "launches of a step (kernel families %s), applied to this class's algorithmic bytes per launch"
                                   % (os.path.basename(f), tj.get("commit"), allf["traffic_ratio"],
                                      int(round(allf["launches_per_step"])), ", ".join(x.split("<")[0] for x in allf["families"])))
                elif allf and tj.get("csrc_sha") != digest:
                    traffic_src = "stale: %s was measured on other kernel sources (csrc %s, now %s)" % (
                        os.path.basename(f), tj.get("csrc_sha", "unrecorded"), digest)
            except Exception:
                pass
        gbs = d["bytes"] / d["launches"] / (avg_ms * 1e-3) / 1e9
        tfl_exec = d.get("flops_executed", d["flops"]) / d["launches"] / (avg_ms * 1e-3) / 1e12
        result["roofline"] = {"bound": "mfma", "kernel": name, "achieved": tfl, "peak": peak, "achieved_executed": tfl_exec,
                              "unit": "TFLOP/s", "frac": tfl / peak, "traffic": traffic, "traffic_source": traffic_src,
                              "launches": d["launches"], "avg_launch_ms": avg_ms,
                              "flops_per_launch": d["flops"] / d["launches"],
                              "share_of_gpu_time": d["total_ms"] / tot_ms,
                              # the other roof, always: algorithmic bytes of the same launches against HBM
                              "hbm": {"achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS}}
        if not depthnet and args.mode == "train" and args.dtype == "fp32" and name.startswith("conv_nt_kernel"):
            # what the FLOP rate of this class does not show ("Round 3")
            result["roofline"]["note"] = (
                "algorithmic conv FLOPs only: about half of this class's launches per step also evaluate a BatchNorm pass on "
                "their A operand while it is staged (second 16-byte load per chunk, per-channel fma, side output) -- work "
                "that used to be separate HBM-bound kernels (bn_bwd + bn_apply 30.8 -> 8 ms per step) and adds bytes, not "
                "algorithmic FLOPs, to these launches; mfma_frac_whole_step is the figure that reflects the trade")
        if args.dtype == "bf16" and "wgrad" not in name and gbs / PEAK_HBM_GBS > tfl / peak:
            # bf16 forward / data-gradient GEMMs of ResNet-50 sit below the bf16 ridge (312 FLOP/B): HBM is the roof that
            # binds, the MFMA fraction is reported alongside
            result["roofline"].update(bound="hbm", achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s", frac=gbs / PEAK_HBM_GBS,
                                      mfma={"achieved": tfl, "peak": peak, "unit": "TFLOP/s", "frac": tfl / peak})
        # tflops: ALGORITHMIC flops (the direct convolution's count) / time; tflops_executed: what the
        # matrix pipes multiplied -- the Winograd row forms of the 3x3 stride-1 layers execute 1/2 (F(4,3)) or 2/3 (F(2,3)) of
        # the algorithmic products, so their algorithmic rate can exceed the MFMA peak while the executed rate cannot
        result["kernel_classes"] = {
            k: {"launches": v["launches"], "ms_per_step": v["total_ms"] / prof_steps,
                "tflops": (v["flops"] / (v["total_ms"] * 1e-3) / 1e12) if v["flops"] else None,
                "tflops_executed": (v.get("flops_executed", v["flops"]) / (v["total_ms"] * 1e-3) / 1e12) if v["flops"] else None,
                "gbs": v["bytes"] / (v["total_ms"] * 1e-3) / 1e9}
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
        fl_alg = sum(v["flops"] for v in prof.values())
        fl_exe = sum(v.get("flops_executed", v["flops"]) for v in prof.values())
        if fl_alg > 0:
            result["executed_flop_fraction"] = fl_exe / fl_alg
            if result.get("mfma_frac_whole_step") is not None:
                result["mfma_frac_whole_step_executed"] = result["mfma_frac_whole_step"] * fl_exe / fl_alg

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "train":
        result["cpu_baseline"] = cpu_baseline_depthnet(args.algo, S, cfg) if depthnet else cpu_baseline(sd, S)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
