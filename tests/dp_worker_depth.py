"""One data-parallel rank of the MiDaS-based net on the HIP path, started by tests/test_gpu_configs.py through
torch.distributed.run (never imported by pytest): InstaDepthNet_od at the size of the reference golden
tests/golden/depthnet_od_S64_B2.npz.

Step A: the ranks start from DIFFERENT weights (DistModule must broadcast rank 0's, utils/distributed_utils.py:13-24, 34-37)
and then both feed the golden's batch.  With loss / world_size (models/supervised_order.py:196) and the gradient SUM
(:208 average_gradients) two identical shards reproduce the single-process step of the golden: every logged loss is
half the golden's, the all-reduced gradients and the updated weights are the golden's -- a reference-pinned check that no
rank skips or doubles the exchange.  Steps B..D: different shards per rank; with hipGraphs the second of them is
captured (one graph per backward stage, the bucket all-reduces in between) and the rest are replays.  The caller
compares ranks and exchange forms (IO_COMM_OVERLAP=0: one flat all-reduce after the whole backward).

Backend: nccl (= RCCL) with one GPU per rank when the box has two GPUs, otherwise gloo with both ranks on GPU 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W = dict(overlap_weight=0.1, distinct_weight=0.9, dorder_weight=1.0, smooth_weight=0.1, occ_order_weight=1.0)


def main():
    out_dir = sys.argv[1]
    import numpy as np
    import torch
    import torch.distributed as dist
    import instaorder_amd as ia
    from instaorder_amd import distributed_utils as du, _lib
    from helpers import GOLDEN, synthetic

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    _lib.require_gpu()
    ngpu = torch.cuda.device_count()
    backend = "nccl" if ngpu >= world else "gloo"
    torch.cuda.set_device(rank % ngpu)
    du.dist_init_("pytorch", backend=backend)
    res = {"rank": rank, "backend": backend, "ngpu": ngpu}
    try:
        g = np.load(os.path.join(GOLDEN, "depthnet_od_S64_B2.npz"), allow_pickle=False)
        spec = [(str(k), tuple(int(d) for d in str(s).split(",") if d), (str(a) or None))
                for k, s, a in zip(g["keys"], g["shapes"], g["aliases"])]
        S, B, seed = (int(v) for v in g["meta"])
        cfg = dict(algo="InstaDepthNet_od", lr=float(g["lr"]), weight_decay=float(g["weight_decay"]), optim="SGD",
                   pretrained_weight=None, use_rgb=True, dtype="fp32", **W)
        m = ia.InstaDepthNet_od(cfg, dist_model=False)
        sd = synthetic.make_spec_state_dict(seed + 13 * rank, spec, prefix="module.")      # ranks start DIFFERENT
        m.model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
        m.model = du.DistModule(m.net)                  # broadcasts every state entry from rank 0
        m.world_size = world
        sd0 = synthetic.make_spec_state_dict(seed, spec, prefix="module.")
        for k, v in m.model.state_dict().items():
            assert np.array_equal(v.cpu().numpy(), np.array(sd0[k])), "broadcast: " + k
        sl = m.grad_stage_slices()
        n = m.optim.flat_grads.numel()
        assert len(sl) == 4 and sl[0][1] == n and sl[-1][0] == 0 and all(sl[i][0] == sl[i + 1][1] for i in range(3))
        res["buckets_mb"] = [round((hi - lo) * 4 / 1e6, 1) for lo, hi in sl]

        def feed(t):
            m.set_input(t["rgb"], t["modal1"], t["modal2"], t["depth_order"], t["count"], t["is_overlap"], t["occ_order"])

        m.switch_to("train")
        t = {k: torch.from_numpy(v.copy()) for k, v in synthetic.make_depth_batch(seed + 100, B, S).items()}
        feed(t)
        logs, out = m.step()
        tot = float(du.reduce_tensors(out["loss"].detach().clone()))          # every collective before the first assert
        torch.cuda.synchronize()
        for k, v in logs.items():
            ref = float(g["step_" + k]) / world
            tol = 5e-3 if k == "loss_disp_order" else 2e-3
            assert abs(float(v) - ref) <= tol * max(1.0 / world, abs(ref)), (k, float(v), ref)
        assert abs(tot - float(g["step_loss"])) <= 5e-3 * abs(float(g["step_loss"])), (tot, float(g["step_loss"]))
        # the all-reduced gradient is the golden's single-process gradient
        idx = (np.arange(64, dtype=np.int64) * 2654435761)
        num = den = 0.0
        bad = 0
        pnames = {id(p): n for n, p in m.net.named_parameters()}
        bad_names = []
        for (off, k), ref_norm, ref_s, prm in zip(m.optim._spans, g["grad_norms"], g["grad_samples"], m.optim._params):
            gr = m.optim.flat_grads[off:off + k].double().cpu().numpy()
            got = float(np.sqrt((gr * gr).sum()))
            if ref_norm > 1e-6 and abs(got - ref_norm) > 0.1 * ref_norm:
                bad += 1
                bad_names.append("%s %.3g/%.3g" % (pnames.get(id(prm), "?"), got, float(ref_norm)))
            s = gr[idx % max(k, 1)]
            num += float(((s - ref_s.astype(np.float64)) ** 2).sum())
            den += float((ref_s.astype(np.float64) ** 2).sum())
        res["bad_names"] = bad_names
        assert bad <= len(m.optim._spans) // 50, bad
        assert (num / den) ** 0.5 < 5e-2, (num / den) ** 0.5
        pn = np.array([float(m.optim.flat_params[off:off + k].double().norm()) for off, k in m.optim._spans])
        assert np.allclose(pn, g["step_param_norms"], rtol=2e-4, atol=1e-6)
        np.save(os.path.join(out_dir, "paramsA_rank%d.npy" % rank), m.optim.flat_params.cpu().numpy())
        np.save(os.path.join(out_dir, "gradsA_rank%d.npy" % rank), m.optim.flat_grads.cpu().numpy())
        # B..D: different shards (BatchNorm statistics stay rank-local, gradients are summed)
        t = {k: torch.from_numpy(v.copy()) for k, v in synthetic.make_depth_batch(seed + 200 + rank, B, S).items()}
        losses = []
        for _ in range(3):
            feed(t)
            losses.append(float(m.step()[1]["loss"]))
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "paramsD_rank%d.npy" % rank), m.optim.flat_params.cpu().numpy())
        rm = torch.cat([b.reshape(-1) for k, b in m.model.named_buffers() if k.endswith("running_mean")])
        np.save(os.path.join(out_dir, "rmD_rank%d.npy" % rank), rm.cpu().numpy())
        # E..G: the VALUES of replayed steps, free of the chaos of a randomly initialised net (a 1e-8 perturbation of the
        # weights moves these tiny-batch BatchNorm gradients by 1e-3): rank 0's initial state again on every rank, lr = 0,
        # a fresh shard per step -- the caller compares the last step's all-reduced gradient across exchange forms
        from instaorder_amd import ops
        m.optim.param_groups[0]["lr"] = 0.0
        m.model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd0.items()}, strict=True)
        ops.WEIGHTS_EPOCH[0] += 1
        for j in range(3):
            feed({k: torch.from_numpy(v.copy()) for k, v in synthetic.make_depth_batch(seed + 300 + 10 * j + rank, B, S).items()})
            m.step()
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "gradsG_rank%d.npy" % rank), m.optim.flat_grads.cpu().numpy())
        res.update(ok=True, losses=losses, staged_graphs=bool(getattr(m, "_dp_graphs", None)),
                   overlap=bool(m._overlap_comm))
    except Exception:   # noqa: BLE001
        import traceback
        res.update(ok=False, error=traceback.format_exc())
    json.dump(res, open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
