"""One data-parallel rank of the HIP path, started by tests/test_gpu_configs.py through torch.distributed.run
(never imported by pytest).  Rank r builds InstaOrderNet_o from the seed of the reference's two-rank golden
(tests/golden/ws2_o_S64_B4.npz, made by running the REAL reference on two gloo ranks), takes ITS shard of the pair
batch, and runs ONE ``step()``: DistModule broadcast from rank 0, loss / world_size, flat gradient SUM all-reduce,
SGD (models/supervised_order.py:535-548, utils/distributed_utils.py:13-37).

Backend: nccl (= RCCL) with one GPU per rank when the box has at least two GPUs, otherwise gloo with both ranks on
GPU 0 -- the same host logic and kernels, only the transport differs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir = sys.argv[1]
    import numpy as np
    import torch
    import torch.distributed as dist
    import instaorder_amd as ia
    from instaorder_amd import distributed_utils as du, _lib
    from helpers import load_golden, norms_and_samples, bn_vectors, synthetic, rel_err

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    _lib.require_gpu()
    ngpu = torch.cuda.device_count()
    backend = "nccl" if ngpu >= world else "gloo"
    torch.cuda.set_device(rank % ngpu)
    du.dist_init_("pytorch", backend=backend)
    res = {"rank": rank, "backend": backend, "ngpu": ngpu}
    try:
        g = load_golden("ws2_o_S64_B4")
        S, B, seed, _ = [int(v) for v in g["meta"]]
        cfg = dict(algo="InstaOrderNet_o", lr=1e-3, weight_decay=1e-4, optim="SGD", backbone_arch="resnet50_cls",
                   backbone_param=dict(in_channels=5, num_classes=2), use_rgb=True)
        # ranks start from DIFFERENT weights: the wrapper must broadcast rank 0's
        m = ia.InstaOrderNet_o(cfg, dist_model=False)
        sd = synthetic.make_state_dict(seed + rank * 7, 5, 2, prefix="module.")
        m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
        m.model = du.DistModule(m.net)                     # broadcasts every parameter and BN buffer from rank 0
        m.world_size = world
        sd0 = synthetic.make_state_dict(seed, 5, 2, prefix="module.")
        for k, v in m.model.state_dict().items():
            assert np.array_equal(v.cpu().numpy(), sd0[k]), "broadcast: " + k
        batch = synthetic.make_pair_batch(seed + 200 + rank, B, S)
        t = {k: torch.from_numpy(v.copy()) for k, v in batch.items()}
        m.switch_to("train")
        m.set_input(t["rgb"], t["modal1"], t["modal2"], t["occ_order"])
        out = m.step()
        tot = float(du.reduce_tensors(out["loss"].detach().clone()))     # every collective before the first assert
        loss = float(out["loss"])
        ref = float(g["rank%d_loss" % rank])
        assert abs(loss - ref) < 1e-3 * abs(ref), ("loss", loss, ref)
        params = list(m.net.parameters())
        gn, _ = norms_and_samples([p.grad for p in params])
        gref = g["rank%d_grad_norms" % rank]
        gerr = np.abs(gn - gref) / np.maximum(gref, 1e-30)
        assert np.median(gerr) < 0.02 and gerr.max() < 0.15, ("grad norms", float(np.median(gerr)), float(gerr.max()))
        pn, _ = norms_and_samples(params)
        assert rel_err(pn, g["rank%d_param_norms" % rank]) < 1e-4
        hs = {k[len("module."):]: v.detach().cpu() for k, v in m.model.state_dict().items()}
        rm, rv, _ = bn_vectors(hs)                      # BN statistics stay rank-local
        assert rel_err(rm, g["rank%d_running_mean" % rank]) < 1e-3
        assert rel_err(rv, g["rank%d_running_var" % rank]) < 1e-3
        assert rel_err(rm, g["rank%d_running_mean" % (1 - rank)]) > 1e-3
        assert abs(tot - (float(g["rank0_loss"]) + float(g["rank1_loss"]))) < 2e-3
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "params_rank%d.npy" % rank), m.net.flat_params.cpu().numpy())
        np.save(os.path.join(out_dir, "grads_rank%d.npy" % rank), m.net.flat_grads.cpu().numpy())
        # three more steps on the same inputs: with hipGraphs enabled the first of them is captured (one graph per backward
        # stage, the all-reduces in between) and the next two are replays; IO_NO_GRAPH=1 runs them eagerly.  The caller
        # compares the two forms bit for bit.
        losses = []
        for _ in range(3):
            m.set_input(t["rgb"], t["modal1"], t["modal2"], t["occ_order"])
            losses.append(float(m.step()["loss"]))
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "params4_rank%d.npy" % rank), m.net.flat_params.cpu().numpy())
        res.update(more_losses=losses, staged_graphs=bool(getattr(m, "_dp_graphs", None)))
        res.update(ok=True, loss=loss, grad_median_err=float(np.median(gerr)))
    except Exception:   # noqa: BLE001
        import traceback
        res.update(ok=False, error=traceback.format_exc())
    json.dump(res, open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
