"""Worker of tests/test_gpu_two_ranks.py: one of TWO data-parallel ranks that share the box's single GPU (gloo moves the
gradient buffers; RCCL refuses two ranks on one device).  Everything else is the product's data-parallel path: per-rank
batches, flat gradient buffers, the early bucket started from inside the backward, the launch list with host points."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _state_tensors(tr):
    ts = []
    for net in [tr.G] + tr.D:
        ts += list(net.parameters()) + list(net.buffers())
    for opt in [tr.optG] + tr.optD:
        for st in opt.state.values():
            ts += [v for v in st.values() if torch.is_tensor(v)]
    return ts


def main(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch
    dev = torch.device("cuda:0")
    B = 4
    reals = synthetic_batch(B, dev, seed=100 + rank)                  # every rank its own batch and noise
    g = torch.Generator().manual_seed(200 + rank)
    z1, z2 = noise(B, dev, g), noise(B, dev, g)
    res = {}
    for mode in ("1", "0"):                                           # early bucket + rest | one flat all-reduce
        os.environ["PDGN_BUCKETS"] = mode
        torch.manual_seed(5 + 17 * rank)                              # different initial weights: rank 0's must win
        tr = PDGNTrainer(device=dev, distributed=True)
        tr.train()
        out = tr.step(reals, z1, z2)
        torch.cuda.synchronize()
        res["early" + mode] = bool(tr.gradG._early_done)
        res["grads" + mode] = torch.cat([p.grad.detach().reshape(-1) for p in tr.G.parameters()]).cpu().numpy()
        res["params" + mode] = torch.cat([p.detach().reshape(-1) for p in tr.G.parameters()]).cpu().numpy()
        res["dparams" + mode] = torch.cat([p.detach().reshape(-1) for d in tr.D for p in d.parameters()]).cpu().numpy()
        res["finite" + mode] = all(bool(torch.isfinite(v)) for v in out.values())
    os.environ["PDGN_BUCKETS"] = "1"
    # the launch list with two ranks: from identical state equal to the eager data-parallel iteration
    torch.manual_seed(5)
    tr = PDGNTrainer(device=dev, distributed=True)
    tr.train()
    ts = _state_tensors(tr)
    tr.step(reals, z1, z2)                                            # (creates the optimizers' state)
    ts = _state_tensors(tr)
    snap = [t.detach().clone() for t in ts]
    eager = {k: v.item() for k, v in tr.step(reals, z1, z2).items()}
    after = torch.cat([p.detach().reshape(-1) for p in tr.G.parameters()]).cpu().numpy()
    tr.capture_list(reals, z1, z2)
    with torch.no_grad():
        for t, v in zip(ts, snap):
            t.copy_(v)
    listed = {k: v.item() for k, v in tr.step_list(reals, z1, z2).items()}
    torch.cuda.synchronize()
    res["list_points"] = len(tr._list_points)
    res["list_params"] = torch.cat([p.detach().reshape(-1) for p in tr.G.parameters()]).cpu().numpy()
    res["eager_params"] = after
    res["loss_eager"] = np.array([eager[k] for k in sorted(eager)])
    res["loss_listed"] = np.array([listed[k] for k in sorted(eager)])
    for _ in range(2):
        out = tr.step_list(reals, z1, z2)
    torch.cuda.synchronize()
    res["list_finite"] = all(bool(torch.isfinite(v)) for v in out.values())
    res["list_params_later"] = torch.cat([p.detach().reshape(-1) for p in tr.G.parameters()]).cpu().numpy()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
    tr._list = None
    torch.cuda.synchronize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
