"""Evaluation metrics of PDGN's test phase (evaluation/evaluation_metrics.py:26-200) on the fused
MI355X kernels.

The reference loops over the samples in Python and, for each, expands it against every reference
batch before calling the CD / EMD extensions (``_pairwise_EMD_CD_`` :85-121; "may take about 2 hours",
README.md:47).  Here the (N_sample x N_ref) matrices are filled by the pair-list kernels
``pdgn_chamfer_gram_indexed`` / ``pdgn_emd_cost_indexed``: no expanded clouds, no (B,N,N) or (B,m,n)
temporaries, tens of thousands of pairs per launch.  MMD / COV / 1-NNA are the reference's small
torch reductions over those matrices.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check, ptr, require, stream_of

F32, I32 = torch.float32, torch.int32
_MAX_PAIRS = 32768          # per launch (grid.y limit of the Chamfer kernel, scratch of the EMD kernel)


def _pair_lists(S, R, start, stop, device):
    p = torch.arange(start, stop, device=device, dtype=torch.int64)
    return (p // R).to(I32).contiguous(), (p % R).to(I32).contiguous()


def shard_pairs(total, fill, group=None):
    """Rows of the evaluation matrices are independent: rank r of the process group fills the
    r-th contiguous slice of the flat pair index space with `fill(start, stop) -> tuple of (stop-start,) tensors`, the
    slices are all-gathered (a few MB) and every rank returns the full-length tensors.  Every rank must call this with
    the same `total`; without a process group it is `fill(0, total)`."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        return tuple(fill(0, total))
    rank = dist.get_rank(group)
    chunk = -(-total // world)
    lo, hi = min(total, rank * chunk), min(total, (rank + 1) * chunk)
    full = []
    for t in fill(lo, hi):
        pad = t.new_zeros(chunk)
        pad[:hi - lo] = t
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        full.append(torch.cat(parts)[:total])
    return tuple(full)


def pairwise_emd_cd(sample_pcs, ref_pcs, batch_size=None, shard_over_ranks=False, group=None):
    """_pairwise_EMD_CD_ (:85-121): all_cd, all_emd of shape (N_sample, N_ref).
    CD = mean_i min_j P + mean_j min_i P with the Gram-form P of distChamfer (:35-45);
    EMD = match_cost / N (:26-31).  `batch_size` is accepted for signature parity and ignored.
    shard_over_ranks: every rank of `group` holds the same two sets and computes 1/world of the pairs (`shard_pairs`)."""
    require(sample_pcs, "sample_pcs", F32, 3)
    require(ref_pcs, "ref_pcs", F32, 3)
    S, N, _ = sample_pcs.shape
    R, M, _ = ref_pcs.shape
    dev = sample_pcs.device
    L = _lib.lib()

    def fill(lo, hi):
        cd = torch.empty(hi - lo, dtype=F32, device=dev)
        emd = torch.empty(hi - lo, dtype=F32, device=dev)
        for start in range(lo, hi, _MAX_PAIRS):
            stop = min(hi, start + _MAX_PAIRS)
            npairs = stop - start
            ia, ib = _pair_lists(S, R, start, stop, dev)
            minx = torch.empty((npairs, N), dtype=F32, device=dev)
            miny = torch.empty((npairs, M), dtype=F32, device=dev)
            argx = torch.empty((npairs, N), dtype=I32, device=dev)
            argy = torch.empty((npairs, M), dtype=I32, device=dev)
            check(L.pdgn_chamfer_gram_indexed(npairs, N, M, 3, ptr(sample_pcs), ptr(ia), ptr(ref_pcs), ptr(ib),
                                              ptr(minx), ptr(argx), ptr(miny), ptr(argy), stream_of(sample_pcs)),
                  "pdgn_chamfer_gram_indexed")
            # distChamfer returns (P.min(1), P.min(2)) = (per ref point, per sample point); :108 adds their means
            cd[start - lo:stop - lo] = miny.mean(dim=1) + minx.mean(dim=1)
            L.pdgn_emd_cost_temp_floats.restype = ctypes.c_longlong
            temp = torch.empty(L.pdgn_emd_cost_temp_floats(ctypes.c_longlong(npairs), N, M), dtype=F32, device=dev)
            out = torch.empty((npairs,), dtype=F32, device=dev)
            check(L.pdgn_emd_cost_indexed(npairs, N, M, ptr(sample_pcs), ptr(ia), ptr(ref_pcs), ptr(ib), ptr(temp),
                                          ptr(out), stream_of(sample_pcs)), "pdgn_emd_cost_indexed")
            emd[start - lo:stop - lo] = out / float(N)
        return cd, emd

    cd, emd = shard_pairs(S * R, fill, group) if shard_over_ranks else fill(0, S * R)
    return cd.view(S, R), emd.view(S, R)


def emd_cd(sample_pcs, ref_pcs, batch_size=None, reduced=True):
    """EMD_CD (:48-82): paired (i, i) Chamfer and EMD."""
    S = sample_pcs.shape[0]
    assert S == ref_pcs.shape[0], "REF:%d SMP:%d" % (ref_pcs.shape[0], S)
    from .losses import chamfer_min
    from .structural_losses import emd_cost
    minx, miny = chamfer_min(sample_pcs.contiguous(), ref_pcs.contiguous())
    cd = miny.mean(dim=1) + minx.mean(dim=1)
    emd = emd_cost(sample_pcs.contiguous(), ref_pcs.contiguous()) / float(sample_pcs.shape[1])
    if reduced:
        cd, emd = cd.mean(), emd.mean()
    return {"MMD-CD": cd, "MMD-EMD": emd}


def lgan_mmd_cov(all_dist):
    """:157-169."""
    N_sample, N_ref = all_dist.size(0), all_dist.size(1)
    min_val_fromsmp, min_idx = torch.min(all_dist, dim=1)
    min_val, _ = torch.min(all_dist, dim=0)
    cov = torch.tensor(float(min_idx.unique().view(-1).size(0)) / float(N_ref)).to(all_dist)
    return {"lgan_mmd": min_val.mean(), "lgan_cov": cov, "lgan_mmd_smp": min_val_fromsmp.mean()}


def knn(Mxx, Mxy, Myy, k, sqrt=False):
    """1-NN two-sample test (:123-154)."""
    n0, n1 = Mxx.size(0), Myy.size(0)
    label = torch.cat((torch.ones(n0), torch.zeros(n1))).to(Mxx)
    M = torch.cat((torch.cat((Mxx, Mxy), 1), torch.cat((Mxy.transpose(0, 1), Myy), 1)), 0)
    if sqrt:
        M = M.abs().sqrt()
    inf_diag = torch.diag(float("inf") * torch.ones(n0 + n1).to(Mxx))
    _, idx = (M + inf_diag).topk(k, 0, False)
    count = torch.zeros(n0 + n1).to(Mxx)
    for i in range(k):
        count = count + label.index_select(0, idx[i])
    pred = torch.ge(count, (float(k) / 2) * torch.ones(n0 + n1).to(Mxx)).float()
    s = {"tp": (pred * label).sum(), "fp": (pred * (1 - label)).sum(),
         "fn": ((1 - pred) * label).sum(), "tn": ((1 - pred) * (1 - label)).sum()}
    s.update({"precision": s["tp"] / (s["tp"] + s["fp"] + 1e-10), "recall": s["tp"] / (s["tp"] + s["fn"] + 1e-10),
              "acc_t": s["tp"] / (s["tp"] + s["fn"] + 1e-10), "acc_f": s["tn"] / (s["tn"] + s["fp"] + 1e-10),
              "acc": torch.eq(label, pred).float().mean()})
    return s


def compute_all_metrics(sample_pcs, ref_pcs, batch_size=None, accelerated_cd=False, shard_over_ranks=False, group=None):
    """compute_all_metrics (:172-200): MMD / COV (CD and EMD) and 1-NNA from three all-pairs passes.
    shard_over_ranks: all ranks call with the same sets, each computes 1/world of every matrix (`shard_pairs`)."""
    results = {}
    kw = {"shard_over_ranks": shard_over_ranks, "group": group}
    M_rs_cd, M_rs_emd = pairwise_emd_cd(sample_pcs, ref_pcs, **kw)
    results.update({"%s-CD" % k: v for k, v in lgan_mmd_cov(M_rs_cd.t()).items()})
    results.update({"%s-EMD" % k: v for k, v in lgan_mmd_cov(M_rs_emd.t()).items()})
    M_rr_cd, M_rr_emd = pairwise_emd_cd(ref_pcs, ref_pcs, **kw)
    M_ss_cd, M_ss_emd = pairwise_emd_cd(sample_pcs, sample_pcs, **kw)
    results.update({"1-NN-CD-%s" % k: v for k, v in knn(M_rr_cd, M_rs_cd, M_ss_cd, 1).items() if "acc" in k})
    results.update({"1-NN-EMD-%s" % k: v for k, v in knn(M_rr_emd, M_rs_emd, M_ss_emd, 1).items() if "acc" in k})
    return results


# ---------------------------------------------------------------------------- JSD (evaluation_metrics.py:206-321)
def unit_cube_grid_point_cloud(resolution, clip_sphere=False, device="cpu"):
    """Cell centres of a resolution^3 grid over the unit cube (:206-224); clip_sphere keeps |c| <= 0.5."""
    spacing = 1.0 / float(resolution - 1)
    ax = torch.arange(resolution, dtype=F32, device=device) * spacing - 0.5
    grid = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1)
    if clip_sphere:
        grid = grid.reshape(-1, 3)
        grid = grid[grid.double().norm(dim=1) <= 0.5]
    return grid, spacing


def occupancy_grid_counters(pclouds, grid_resolution, in_sphere=False):
    """grid_counters and per-cell Bernoulli counts of entropy_of_occupancy_grid (:242-283): every point votes for
    its nearest grid cell.  The reference loops over clouds with sklearn's NearestNeighbors; here ALL points go
    through one brute-force 1-NN launch (pdgn_knnquery) and two bincounts."""
    from . import pointops
    require(pclouds, "pclouds", F32, 3)
    S, N, _ = pclouds.shape
    grid, _ = unit_cube_grid_point_cloud(grid_resolution, in_sphere, device=pclouds.device)
    grid = grid.reshape(1, -1, 3).contiguous()
    G = grid.shape[1]
    idx = pointops.knnquery(1, grid, pclouds.reshape(1, S * N, 3).contiguous()).view(S, N).long()
    counters = torch.bincount(idx.reshape(-1), minlength=G).double()
    cloud = torch.arange(S, device=pclouds.device).view(S, 1).expand(S, N)
    pairs = torch.unique(cloud.reshape(-1) * G + idx.reshape(-1))          # one vote per (cloud, cell)
    bernoulli = torch.bincount(pairs % G, minlength=G).double()
    return counters, bernoulli


def _entropy(p, base=None):
    p = p / p.sum()
    nz = p > 0
    h = -(p[nz] * torch.log(p[nz])).sum()
    return h / torch.log(torch.tensor(float(base), dtype=p.dtype, device=p.device)) if base else h


def entropy_of_occupancy_grid(pclouds, grid_resolution, in_sphere=False):
    """:242-283 -> (mean Bernoulli entropy per cell, grid_counters)."""
    counters, bern = occupancy_grid_counters(pclouds, grid_resolution, in_sphere)
    p = bern[bern > 0] / float(pclouds.shape[0])
    q = 1.0 - p
    h = -(p * torch.log(p)).sum() - (q[q > 0] * torch.log(q[q > 0])).sum()
    return h / counters.numel(), counters


def jensen_shannon_divergence(P, Q):
    """:286-305 (base-2 entropies)."""
    if bool((P < 0).any()) or bool((Q < 0).any()):
        raise ValueError("Negative values.")
    if P.numel() != Q.numel():
        raise ValueError("Non equal size.")
    P_, Q_ = P / P.sum(), Q / Q.sum()
    return _entropy((P_ + Q_) / 2.0, 2) - (_entropy(P_, 2) + _entropy(Q_, 2)) / 2.0


def jsd_between_point_cloud_sets(sample_pcs, ref_pcs, resolution=28):
    """:227-239: JSD between the occupancy distributions of two sets of clouds (unit-sphere grid)."""
    s = entropy_of_occupancy_grid(sample_pcs, resolution, True)[1]
    r = entropy_of_occupancy_grid(ref_pcs, resolution, True)[1]
    return jensen_shannon_divergence(s, r)


# ---------------------------------------------------------------------------- the test phase (models/PDGNet_v2.py:296-331)
@torch.no_grad()
def generate_and_evaluate(generator, ref_pcs, batch_size, normalize=None, rng=None, with_jsd=True):
    """PDGNet_v2.test: draw ceil(N_ref / batch_size) batches of z ~ N(0, 1) (:304 -- sigma 1, unlike training's
    0.2), keep the finest cloud of each, truncate to N_ref, normalise like the reference set (`normalize` =
    the data set's scale mode: shape_unit / shape_bbox / None), then compute_all_metrics (+ 'jsd').
    Returns (generated clouds (N_ref, N, 3), results dict of floats-on-device)."""
    from .data import normalize_point_clouds
    dev = ref_pcs.device
    n_ref = ref_pcs.shape[0]
    gen = []
    for _ in range((n_ref + batch_size - 1) // batch_size):
        z = torch.randn(batch_size, 128, generator=rng, device=dev if rng is None or rng.device.type != "cpu" else "cpu").to(dev)
        gen.append(generator(z)[3].transpose(2, 1).contiguous())
    gen_pcs = torch.cat(gen, dim=0)[:n_ref].contiguous()
    gen_pcs = normalize_point_clouds(gen_pcs, normalize)
    results = compute_all_metrics(gen_pcs, ref_pcs, batch_size)
    if with_jsd:
        results["jsd"] = jsd_between_point_cloud_sets(gen_pcs, ref_pcs)
    return gen_pcs, results
