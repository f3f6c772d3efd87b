"""CPU, world_size 2, gloo: the N>1 path of the G+D step (flat-buffer gradient all-reduce, the
sum-vs-mean loss scaling of SURVEY.md section 8-e) with torch / C-oracle stand-ins for the HIP ops."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build_trainer(distributed, seed=7):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import pdgnet_ref
    from pdgn_amd import deconv
    from pdgn_amd.trainer import PDGNTrainer
    from torch_standins import EdgeGatherSumTorch, linear_cl_torch, softmax_slots_permute_torch, bn_softmax_slots_permute_torch, bilateral_weighting_torch, bn_act_maxpool_torch, bn_act_torch, feature_knn_torch, patch_losses
    deconv.EdgeGatherSum, deconv.bn_act, deconv.feature_knn = EdgeGatherSumTorch, bn_act_torch, feature_knn_torch
    deconv.linear_cl = linear_cl_torch
    deconv.bn_act_maxpool = bn_act_maxpool_torch
    deconv.flush_bn_counters = lambda: None
    deconv.softmax_slots_permute = softmax_slots_permute_torch
    deconv.bn_softmax_slots_permute = bn_softmax_slots_permute_torch
    deconv.bilateral_weighting = bilateral_weighting_torch
    torch.manual_seed(seed)                                # identical initial weights on every rank (unless a test varies it)
    tr = PDGNTrainer(device="cpu", base_points=16, distributed=distributed)
    patch_losses(None)
    tr.train()
    return tr


def _batch(rank, B=8):
    g = torch.Generator().manual_seed(100 + rank)
    reals = [torch.randn(B, 3, n, generator=g) for n in (32, 64, 128, 256)]
    return reals, torch.randn(B, 128, generator=g) * 0.2, torch.randn(B, 128, generator=g) * 0.2


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = _build_trainer(distributed=True)
    reals, z1, z2 = _batch(rank)
    out = tr.step(reals, z1, z2)
    flat = torch.cat([p.detach().reshape(-1) for p in tr.G.parameters()] +
                     [p.detach().reshape(-1) for d in tr.D for p in d.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ok_sync = all(torch.equal(gathered[0], g) for g in gathered)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), sync=ok_sync, params=flat.numpy(),
             gradG=tr.gradG.buf.numpy(), g_loss=out["g_loss"].item(), finite=bool(torch.isfinite(flat).all()))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_step_gloo(request):
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        r = [dict(np.load(os.path.join(d, "rank%d.npz" % i))) for i in range(world)]
    assert all(bool(x["sync"]) and bool(x["finite"]) for x in r)
    np.testing.assert_array_equal(r[0]["params"], r[1]["params"])          # replicas stay identical
    np.testing.assert_array_equal(r[0]["gradG"], r[1]["gradG"])            # the reduced gradient
    # expected reduced G gradient from two single-process runs: mean over ranks of
    # grad(adv_r + 0.1 * world * similar_r)  (MSE is a batch mean, the shape loss a batch sum)
    import torch.nn.functional as F
    request.addfinalizer(lambda n=torch.get_num_threads(): torch.set_num_threads(n))
    torch.set_num_threads(2)                               # same reduction order as the workers
    grads = []
    for rank in range(world):
        tr = _build_trainer(distributed=False)
        reals, z1, z2 = _batch(rank)
        B = z1.shape[0]
        ones, zeros = torch.ones(B, 1), torch.zeros(B, 1)
        with torch.no_grad():
            fakes = tr.G(z1)
        # D updates are needed first (the G loss sees the updated discriminators), with the
        # rank-mean of the D gradients: emulate by averaging the two ranks' D grads
        grads.append((tr, reals, fakes, z2, ones, zeros))
    # D step with averaged gradients
    for i in range(4):
        for tr, reals, fakes, z2, ones, zeros in grads:
            tr.gradD[i].begin()
            ((F.mse_loss(tr.D[i](reals[i]), ones) + F.mse_loss(tr.D[i](fakes[i]), zeros)) / 2.0).backward()
            tr.gradD[i].pack()
        mean = (grads[0][0].gradD[i].buf + grads[1][0].gradD[i].buf) / world
        for tr, *_ in grads:
            tr.gradD[i].buf.copy_(mean)
            tr.optD[i].step()
    gG = []
    for tr, reals, fakes, z2, ones, zeros in grads:
        tr.gradG.begin()
        tr._freeze_D(True)
        gen = tr.G(z2)
        sim = tr.similar_loss(gen)
        gl = [F.mse_loss(tr.D[i](gen[i]), ones) for i in range(4)]
        (1.2 * gl[0] + 1.2 * gl[1] + 1.2 * gl[2] + gl[3] + 0.1 * world * sim).backward()
        tr.gradG.pack()
        gG.append(tr.gradG.buf.clone())
    expect = (gG[0] + gG[1]) / world
    np.testing.assert_allclose(r[0]["gradG"], expect.numpy(), rtol=2e-3, atol=1e-5 * float(expect.abs().max()))


def _worker_diverged_seeds(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = _build_trainer(distributed=True, seed=100 + 17 * rank)      # every rank initialises DIFFERENT weights
    names = [n for net in [tr.G] + tr.D for n, _ in list(net.named_parameters()) + list(net.named_buffers())]
    flat0 = torch.cat([t.detach().double().reshape(-1) for net in [tr.G] + tr.D
                       for t in list(net.parameters()) + list(net.buffers())])
    reals, z1, z2 = _batch(rank)
    tr.step(reals, z1, z2)
    flat1 = torch.cat([p.detach().reshape(-1) for net in [tr.G] + tr.D for p in net.parameters()])
    adam = torch.cat([v.detach().double().reshape(-1) for opt in [tr.optG] + tr.optD for st in opt.state.values()
                      for v in st.values() if torch.is_tensor(v)])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), init=flat0.numpy(), after=flat1.numpy(), adam=adam.numpy(),
             n=len(names))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_replicas_are_broadcast_from_rank0_gloo():
    """ranks that construct their networks under different seeds must still hold rank 0's parameters and
    buffers (nn.DataParallel replicates module 0, models/PDGNet_v2.py:101-105), and stay identical after a step."""
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_diverged_seeds, args=(world, _free_port(), d), nprocs=world, join=True)
        r = [dict(np.load(os.path.join(d, "rank%d.npz" % i))) for i in range(world)]
    np.testing.assert_array_equal(r[0]["init"], r[1]["init"])
    np.testing.assert_array_equal(r[0]["after"], r[1]["after"])
    np.testing.assert_array_equal(r[0]["adam"], r[1]["adam"])
    # and rank 0's own initialisation is what survived (seed 100), not something else
    tr = _build_trainer(distributed=False, seed=100)
    flat = torch.cat([t.detach().double().reshape(-1) for net in [tr.G] + tr.D
                      for t in list(net.parameters()) + list(net.buffers())])
    np.testing.assert_array_equal(r[0]["init"], flat.numpy())


def test_flat_grads_views_and_zero():
    sys.path.insert(0, ROOT)
    from pdgn_amd.trainer import FlatGrads
    m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    fg = FlatGrads(m.parameters())
    fg.begin()
    m(torch.randn(5, 4)).sum().backward()
    fresh = [p.grad.clone() for p in m.parameters()]
    fg.pack()
    assert fg.buf.abs().sum() > 0
    off = 0
    for p, g in zip(m.parameters(), fresh):
        assert p.grad.data_ptr() == fg.buf[off:].data_ptr()               # .grad is now a view of the buffer
        assert torch.equal(p.grad, g)
        assert p.grad.data_ptr() % 16 == fg.buf.data_ptr() % 16           # every slot starts on a 16-byte boundary of the buffer ...
        off += (p.numel() + 3) // 4 * 4
    assert off == fg.buf.numel() and int((fg.buf != 0).sum()) == sum(int((g != 0).sum()) for g in fresh)      # ... and the padding holds zeros
    fg.all_reduce_mean()                                                   # no process group: no-op


def _shard_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pdgn_amd.evaluation import shard_pairs
    seen = []

    def fill(lo, hi):                                     # stands in for the pair-list kernels: value = f(pair index)
        seen.append((lo, hi))
        p = torch.arange(lo, hi, dtype=torch.float32)
        return p * 2.0 + 1.0, -p

    res = {}
    for total in (7, 8, 1):                               # uneven split, even split, an empty slice on rank 1
        a, b = shard_pairs(total, fill)
        res["a%d" % total], res["b%d" % total] = a.numpy(), b.numpy()
    np.savez(os.path.join(out_dir, "shard%d.npz" % rank), seen=np.array(seen), **res)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_evaluation_pairs_shard_over_two_ranks_gloo():
    """SURVEY.md section 8-e, eval path: each rank fills its slice of the flat pair space, all-gather restores the
    full matrices on every rank."""
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_shard_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        r = [dict(np.load(os.path.join(d, "shard%d.npz" % i))) for i in range(world)]
    for total in (7, 8, 1):
        p = np.arange(total, dtype=np.float32)
        for x in r:
            np.testing.assert_array_equal(x["a%d" % total], p * 2 + 1)
            np.testing.assert_array_equal(x["b%d" % total], -p)
    assert r[0]["seen"].tolist() == [[0, 4], [0, 4], [0, 1]]
    assert r[1]["seen"].tolist() == [[4, 7], [4, 8], [1, 1]]


def test_shard_pairs_without_process_group_is_identity():
    sys.path.insert(0, ROOT)
    from pdgn_amd.evaluation import shard_pairs
    a, = shard_pairs(5, lambda lo, hi: (torch.arange(lo, hi),))
    assert a.tolist() == [0, 1, 2, 3, 4]


def _worker_buckets(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    for mode in ("1", "0"):
        os.environ["PDGN_BUCKETS"] = mode
        tr = _build_trainer(distributed=True)
        reals, z1, z2 = _batch(rank)
        tr.step(reals, z1, z2)
        res["early" + mode] = bool(tr.gradG._early_done)
        res["n_early" + mode] = tr.gradG.n_early
        res["params" + mode] = torch.cat([p.detach().reshape(-1) for p in tr.G.parameters()]).numpy()
        # the reduced gradient per PARAMETER (the flat layouts of the two modes differ)
        res["grads" + mode] = torch.cat([p.grad.detach().reshape(-1) for p in tr.G.parameters()]).numpy()
    os.environ.pop("PDGN_BUCKETS")
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_bucketed_all_reduce_equals_flat_gloo():
    """The deepest block's gradients all-reduced from inside the backward (early bucket, asynchronous) + the rest at the end
    give the same reduced gradients and the same updated parameters as one flat all-reduce."""
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_buckets, args=(world, _free_port(), d), nprocs=world, join=True)
        r = [dict(np.load(os.path.join(d, "rank%d.npz" % i))) for i in range(world)]
    for x in r:
        assert bool(x["early1"]) and int(x["n_early1"]) > 0          # the early bucket was taken from inside the backward
        assert not bool(x["early0"]) and int(x["n_early0"]) == 0
        np.testing.assert_array_equal(x["grads1"], x["grads0"])
        np.testing.assert_array_equal(x["params1"], x["params0"])
    np.testing.assert_array_equal(r[0]["params1"], r[1]["params1"])


def _worker_bench(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import argparse
    import contextlib
    import io
    import bench
    tr = _build_trainer(distributed=True)
    B = 4
    g = torch.Generator().manual_seed(100 + rank)
    res = (32, 64, 128, 256)
    reals = [torch.randn(B, 3, n, generator=g) for n in res]
    args = argparse.Namespace(batch=B, steps=1, warmup=0, graph=False, no_graph=False, no_roofline=True,
                              no_cpu_baseline=True, base_points=16, cpu_sample_batch=2)
    zs = [(torch.randn(B, 128, generator=g) * 0.2, torch.randn(B, 128, generator=g) * 0.2)]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        line = bench.measure_and_report(args, tr, reals, zs, world, rank, torch.device("cpu"), res)
    dist.barrier()                                         # every rank gets here: nobody hangs in a lone collective
    np.savez(os.path.join(out_dir, "bench%d.npz" % rank), printed=buf.getvalue(), has_line=line is not None)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_bench_measurement_with_two_ranks_gloo():
    """bench.py's flop-logging iteration is a full trainer.step() with its gradient all-reduces -- run by
    rank 0 alone it waits for collectives nobody joins.  bench.measure_and_report (everything between building the
    trainer and printing the line) with two gloo ranks: both return, rank 0 prints one line with n_gpus = 2 and the
    executed flops, rank 1 prints nothing."""
    import json
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_bench, args=(world, _free_port(), d), nprocs=world, join=True)
        r = [dict(np.load(os.path.join(d, "bench%d.npz" % i))) for i in range(world)]
    assert bool(r[0]["has_line"]) and not bool(r[1]["has_line"]) and str(r[1]["printed"]) == ""
    lines = [l for l in str(r[0]["printed"]).splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["config"]["global_batch"] == 8
    assert line["config"]["losses_finite"] is True and line["ms_per_step_min_rank"] <= line["ms_per_step_max_rank"]
    assert "error" not in json.dumps(line.get("executed_flops_per_step"))


def _worker_late_gradient(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from pdgn_amd.trainer import FlatGrads
    res = {}
    for mode in ("bucketed", "flat"):
        torch.manual_seed(3)
        w0 = torch.nn.Parameter(torch.randn(5, 6))          # a shallow layer (late gradients)
        w1 = torch.nn.Parameter(torch.randn(6, 4))          # deepest block, reached through an operand made BEFORE the forward
        w2 = torch.nn.Parameter(torch.randn(6, 4))          # deepest block, used directly
        fg = FlatGrads([w0, w1, w2], first=[w1, w2] if mode == "bucketed" else None)
        if mode == "bucketed":
            fg.arm_early()
        fg.begin()
        wa = w1 * 2.0                                        # the "pre-assembled" operand: lowest sequence number of the graph
        x = torch.randn(7, 5, generator=torch.Generator().manual_seed(50 + rank))
        h = torch.tanh(x @ w0)
        seen = {}
        # what the round-4 trigger did: fire when the backward reaches the deepest block's input
        h.register_hook(lambda g: seen.update(at_input=[p.grad is not None for p in (w1, w2)]))
        ((h @ wa).sum() + (h @ w2).pow(2).sum()).backward()
        res[mode + "_early"] = bool(fg._early_done)
        fg.pack()
        fg.all_reduce_mean()
        res[mode] = torch.cat([p.grad.reshape(-1) for p in (w0, w1, w2)]).numpy()
        res[mode + "_at_input"] = np.array(seen["at_input"])
    np.savez(os.path.join(out_dir, "late%d.npz" % rank), **res)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_early_bucket_waits_for_gradients_that_arrive_late_gloo():
    """a weight reached through a node created before the forward (PointGenerator.preassemble) gets its
    gradient AFTER the backward has passed its block's input.  The early bucket is started by the arrival of its last
    gradient (FlatGrads.arm_early), so it equals the flat all-reduce whatever order autograd picks."""
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_late_gradient, args=(world, _free_port(), d), nprocs=world, join=True)
        r = [dict(np.load(os.path.join(d, "late%d.npz" % i))) for i in range(world)]
    for x in r:
        assert bool(x["bucketed_early"]) and not bool(x["flat_early"])
        # the situation the test is about: at the block's input w1's gradient does not exist yet
        assert not bool(x["bucketed_at_input"][0])
        np.testing.assert_array_equal(x["bucketed"], x["flat"])
    np.testing.assert_array_equal(r[0]["bucketed"], r[1]["bucketed"])
    assert not np.array_equal(r[0]["bucketed"], np.zeros_like(r[0]["bucketed"]))


def test_launch_list_ranges_and_host_points_order():
    """PDGNTrainer.step_list issues the list in ranges with the host points' calls between them (the data-parallel collectives)
    and the pacing event where the cut falls: the order of launches, calls and the event record, on a fake list."""
    sys.path.insert(0, ROOT)
    from pdgn_amd import trainer as T

    log = []

    class FakeList:
        info = {"nodes": 100}

        def bind(self, streams, spare):
            pass

        def launch(self, lo=0, hi=None):
            log.append(("launch", lo, 100 if hi is None else hi))

        def position(self, label, fraction):
            return 60

    class FakeEvent:
        def synchronize(self):
            log.append(("sync",))

        def record(self, stream):
            log.append(("record", stream))

    tr = T.PDGNTrainer.__new__(T.PDGNTrainer)
    tr.device = torch.device("cpu")
    tr._static = {"reals": [], "z1": torch.zeros(1), "z2": torch.zeros(1), "out": {"x": 1}}
    tr._list, tr._list_spare, tr._list_done, tr._list_pace = FakeList(), [], FakeEvent(), 0.6
    tr._list_points = [(10, lambda: log.append(("call", "d1")), 1), (59, lambda: log.append(("call", "early")), 0),
                       (80, lambda: log.append(("call", "rest")), 0)]

    class Ctx:
        def __init__(self, s):
            self.s = s

        def __enter__(self):
            log.append(("stream", self.s))

        def __exit__(self, *a):
            pass

    import types
    fake_streams = types.SimpleNamespace(lp="lp", knn="knn", d=["d1", "d2", "d3", "d4"])
    orig = (T._streams.plan, torch.cuda.current_stream, torch.cuda.stream)
    T._streams.plan, torch.cuda.current_stream, torch.cuda.stream = (lambda dev: fake_streams), (lambda dev=None: "main"), Ctx
    try:
        out = tr.step_list()
    finally:
        T._streams.plan, torch.cuda.current_stream, torch.cuda.stream = orig
    assert out == {"x": 1}
    assert log == [("sync",), ("launch", 0, 11), ("stream", "d1"), ("call", "d1"), ("launch", 11, 60), ("record", "main"),
                   ("stream", "main"), ("call", "early"), ("launch", 60, 81), ("stream", "main"), ("call", "rest"),
                   ("launch", 81, 100)], log
