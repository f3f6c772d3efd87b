#!/usr/bin/env python3
"""bench.py -- generator+D step throughput of the PDGN hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 35] [--no-cpu-baseline] [--eval]

A "step" is one complete training iteration of models/PDGNet_v2.py:171-256 (generator forward
x2, four discriminator updates, six local-pair shape losses, generator backward, five Adam
steps) on a synthetic ShapeNet-shaped batch that is already resident in HBM.  Metric (BASELINE.json):
final-resolution points per second = n_gpus * B * 2048 / t_step, B = 35 per GPU (weak scaling).

N > 1: one rank per GPU over RCCL.  Launched through torch.distributed.run (the driver's way) the ranks
are already there; launched as plain `python bench.py --gpus N`, this process starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD -- before anything here has
touched a GPU -- relays rank 0's JSON line and exits with the child's code.  Rank 0 prints ONE JSON line.

--eval times configuration C5 instead (evaluation path: Chamfer + approximate EMD on 512 pairs of
2048-point clouds, evaluation/evaluation_metrics.py:26-45, 85-121) and prints its own JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_32x32x2_f32 / 16x16x4_f32)
_GEMM_ENV = os.environ.get("PDGN_GEMM", "")                      # (the library reads the same variable at first use)
GEMM_ARITHMETIC = ("fp32 matrix instructions (PDGN_GEMM=fp32)" if _GEMM_ENV.startswith("f") else
                   "fp32 operands, results and accumulation; the large contractions: each product = three fp16 MFMA partial products of "
                   "the operands' two-way fp16 splits after an exact power-of-two scaling of every operand ROW by that row's largest "
                   "magnitude (csrc/gemm_x3.hip, NP = 2; per row since round 6, per operand before), the others six bf16 products of "
                   "three-way bf16 splits: against fp64 below the fp32 matrix instructions' error, also row by row (gemm_accuracy, "
                   "gemm_accuracy.row_scaled; tests/test_gpu_deconv.py); ms_per_step_x3 = the same step with three bf16 parts "
                   "everywhere (PDGN_GEMM=x3); PDGN_GEMM=fp32 selects the fp32 instructions" if not _GEMM_ENV.startswith("x3") else
                   "fp32 operands, results and accumulation; each product = six bf16 MFMA partial products of the operands' "
                   "three-way bf16 splits (csrc/gemm_x3.hip): error per product <= 2^-23, against fp64 below the fp32 matrix "
                   "instructions' (tests/test_gpu_deconv.py); PDGN_GEMM=fp32 selects those instructions")
# SURVEY.md section 8-d / BASELINE.md section 2: the reference's direct form costs ~222 GFLOP per sample and G+D step
DIRECT_FORM_FLOPS_PER_SAMPLE = 222e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=35, help="per-GPU batch (BASELINE.json: 35)")
    ap.add_argument("--base-points", type=int, default=128, help="128: 256->2048 (reference); 256: 512->4096")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=35,
                    help="batch of the CPU-baseline sample: 35 = the batch the metric is quoted on (one iteration, ~18 s on "
                         "16 threads); below 24 three iterations are timed and the median reported")
    ap.add_argument("--graph", action="store_true",
                    help="replay the iteration as hipGraphs (trainer.capture).  Off by default: the eager step is "
                         "faster on MI355X (DESIGN.md section 6), and on ROCm 7.2 graph launches are not reliably "
                         "ordered against RCCL / eager kernels on the same stream")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--issue", default="list", choices=("list", "eager"),
                    help="list (default, one process): the iteration is captured once and every step re-issues its launches "
                         "from C on the eager schedule's streams (trainer.capture_list, csrc/replay.hip); eager: every "
                         "launch issued from Python")
    ap.add_argument("--no-roofline", action="store_true", help="skip the roofline kernels (for clean rocprof traces)")
    ap.add_argument("--no-x3-leg", action="store_true", help="skip the extra 10-step loop on three bf16 parts everywhere (ms_per_step_x3)")
    ap.add_argument("--eval", action="store_true", help="time config C5 (Chamfer + EMD, 512 pairs of 2048 points)")
    ap.add_argument("--eval-pairs", type=int, default=512)
    ap.add_argument("--no-eval-c5", action="store_true", help="leave the C5 sub-object (eval_c5) out of the default line")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo-stub"),
                    help="gloo-stub: CPU-only plumbing check of the N-rank launch path (tests), no HIP work")
    ap.add_argument("--master-port", type=int, default=0)
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child job and relay its result.  Nothing in
    this process has initialised a GPU (torch is not even imported yet), and the child is a fresh process."""
    import socket
    port = args.master_port
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print("bench.py: the %d-rank child job failed (exit code %d)" % (args.gpus, proc.returncode), file=sys.stderr)
        sys.exit(proc.returncode or 1)
    got = json.loads(line).get("n_gpus")
    if got != args.gpus:
        print("bench.py: asked for %d GPUs, the job reports n_gpus = %r" % (args.gpus, got), file=sys.stderr)
        sys.exit(1)
    print(line, flush=True)
    sys.exit(0)


def host_cpu():
    """CPU model, sockets and physical cores of this host (from /proc/cpuinfo)."""
    model, phys, cores = "unknown", set(), set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name") and model == "unknown":
                    model = ln.split(":", 1)[1].strip()
                elif ln.startswith("physical id"):
                    pid = ln.split(":", 1)[1].strip()
                    phys.add(pid)
                elif ln.startswith("core id"):
                    cid = ln.split(":", 1)[1].strip()
                    cores.add((pid, cid))
    except OSError:
        pass
    return {"model": model, "sockets": len(phys) or None, "physical_cores": len(cores) or None,
            "logical_cpus": os.cpu_count()}


def cpu_baseline(sample_batch, samples=None):
    """The oracle's torch-CPU/C restatement of the same iteration (oracle/pdgnet_ref.TrainerRef), timed on this box's
    host cores on a bounded sample: `samples` full G+D iterations at a reduced batch, median reported per point."""
    import torch
    from oracle import cref, pdgnet_ref
    from pdgn_amd.trainer import synthetic_batch
    samples = samples if samples is not None else (1 if sample_batch >= 24 else 3)   # a bounded sample: ~20 s of CPU work
    cref.build()
    # 16 threads is the fastest setting measured on the 2x EPYC 9575F GPU-box host (16 thr: 17.7 s, 32: 20.9 s,
    # 64: 26.3 s, 256: minutes per B=35 iteration -- torch's intra-op pool oversubscribes).
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    torch.manual_seed(9999)
    tr = pdgnet_ref.TrainerRef()
    g = torch.Generator().manual_seed(1)
    z = lambda b: torch.randn(b, 128, generator=g) * 0.2
    tr.step(synthetic_batch(2, "cpu"), z(2), z(2))                 # warm-up (thread pools, allocator)
    reals = synthetic_batch(sample_batch, "cpu")
    times = []
    for _ in range(samples):
        t0 = time.perf_counter()
        tr.step(reals, z(sample_batch), z(sample_batch))
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": sample_batch * 2048 / med, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
            "samples_s": [round(t, 3) for t in times], "host": host_cpu(),
            "sample": "%d full G+D iterations (oracle/pdgnet_ref.TrainerRef: torch-CPU fp32 + C pointops) at B=%d, "
                      "256->2048 pts, after a B=2 warm-up; median %.1f s on %d threads (the fastest thread count "
                      "measured on this host class)" % (samples, sample_batch, med, cores)}


def stub_main(args):
    """CPU-only plumbing of the N-rank path (tests/test_bench_launch.py): gloo, a toy step with one all-reduce."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        sys.exit(2)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.ones(1024)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = (x * 2).sum()
        if world > 1:
            dist.all_reduce(y)
    dt = time.perf_counter() - t0
    ts = [torch.tensor([dt], dtype=torch.float64) for _ in range(world)]
    if world > 1:
        dist.all_gather(ts, torch.tensor([dt], dtype=torch.float64))
    if rank == 0:
        ms = [t.item() / args.steps * 1e3 for t in ts]
        print(json.dumps({"metric": "plumbing stub (no HIP work)", "value": world * 1024 / max(ms) * 1e3, "unit": "elements/s",
                          "n_gpus": world, "rccl_ranks": dist.get_world_size() if world > 1 else 1, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": max(ms), "ms_per_step_min_rank": min(ms),
                          "ms_per_step_max_rank": max(ms), "stub": True, "allreduce_check": float(y)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def init_dist(args):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("PDGN_FORCE_DIST") == "1"        # exercise the RCCL path on a single GPU
    if args.gpus != world:
        # never print a line whose n_gpus differs from what was asked for
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    return world, rank, local


def barrier():
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available():                 # (the CPU / gloo tests of this file's rank logic have no GPU)
        torch.cuda.synchronize()


def executed_flops(trainer, reals, z1, z2):
    """Dense-contraction flops ONE iteration really launches: every pdgn_gemm_nt / _nn / _tn problem (2 m n k) plus the
    feature-space kNN Gram products (2 N^2 F per sample), logged by the wrappers during one extra untimed step."""
    import torch
    from pdgn_amd import deconv, fused
    fused.GEMM_LOG, deconv.KNN_LOG = [], []
    trainer.step(reals, z1, z2)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    gemm, knn = fused.GEMM_LOG, deconv.KNN_LOG
    fused.GEMM_LOG = deconv.KNN_LOG = None
    f_gemm = sum(2.0 * m * n * k for (_, m, n, k) in gemm)
    f_knn = sum(2.0 * b * n * n * f for (b, f, n) in knn)
    return {"gemm": f_gemm, "feature_knn_gram": f_knn, "gemm_launches": len(gemm), "knn_launches": len(knn)}


def _event_ms(fn, iters=5):
    import torch
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


# Instruction mix of emd_cost_kernel<true>'s inner loops (csrc/structural.hip), read off the gfx950 ISA hipcc emits
# (tools/emd_isa_mix.sh): vector instructions per 16 (own point, staged point) elements of a lane -- one unrolled trip of
# 4 staged points x 4 own points -- as (packed-fp32 + plain, transcendental).  Issue cost per wave instruction from
# MI355X_MICROARCH.md, cycle constants: v_add / v_fma / v_pk_* 4 cycles, v_exp_f32 / v_sqrt_f32 8.
EMD_LOOPS = {"phase1_or_2": (63, 16), "phase3_plus_next_phase1": (112, 32), "phase3_last": (87, 32)}


def emd_executed_elements(a, b):
    """Point-pair elements emd_cost_kernel<SORTED> really evaluates, per sweep kind, summed over the pairs (a (P, n, 3),
    b (P, m, 3) on the GPU, n, m <= 2048): both clouds are sorted by x and a wave (256 own points, x in [xlo, xhi]) walks of
    every staged tile (1024 points) only the run with x in [xlo - r, xhi + r], r = sqrt(150) / s -- everything outside would
    have added exact zeros (csrc/structural.hip).  Restated here with torch.searchsorted so that the roofline's work is the
    work executed, not the dense 19 n m."""
    import torch
    P, n, m = a.shape[0], a.shape[1], b.shape[1]
    xa, xb = a[:, :, 0].sort(dim=1)[0].contiguous(), b[:, :, 0].sort(dim=1)[0].contiguous()
    SQ = 1.2011224087864498                                    # sqrt(log2 e)

    def sweep(own, staged, s):
        r = 12.2475 / s
        tot = 0.0
        for w0 in range(0, own.shape[1], 256):
            blk = own[:, w0:w0 + 256]
            lo_x, hi_x = blk[:, :1] - r, blk[:, -1:] + r
            for t0 in range(0, staged.shape[1], 1024):
                tile = staged[:, t0:t0 + 1024].contiguous()
                lo = torch.searchsorted(tile, lo_x.contiguous(), right=False)
                hi = torch.searchsorted(tile, hi_x.contiguous(), right=True)
                lpad = (tile.shape[1] + 3) // 4 * 4
                run = (torch.clamp((hi + 3) // 4 * 4, max=lpad) - lo // 4 * 4).clamp(min=0)
                tot += float(run.sum().item()) * 256.0
        return tot
    out = {"phase1_or_2": sweep(xa, xb, SQ * 2.0 ** 7), "phase3_plus_next_phase1": 0.0, "phase3_last": 0.0}
    for j in range(7, -2, -1):
        out["phase1_or_2"] += sweep(xb, xa, SQ * 2.0 ** j)
        if j > -1:
            out["phase3_plus_next_phase1"] += sweep(xa, xb, SQ * 2.0 ** (j - 1))
        else:
            out["phase3_last"] += sweep(xa, xb, SQ * 2.0 ** j)
    return out


def emd_issue_model(elements):
    """(issue cycles over all pairs on one SIMD, vector instructions per executed element, share of the cycles spent on
    transcendentals) for `elements` = {sweep kind: executed point-pair elements}."""
    cyc = ins = trans = tot = 0.0
    for name, el in elements.items():
        plain, tr = EMD_LOOPS[name]
        cyc += el * (plain * 4 + tr * 8) / 16.0
        ins += el * (plain + tr) / 16.0
        trans += el * tr * 8 / 16.0
        tot += el
    return cyc / 64.0, ins / tot, trans / cyc


def eval_c5(pairs=512, steps=5, warmup=1):
    """Config C5: Chamfer (nn_distance) + approximate EMD (fused pdgn_emd_cost) on `pairs` pairs of 2048 x 2048 points,
    evaluation/evaluation_metrics.py:26-45, 85-121.  Returns the fields of its bench line."""
    import torch
    from pdgn_amd.structural_losses import emd_cost, nn_distance
    P, N = pairs, 2048
    g = torch.Generator().manual_seed(9999)
    a = (torch.rand(P, N, 3, generator=g) * 2 - 1).cuda()
    b = (torch.rand(P, N, 3, generator=g) * 2 - 1).cuda()

    def step():
        d1, d2 = nn_distance(a, b)
        return d1.mean(1) + d2.mean(1), emd_cost(a, b)

    for _ in range(max(warmup, 1)):
        cd, emd = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        cd, emd = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms_emd, ms_cd = _event_ms(lambda: emd_cost(a, b)), _event_ms(lambda: nn_distance(a, b))
    # emd_cost_kernel is bound by vector-instruction ISSUE, not by HBM (inputs: 25 MB) and not by the transcendental
    # unit alone: `peak` = the pair rate at which the instruction stream it EXECUTES on this input (the sorted sweeps skip
    # runs of exact zeros: emd_executed_elements) would issue back to back on all 1024 SIMDs at 2.4 GHz; `exp_frac` = the
    # share of those issue cycles taken by v_exp_f32 / v_sqrt_f32
    elements = emd_executed_elements(a, b)
    cyc, instr_per_element, exp_frac = emd_issue_model(elements)
    peak_pairs = 4 * 256 * 2.4e9 / (cyc / P)
    ach = P / (ms_emd * 1e-3)
    dense = 19.0 * N * N * P
    # the ALGORITHM's own floor, next to the issue-mix one: every executed element takes one v_exp_f32 or
    # v_sqrt_f32 (quarter rate: 8 lanes per clock and SIMD); the rest of its 6.9 instructions is this kernel's way of feeding it
    trans_floor_us = sum(elements.values()) / (4 * 256 * 8 * 2.4e9) * 1e6
    return {"pairs": P, "pairs_per_s": P / dt, "ms_per_step": dt * 1e3, "ms_per_512": dt * 1e3 * 512.0 / P,
            "finite": bool(torch.isfinite(cd).all() and torch.isfinite(emd).all()),
            "roofline": {"kernel": "emd_cost_kernel (fused approximate-EMD cost)", "bound": "valu-issue",
                         "achieved": ach, "peak": peak_pairs, "unit": "pairs/s", "frac": ach / peak_pairs,
                         "instr_per_element": instr_per_element, "exp_frac": exp_frac,
                         "transcendental_floor_us": trans_floor_us, "frac_of_transcendental_rate": trans_floor_us / (ms_emd * 1e3),
                         "executed_elements_per_pair": sum(elements.values()) / P, "dense_elements_per_pair": 19.0 * N * N,
                         "executed_fraction_of_dense": sum(elements.values()) / dense,
                         "traffic": None, "us_per_launch": ms_emd * 1e3,
                         "algorithmic_bytes_per_launch": P * 2.0 * N * 12,
                         "others": [{"kernel": "nndist_kernel (both Chamfer directions)", "bound": "valu",
                                     "achieved": 2.0 * P * N * N / (ms_cd * 1e-3) / 1e9, "unit": "G pair evaluations/s",
                                     "us_per_launch": ms_cd * 1e3, "algorithmic_bytes_per_launch": P * (2.0 * N) * 20}]}}


def eval_main(args):
    """bench.py --eval: config C5 as a line of its own."""
    import torch
    torch.cuda.set_device(0)
    r = eval_c5(args.eval_pairs, args.steps, args.warmup)
    P = r["pairs"]
    line = {"metric": "Chamfer + approx-EMD structural losses, pairs/sec (2048-pt pairs, B=%d)" % P,
            "value": r["pairs_per_s"], "unit": "pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "eval path C5: nn_distance (Chamfer) + fused approximate EMD cost on %d pairs of 2048 x 2048 "
                                   "uniform points in [-1, 1]^3 (evaluation_metrics.py:26-45, 85-121)" % P,
                       "finite": r["finite"]},
            "roofline": r["roofline"]}
    print(json.dumps(line), flush=True)


def main():
    args = parse()
    # (PDGN_FORCE_DIST=1 with one GPU: the SAME child-launch path -- this process starts torch.distributed.run before anything has
    # touched a GPU, the child builds a one-rank RCCL group and issues the launch list with the collectives as host points -- so that
    # the N-rank code runs on the one-GPU boxes too: tests/test_gpu_bench_contract.py)
    forced = os.environ.get("PDGN_FORCE_DIST") == "1" and args.backend != "gloo-stub" and not args.eval
    if (args.gpus > 1 or forced) and "WORLD_SIZE" not in os.environ:
        self_launch(args)                                        # does not return
    if args.backend == "gloo-stub":
        return stub_main(args)
    if args.eval:
        return eval_main(args)
    import torch
    import torch.distributed as dist
    world, rank, local = init_dist(args)
    device = torch.device("cuda", local)
    from pdgn_amd.trainer import PDGNTrainer, noise, synthetic_batch

    torch.manual_seed(9999)                                      # the same initial weights on every rank (they are
    res = tuple((2 * args.base_points) << i for i in range(4))   # broadcast from rank 0 as well); data differs per rank
    trainer = PDGNTrainer(device=device, base_points=args.base_points,
                          distributed=True if os.environ.get("PDGN_FORCE_DIST") == "1" else None)
    trainer.train()
    B = args.batch
    reals = synthetic_batch(B, device, seed=9999 + rank, n_points=res[3], resolutions=res)
    gen = torch.Generator().manual_seed(1234 + rank)
    zs = [(noise(B, device, gen), noise(B, device, gen)) for _ in range(args.warmup + args.steps)]

    measure_and_report(args, trainer, reals, zs, world, rank, device, res)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def x3_leg(args, device, reals, zs, steps=10, warmup=3):
    """The same iteration with every contraction on three bf16 parts (six bf16 MFMA products per fp32 product: the arithmetic that
    needs no operand scale at all, PDGN_GEMM=x3), from a second trainer built in that mode, issued as a launch list like the
    headline: ms per step (VERDICT r5 next #1d: the reference-equivalent-precision number beside the headline)."""
    import torch
    from pdgn_amd import _lib
    from pdgn_amd.trainer import PDGNTrainer
    old = _lib.set_gemm_mode("x3")
    try:
        torch.manual_seed(9999)
        tr = PDGNTrainer(device=device, base_points=args.base_points)
        tr.train()
        for _ in range(2):
            tr.step(reals, *zs[0])
        step = lambda z: tr.step(reals, *z)
        if getattr(args, "issue", "eager") == "list" and getattr(tr, "overlap", False):
            tr.capture_list(reals, *zs[0])
            step = lambda z: tr.step_list(None, *z)
        for i in range(warmup):
            step(zs[i % len(zs)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(zs[(warmup + i) % len(zs)])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    finally:
        _lib.set_gemm_mode(old)


def measure_and_report(args, trainer, reals, zs, world, rank, device, res):
    """Warm-up, the timed region (barrier + synchronise on both sides, max over ranks), the flop-logging iteration and
    rank 0's JSON line.  Everything in here that issues a collective runs on EVERY rank (tests/test_distributed_cpu.py
    drives this function with two gloo ranks)."""
    import torch
    import torch.distributed as dist
    B = args.batch
    step, graphed = trainer.step, False
    if args.graph and not args.no_graph and not trainer.distributed:
        try:                                   # one hipGraph per iteration (trainer.capture)
            trainer.capture(reals, *zs[0])
            step, graphed = (lambda reals, z1, z2: trainer.step_graphed(None, z1, z2)), True
        except Exception as e:                 # a capture problem must not lose the measurement
            print("hipGraph capture failed (%r): running eagerly" % (e,), file=sys.stderr)
            torch.cuda.synchronize()
    # Two untimed priming iterations grow the caching allocator's pools and measure the stream-to-queue map; they must
    # never land in the timed region, whatever W the caller asks for.
    for _ in range(2):
        step(reals, *zs[0])
    issue = "hipgraph" if graphed else "eager"
    if not graphed and getattr(args, "issue", "eager") == "list" and getattr(trainer, "overlap", False):
        # the SAME launches as the eager step, re-issued from a recorded list; under data parallelism the gradient all-reduces are
        # host points of the list (trainer.capture_list), issued between its ranges
        ok = 1
        try:
            trainer.capture_list(reals, *zs[0])
        except Exception as e:                 # a capture problem must not lose the measurement
            print("launch-list capture failed (%r): issuing eagerly" % (e,), file=sys.stderr)
            torch.cuda.synchronize()
            ok = 0
        if world > 1:                          # every rank issues the same sequence of collectives: all on the list, or none
            flag = torch.tensor([ok], device=device, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            step, issue = (lambda reals, z1, z2: trainer.step_list(None, z1, z2)), "list"
            step(reals, *zs[0])
    for i in range(args.warmup):
        step(reals, *zs[i])
    # the dominant kernel (conv2's dense half, stage 4, forward: one launch per generator pass) timed INSIDE the timed steps: HIP
    # events around each of its launches, on the stream the launch list issues it on (csrc/replay.hip)
    in_step = None
    if issue == "list" and not getattr(args, "no_roofline", False):
        try:
            from pdgn_amd import roofline as _rf
            spans = _rf.conv2_in_step_spans(trainer._list, args.batch, args.base_points)
            if spans:
                trainer._list.time_spans(spans, args.steps)
                in_step = True
        except Exception as e:
            print("in-step kernel timing unavailable (%r)" % (e,), file=sys.stderr)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(reals, *zs[args.warmup + i])
    barrier()
    dt_local = time.perf_counter() - t0
    if in_step:
        try:
            in_step = trainer._list.timed_ms()
            trainer._list.time_spans([], 0)
        except Exception as e:
            print("in-step kernel timing failed (%r)" % (e,), file=sys.stderr)
            in_step = None
    dt, dts = dt_local, [dt_local]
    if world > 1:
        t = torch.tensor([dt_local], device=device, dtype=torch.float64)
        all_t = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(all_t, t)
        dts = [x.item() for x in all_t]
        dt = max(dts)
    finite = all(torch.isfinite(v).item() for v in out.values())

    # The flop-logging iteration is a full trainer.step(): under data parallelism it issues the gradient all-reduces, so
    # EVERY rank runs it (rank 0 alone would wait for collectives nobody else joins); only rank 0's log is reported.
    flops = None
    if not graphed:
        try:
            flops = executed_flops(trainer, reals, *zs[0])
        except Exception as e:
            flops = {"error": repr(e)}
        barrier()

    if rank == 0:
        ms = dt / args.steps * 1e3
        line = {
            # BASELINE.json's metric at its own configuration; any other --batch / --base-points is named as what it is
            "metric": "generator+D step points/sec (B=%d, %d->%d pts)" % (B, res[0], res[3]),
            "value": world * B * res[3] / (dt / args.steps),
            "unit": "points/s",
            "n_gpus": world, "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "ms_per_step_min_rank": min(dts) / args.steps * 1e3, "ms_per_step_max_rank": ms,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PDGNet_v2 G+D iteration (models/PDGNet_v2.py:171-256), chair-shaped "
                                   "synthetic batch, per-GPU batch %d, %d->%d pts, random-init weights"
                                   % (B, res[0], res[3]),
                       "global_batch": world * B, "points_all_resolutions_per_s": world * B * sum(res) / (dt / args.steps),
                       "parallelism": "dp%d" % world, "losses_finite": finite, "hipgraph": graphed,
                       # list: every launch of the iteration re-issued from C (csrc/replay.hip), one iteration in flight
                       "issue": issue,
                       # the collective backend the gradient all-reduces ran on (None: one process, no process group)
                       "process_group": dist.get_backend() if dist.is_initialized() else None,
                       # how one fp32 product of the dense contractions is formed
                       "gemm_arithmetic": GEMM_ARITHMETIC},
        }
        if flops is not None and "error" in flops:
            line["executed_flops_per_step"] = flops
        elif flops is not None:
            total = flops["gemm"] + flops["feature_knn_gram"]                 # per rank = per GPU
            line["executed_flops_per_step"] = total
            line["executed_flops_detail"] = flops
            # the iteration's executed contraction flops per second against the roof of the instruction the contractions
            # run on -- the same roof `roofline` prices the dominant kernel against (bf16 matrix peak / 6 products per fp32
            # product for the x3 kernels, the fp32 matrix peak for PDGN_GEMM=fp32) -- and, named, against the fp32 peak
            from pdgn_amd import roofline as _rf
            nprod = _rf.products()
            peak = (_rf.MFMA_BF16_PEAK_TFLOPS / nprod if nprod else MFMA_F32_PEAK_TFLOPS) * 1e12
            line["step_mfma_frac"] = total / (ms * 1e-3) / peak
            line["step_mfma_frac_peak_tflops"] = peak / 1e12
            line["step_frac_of_fp32_instruction_peak"] = total / (ms * 1e-3) / (MFMA_F32_PEAK_TFLOPS * 1e12)
            line["direct_form_flops_per_step"] = DIRECT_FORM_FLOPS_PER_SAMPLE * B
            line["algebraic_saving"] = DIRECT_FORM_FLOPS_PER_SAMPLE * B / total if total else None
        if world == 1 and not graphed and not getattr(args, "no_x3_leg", False) and not args.no_roofline:      # (traces run with --no-roofline: one trainer's kernels only)
            try:
                from pdgn_amd import _lib as _l
                if _l.gemm_mode() == "x2":
                    line["ms_per_step_x3"] = x3_leg(args, device, reals, zs)
            except Exception as e:                               # never lose the headline number
                line["ms_per_step_x3"] = {"error": repr(e)}
        if not args.no_roofline:
            try:
                from pdgn_amd import roofline
                line["roofline"] = roofline.measure(args.batch, args.base_points, device)
                if in_step:
                    line["roofline"] = roofline.attach_in_step(line["roofline"], in_step)
                line["gemm_accuracy"] = roofline.gemm_accuracy(device)        # both kernels against fp64, same operands
            except Exception as e:                               # never lose the headline number
                line["roofline"] = {"error": repr(e)}
        if world == 1 and not args.no_eval_c5:               # BASELINE.json configs[4], ~0.3 s: Chamfer + EMD, 512 pairs
            try:
                line["eval_c5"] = eval_c5()
            except Exception as e:
                line["eval_c5"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:          # host baseline: rank 0, N = 1 only
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample_batch)
        print(json.dumps(line), flush=True)
        return line
    return None


if __name__ == "__main__":
    main()
