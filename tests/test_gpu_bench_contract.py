"""bench.py's one-line JSON contract (driver-facing): fields, units, the roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    # a small CPU sample keeps this test short; the default run uses the full B=35 iteration
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                        "--cpu-sample-batch", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "points/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["losses_finite"] is True
    assert abs(d["value"] - 35 * 2048 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == "points/s" and c["value"] > 0 and c["cores"] >= 1
    assert "host" in c and "samples_s" in c and len(c["samples_s"]) >= 3
    assert d["rccl_ranks"] == 1 and d["ms_per_step_min_rank"] <= d["ms_per_step_max_rank"]
    assert d["executed_flops_per_step"] > 1e12 and 0.0 < d["step_mfma_frac"] < 1.0 and d["algebraic_saving"] > 1.0
    assert r["kernel"].startswith("gemm_x3_kernel") and "Cijk" not in json.dumps(r)      # the roofline names own kernels only
    # the dominant contraction runs on two scaled fp16 parts: three fp16 MFMA products per fp32 product, priced against THAT roof
    np_ = r["mfma"]["products_per_fp32_product"]
    assert np_ == 3 and r["mfma"]["instruction"] == "v_mfma_f32_32x32x16_f16" and abs(r["peak"] * np_ - r["mfma"]["instruction_peak_tflops"]) < 1e-6
    assert "back_to_back" in r and r["back_to_back"]["us_per_launch"] > 0 and "timing" in r      # timed inside the steps; alone beside it
    acc = d["gemm_accuracy"]                                     # the arithmetic of the contractions, measured in the same run
    for mode in ("x2", "x3"):
        assert 0.0 < acc["max_error_vs_fp64_" + mode] < 1e-6 and acc["max_error_vs_fp64_" + mode] <= 1.25 * acc["max_error_vs_fp64_fp32"]
    assert acc["max_error_vs_fp64_x2"] != acc["max_error_vs_fp64_x3"]                           # (the shape is one the two-part form takes)
    c5 = d["eval_c5"]                                            # config C5 rides in the default line
    assert c5["pairs"] == 512 and c5["finite"] is True and c5["pairs_per_s"] > 0 and c5["ms_per_512"] > 0
    assert c5["roofline"]["bound"] == "valu-issue" and 0 < c5["roofline"]["frac"] < 1 and 0 < c5["roofline"]["exp_frac"] < 1


@pytest.mark.timeout(600)
def test_bench_eval_mode_c5_line():
    """bench.py --eval: config C5 (Chamfer + EMD on 512 pairs of 2048-point clouds) at its stated batch."""
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--eval", "--eval-pairs", "512", "--steps", "2",
                        "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=550)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["unit"] == "pairs/s" and d["value"] > 0 and d["config"]["finite"] is True
    assert d["roofline"]["kernel"].startswith("emd_cost_kernel") and 0 < d["roofline"]["frac"] < 1


@pytest.mark.timeout(900)
def test_bench_two_gpus_over_rccl():
    """the first box with two GPUs exercises the N > 1 path on hardware -- bench.py launches its own two
    ranks, RCCL all-reduces (early bucket underneath the backward included), the post-communicator stream -> queue
    probe, the flop-logging iteration on every rank.  Skipped on the single-GPU boxes of this pool."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-roofline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["config"]["global_batch"] == 70
    assert d["config"]["losses_finite"] is True and d["executed_flops_per_step"] > 1e12
    assert abs(d["value"] - 2 * 35 * 2048 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


@pytest.mark.timeout(900)
def test_bench_n_rank_launch_path_on_one_gpu():
    """the path `bench.py --gpus N` takes for N > 1 -- self-launch of torch.distributed.run from a process that
    has not touched the GPU, RANK / WORLD_SIZE from the environment, an RCCL process group, replicas broadcast from rank 0, the
    launch list captured with the six gradient all-reduces as host points and issued in ranges around them -- executed on the ONE
    GPU this pool's boxes have (PDGN_FORCE_DIST=1: a one-rank RCCL group), next to the single-process run on the same box: same
    contract fields, `issue == "list"`, `rccl_ranks == 1`, and a step within 1.0 ms of the single-process step (measured +0.4 ..
    +0.6 ms: six pack copies and six one-rank all-reduces).  No curve is simulated: n_gpus stays 1."""
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PDGN_FORCE_DIST")}
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3", "--no-cpu-baseline",
            "--no-roofline", "--no-eval-c5"]
    res = {}
    for tag, env in (("single", base), ("rccl", dict(base, PDGN_FORCE_DIST="1"))):
        p = subprocess.run(args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
        assert p.returncode == 0, (tag, p.stderr[-3000:])
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, (tag, p.stdout[-2000:])
        res[tag] = json.loads(lines[0])
    s, r = res["single"], res["rccl"]
    for d in (s, r):
        assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["config"]["issue"] == "list" and d["config"]["losses_finite"] is True
        assert d["config"]["global_batch"] == 35 and abs(d["value"] - 35 * 2048 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert s["config"]["process_group"] is None and r["config"]["process_group"] == "nccl"       # (= RCCL on ROCm)
    assert r["config"]["parallelism"] == "dp1" and r["executed_flops_per_step"] == s["executed_flops_per_step"]
    assert r["ms_per_step"] <= s["ms_per_step"] + 1.0, (r["ms_per_step"], s["ms_per_step"])
