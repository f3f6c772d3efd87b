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
    assert r["kernel"].startswith("gemm_nt_kernel") and "Cijk" not in json.dumps(r)      # the roofline names own kernels only


@pytest.mark.timeout(600)
def test_bench_eval_mode_c5_line():
    """bench.py --eval: config C5 (Chamfer + EMD on 2048-point pairs) at a reduced pair count."""
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--eval", "--eval-pairs", "64", "--steps", "2",
                        "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=550)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["unit"] == "pairs/s" and d["value"] > 0 and d["config"]["finite"] is True
    assert d["roofline"]["kernel"].startswith("approxmatch_kernel") and 0 < d["roofline"]["frac"] < 1
