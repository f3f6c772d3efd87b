"""bench.py --gpus N launches its N ranks itself when no launcher did: CPU-only plumbing check through
the gloo stub path -- two ranks come up, rank 0's JSON line is relayed, a mismatching rank count is refused."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo-stub", "--steps", "3"] + extra,
                          cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)


@pytest.mark.timeout(300)
def test_gpus_2_self_launches_two_ranks():
    p = _run(["--gpus", "2"])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["stub"] is True
    assert d["allreduce_check"] == 2 * 2048.0                       # both ranks took part in the collective
    assert d["ms_per_step_min_rank"] <= d["ms_per_step_max_rank"]


@pytest.mark.timeout(120)
def test_rank_count_mismatch_is_refused():
    # a launcher-provided world of 1 with --gpus 2 must not print an n_gpus = 1 line
    p = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
