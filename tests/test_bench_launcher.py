"""CPU tier: `python bench.py --gpus N` (RANK unset) launches its own ranks, relays rank 0's single JSON
line and propagates failure.  --dry-run swaps the GPU work for a gloo rendezvous + all-reduce."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_self_launch_relays_one_json_line():
    p = _run("--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 5 and res["dry_run"] is True
    assert res["instances_seen"] == 2 * 1024  # both ranks took part in the all-reduce


def test_self_launch_propagates_rank_failure():
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("needs a box without a GPU: the ranks must fail")
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras", "--cpu-sample", "0")
    assert p.returncode != 0
    assert "metric" not in p.stdout
