"""bench.py's own launcher (`python bench.py --gpus 2`, no torchrun) end to end on CPU: the parent spawns two
rank processes before anything touches a GPU, the ranks rendezvous over gloo, run the frame loop with the
per-frame visible-list all-gather and rank 0 prints ONE JSON line with n_gpus == 2.

The scene work runs on a host-memory implementation of the C-ABI handed in through --lib (the CPU oracle:
this file is test code, bench.py itself knows nothing about oracle/); the launcher, the rank set-up, the
exchange schedule and the max-over-ranks timing are the code the driver's multi-GPU run executes."""
import json
import os
import subprocess
import sys

import itm_testlib as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=600):
    T.oracle_backend()   # makes sure the library is built
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--lib", T.ORACLE_LIB, "--lib-prefix", "itmo_"] + extra
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_spawns_two_ranks_and_reports_them():
    res = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout          # exactly one JSON line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["streams"] == 2 and out["config"]["world_size_seen"] == 2
    assert out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["collective_backend"] == "gloo" and "8 frame(s) per collective" in out["config"]["exchange"]
    assert out["value"] > 0 and out["config"]["per_rank_fps_min_max"][0] <= out["config"]["per_rank_fps_min_max"][1]
    assert out["roofline"] is None and out["cpu_baseline"] is None      # not the product: nothing is priced
    assert "ALTERNATIVE BACKEND" in out["data"]


def test_bench_rejects_mismatched_world_size():
    # under a launcher that sets WORLD_SIZE, --gpus must agree with it (round 1 silently ran one rank)
    res = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0 and "WORLD_SIZE" in res.stderr


def test_bench_single_rank_line():
    res = _run(["--steps", "2", "--warmup", "1", "--force-exchange"])
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["config"]["streams"] == 1
