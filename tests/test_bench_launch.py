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


def test_bench_eight_ranks_on_host_memory():
    """World 8 -- the size the driver's scaling run ends with -- on the host-memory backend: eight rank processes, gloo, a collective every
    second frame, every rank on its own share of the host cores; rank 0's line names eight GPUs' worth of streams and the gathered
    table decodes into eight blocks whose cameras are the eight streams' (0.05 g m apart in y)."""
    res = _run(["--gpus", "8", "--steps", "4", "--warmup", "2", "--exchange-batch", "2"], timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == 8 and cfg["streams"] == 8 and cfg["world_size_seen"] == 8 and out["scaling"] == "weak"
    lo, hi = cfg["per_rank_fps_min_max"]
    assert 0 < lo <= hi and out["value"] > 0
    assert cfg["exchange_frames_per_collective"] == 2 and "2 frame(s) per collective" in cfg["exchange"]
    assert cfg["exchange_cost_measured"]["frames_per_collective"] == 2
    tab = cfg["exchange_table"]
    assert tab["blocks"] == 8 and tab["blocks_with_ids"] == 8, tab
    assert all(n > 1000 for n in tab["visible_counts"]), tab
    assert [round(y / 0.05) for y in tab["camera_y_m"]] == list(range(8)), tab          # block g is stream g's
    hc = cfg["host_cores"]
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 8:
        per = ncpu // 8
        assert hc["per_rank_count"] == [per] * 8, hc
        assert len(set(hc["per_rank_first_core"])) == 8, hc                        # disjoint shares


def test_rank_cpu_sets_are_disjoint_and_cover():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench._parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11}
    avail = os.sched_getaffinity(0)
    for world in (2, 4, 8):
        if len(avail) < world:
            continue
        sets = [bench.rank_cpu_set(r, world) for r in range(world)]
        assert all(sets) and all(s <= avail for s in sets)
        assert sum(len(s) for s in sets) == len(set().union(*sets))               # pairwise disjoint
    assert bench.rank_cpu_set(0, 1) is None
