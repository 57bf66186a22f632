"""bench.py as the driver calls it for N > 1: `python bench.py --gpus N ...` with no WORLD_SIZE in the environment
must start its own ranks (torch.distributed.run children of a parent that touches no GPU), print rank 0's JSON line
as its LAST line and exit with the children's code.  Driven here with --dry-run (gloo, no GPU work)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(extra, gpus=2):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup",
                           "1", "--dry-run"] + extra, env=env, capture_output=True, text=True, timeout=600)


def test_bench_self_launches_its_ranks_and_relays_one_json_line():
    p = _run([])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    rec = json.loads(lines[-1])                                  # the LAST line is the result line
    assert rec["n_gpus"] == 2 and rec["rccl_world"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["rank_ms_per_step"] == [1.0, 2.0]                 # one entry per rank, gathered over the group
    assert set(rec["product_paths"]) == {"graph_path", "streamed_scan"}      # the N > 1 blocks of the two other product paths
    assert len(rec["product_paths"]["graph_path"]["sites_per_rank"]) == 2
    assert sum(1 for ln in lines if ln.lstrip().startswith("{") and '"metric"' in ln) == 1


def test_bench_launcher_passes_a_failing_rank_on():
    p = _run(["--dry-run-fail-rank", "1"])
    assert p.returncode != 0
    assert not any(ln.lstrip().startswith("{") and '"metric"' in ln for ln in p.stdout.splitlines())


def test_eight_ranks_as_the_driver_launches_them():
    """--gpus 8, the driver's largest run: eight children, one JSON line with eight per-rank entries; and a rank that
    dies (rank 5) takes the run's exit code with it instead of leaving seven ranks in a collective."""
    p = _run([], gpus=8)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert rec["n_gpus"] == 8 and rec["rccl_world"] == 8 and rec["rank_ms_per_step"] == [float(r + 1) for r in range(8)]
    # the sharded product paths' plan: every rank an eighth of the regions and -- shard_index -- about an eighth of the graph's
    # site records, not a replica; the ranks of a node SHARE its cores: a rank's parse threads are its share of them
    plan = rec["product_paths"]
    gp, ss = plan["graph_path"], plan["streamed_scan"]
    assert len(gp["regions_per_rank"]) == 8 and len(set(gp["regions_per_rank"])) == 1
    assert all(0 < x < gp["sites_total"] / 4 for x in gp["sites_per_rank"]) and sum(gp["sites_per_rank"]) <= 1.2 * gp["sites_total"]
    cores = ss["host_cores"]
    assert ss["parse_threads_per_rank"] == [max(1, cores // 8)] * 8 and sum(ss["parse_threads_per_rank"]) <= max(cores, 8)
    assert sum(ss["files_per_rank"]) == 8 * ss["files_per_rank"][0]
    p = _run(["--dry-run-fail-rank", "5"], gpus=8)
    assert p.returncode != 0
    assert not any(ln.lstrip().startswith("{") and '"metric"' in ln for ln in p.stdout.splitlines())
