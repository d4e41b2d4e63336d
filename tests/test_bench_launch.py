"""`bench.py --gpus N` without a launcher starts N ranks itself: the parent spawns fresh children
before touching a GPU, every child sees RANK / WORLD_SIZE / MASTER_*, rank 0 prints the one JSON line with n_gpus = N.
Runs on CPU through the --dry-run stand-in (gloo group, same launch code, no product call)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "1", "--warmup", "0"] + extra,
                       env=e, capture_output=True, text=True, timeout=180)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks():
    out = _run(["--gpus", "2"])
    assert out["n_gpus"] == 2 and out["dry_run"] is True
    assert out["pid"] != out["ppid"]          # the rank is a child of the launcher, not the launcher itself
    # the line carries what every rank did and what the process group connected (the GPU modes print the same fields)
    assert out["collective"]["world_size_seen_by_process_group"] == 2 and [r["rank"] for r in out["ranks"]] == [0, 1]
    assert len({r["pid"] for r in out["ranks"]}) == 2


def test_gpus_1_runs_in_process():
    assert _run([])["n_gpus"] == 1


def test_under_a_launcher_the_environment_wins():
    """Under torch.distributed.run the ranks already exist: WORLD_SIZE is set and bench.py must not spawn again."""
    out = _run(["--gpus", "1"], env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert out["n_gpus"] == 1


def test_parent_never_imports_torch_before_spawning():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[: src.index("def launch_ranks")]
    assert "import torch" not in head and "threecrate_amd" not in head.replace("sys.path", "")
    body = src[src.index("def main():"):]
    spawn = body.index("launch_ranks(args.gpus)")
    assert "import torch" not in body[:spawn]


def test_a_dead_rank_ends_the_job():
    """launch_ranks waited on every child in turn -- one dead rank left its peers (blocked in a collective) and the
    launcher hanging.  The launcher polls: the first non-zero exit terminates the rest and is the job's exit code."""
    import time
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e["TC_BENCH_DRY_FAIL_RANK"] = "1"
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=e, capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and time.time() - t0 < 60
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]       # no JSON line from a failed job
