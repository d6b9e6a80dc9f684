"""The launch path of bench.py without a GPU: `python bench.py --gpus N` with no launcher around it must start N ranks
itself (a child torch.distributed.run, before anything touches the GPU), report n_gpus = N and fail when a rank fails.
`--dry-run` swaps the engine for stand-in payloads on gloo; everything else (argument handling, rank environment,
rendezvous on 127.0.0.1, the collectives' shapes, rank 0 printing ONE JSON line) is the real thing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks():
    r = _run(["--gpus", "2", "--dry-run", "--points", "20000", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["dry_run"] is True and out["steps"] == 2


def test_bench_dry_run_exercises_the_native_driver():
    """N > 1 defaults to the native tiled driver (libvgs_tiles.so): the dry run loads it, runs its boundary merge on records
    gathered over the process group and checks that every rank gets the same tables; --python-twin leaves it out."""
    r = _run(["--gpus", "2", "--dry-run", "--points", "20000", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["driver"].startswith("native") and out["native_merge_kept"] == 5 + 6 + 4
    r = _run(["--gpus", "2", "--dry-run", "--python-twin", "--points", "20000", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["driver"] == "python twin"


def test_bench_four_ranks_2x2():
    r = _run(["--gpus", "4", "--dry-run", "--points", "10000", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 4 and "2x2" in out["config"]["workload"]


def test_bench_refuses_a_mismatched_launcher():
    """WORLD_SIZE from a launcher must equal --gpus: a silent 1-GPU run that prints n_gpus 1 is what this guards against."""
    r = _run(["--gpus", "2", "--dry-run", "--points", "1000"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "ranks" in r.stderr


def test_bench_fails_when_a_rank_fails():
    r = _run(["--gpus", "2", "--dry-run", "--points", "20000", "--steps", "1"], env_extra={"VGS_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
