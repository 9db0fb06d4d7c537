"""`python bench.py --gpus N` without a launcher must start its N ranks itself (the env:// rendezvous died
on a missing MASTER_ADDR).  CPU check of the built-in launcher: N = 2, `gloo` switch, rendezvous + one reduction per rank and
no GPU work (--spawn-check); the launcher forwards rank 0's JSON line and fails when a rank fails."""
import json
import os
import subprocess
import sys

from tests.util import ROOT


def _run(extra_env, *flags, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True, timeout=timeout)


def test_builtin_launcher_starts_two_ranks_without_torchrun():
    out = _run({"DPGO_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--spawn-check")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # ONE JSON line, rank 0's
    d = json.loads(lines[0])
    assert d["spawn_check"] and d["world_size"] == 2 and d["sum_of_rank_plus_one"] == 3.0
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and [r["local_rank"] for r in d["ranks"]] == [0, 1]
    assert d["ranks"][0]["pid"] != d["ranks"][1]["pid"]


def test_builtin_launcher_reports_a_failing_rank():
    # an unknown backend makes every rank raise in init_process_group: the launcher must come back non-zero, not hang
    out = _run({"DPGO_BENCH_BACKEND": "no-such-backend", "DPGO_BENCH_SPAWN_TIMEOUT": "120"}, "--gpus", "2", "--spawn-check")
    assert out.returncode != 0


def test_under_a_launcher_the_environment_wins():
    # RANK present (torchrun's contract): no second generation of processes is spawned
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {"DPGO_BENCH_BACKEND": "gloo", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
    out = _run(env, "--gpus", "2", "--spawn-check")
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["world_size"] == 1
