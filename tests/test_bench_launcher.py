"""``bench.py --gpus N`` without WORLD_SIZE starts its own ranks (fresh child processes, before any GPU call) and
prints ONE JSON line.  Run here in the GPU-less control-flow mode: launcher, gloo rendezvous on 127.0.0.1, barrier,
max-over-ranks timing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launcher_prints_one_line_at_world_2():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["dry_run"] is True


def test_failed_rank_ends_the_launch():
    """A rank that cannot start (an unknown flag) makes the launcher return its code instead of hanging in a barrier."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--config", "cfg3", "--steps", "x"],
                         capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode != 0
