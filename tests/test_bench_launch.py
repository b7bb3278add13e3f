"""bench.py --gpus N without a launcher starts N ranks itself, and a line never claims more ranks than ran.

No GPU here: `--dry-run` keeps the launch, the process group, the barriers and the max-over-ranks reduction
of the timed region and leaves out the GPU work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_without_launcher_starts_two_ranks():
    r = _run(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["backend"] == "gloo"


def test_single_rank_needs_no_process_group():
    r = _run(["--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 1 and line["ranks"] == 1


def test_world_mismatch_is_refused():
    # a launcher that started one rank for --gpus 2: no line, non-zero status
    r = _run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "refusing" in r.stderr


def test_failing_rank_fails_the_parent():
    # every rank rejects the workload name: the parent returns their failure
    r = _run(["--gpus", "2", "--dry-run", "--workload", "no_such_workload"])
    assert r.returncode != 0
