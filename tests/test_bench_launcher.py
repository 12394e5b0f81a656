"""bench.py's rank launcher: `python bench.py --gpus N` must run N ranks, not one.  CPU rehearsal through
the `--dry-run` mode (gloo, no model, no kernels): same argument parsing, child launch (torch.distributed.run), rank
environment, world-size check and JSON line as the GPU run.  Reference launch it stands for: `torchrun --nproc-per-node N
-m multimeditron train` (docs/source/guides/training.rst:121,176-185)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_launches_two_ranks():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["dry_run"] is True


def test_gpus_1_runs_in_process():
    r = _run(["--gpus", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_world_size_mismatch_fails_loudly():
    # a 1-rank job claiming --gpus 2 must not print a line labelled n_gpus=2 (or any line): non-zero exit
    r = _run(["--gpus", "2", "--dry-run"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                            "MASTER_PORT": "29731"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
