"""CPU suite: `python bench.py --gpus N` starts N ranks itself (child torch.distributed.run) and rank 0 prints ONE JSON line
with n_gpus = N.  --dry-run skips the device work (there is no GPU here): launcher, rendezvous (gloo through the
PEPS_BENCH_BACKEND hook), barrier, max-over-ranks reduction and the JSON contract are what is covered."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = dict(os.environ, PEPS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks():
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run"])
    assert out["n_gpus"] == 2 and out["ranks_reported_by_backend"] == 2
    assert out["dry_run"] is True and out["value"] is None          # a dry run never carries a number
    assert out["scaling"] == "weak" and out["metric"] == "configuration-amplitudes/sec"


def test_bench_gpus_1_dry_run_is_single_process():
    out = _run(["--gpus", "1", "--dry-run"])
    assert out["n_gpus"] == 1 and out["ranks_reported_by_backend"] == 1
