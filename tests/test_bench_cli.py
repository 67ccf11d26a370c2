"""bench.py's launcher logic, without a GPU: `--gpus N` must start N ranks or fail loudly, never time one GPU silently."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_more_gpus_than_visible_is_an_error():
    import torch
    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stdout + r.stderr)
    assert '"metric"' not in r.stdout   # no bench line for a job that did not run


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "one rank per GPU" in (r.stdout + r.stderr)
