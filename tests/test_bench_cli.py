"""bench.py's launcher behaviour that can be checked without a GPU: a world that cannot be
started is refused loudly, never reported as a smaller one."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          env=e, timeout=300)


def test_gpus_flag_is_never_silently_reduced():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(["--gpus", str(have + 2)])
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "exposes %d HIP device(s)" % have in r.stderr


def test_world_size_and_gpus_flag_must_agree():
    r = _run(["--gpus", "1"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "WORLD_SIZE=2" in r.stderr
