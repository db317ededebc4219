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
    e["AASR_BENCH_RECIPE_SMALL"] = "1"      # the 256-utterance recipe instead of the 10 000-utterance one (55 + 109 GB of files)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          env=e, timeout=600)


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


import json

import pytest


def _have_device():
    import torch
    return torch.cuda.is_available() and torch.cuda.device_count() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("args,key", [(["--utts", "24", "--frames", "40000", "--steps", "2", "--warmup", "1", "--cpu-frames", "200", "--cpu-procs", "2"], "default"),
                                      (["--workload", "gmm", "--frames", "40000", "--steps", "2", "--warmup", "1", "--cpu-frames", "0"], "gmm"),
                                      (["--workload", "recipe", "--utts", "40", "--steps", "1", "--warmup", "1", "--cpu-frames", "0"], "recipe")])
def test_bench_line_on_a_gpu(args, key):
    """One JSON line on stdout with the contract's keys, for each workload (small sizes).  The default
    workload is BASELINE configs[2] -- the metric's own "GMM log-lik + MFCC" chain."""
    if not _have_device():
        pytest.skip("no HIP device on this host (bench.py has no CPU fallback)")
    r = _run(args)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["unit"] == "frames/s" and "workload" in d["config"]
    if key == "default":
        assert d["config"]["workload"].startswith("configs[2]")
        assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
        stages = {e["stage"]: e for e in d["roofline"]["stages"]}
        assert set(stages) == {"features", "lna"}
        assert all(e["bound"] == "hbm" and 0 < e["frac"] < 1 for e in stages.values())
        assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
        assert "error" not in d["config"]["configs1"] and "error" not in d["config"]["recipe_e2e"]
        assert d["config"]["configs1"]["frames_per_s"] > 0
        assert d["config"]["recipe_e2e"]["lnabytes_2"]["frames_per_s_wall"] > 0 and d["config"]["recipe_e2e"]["lnabytes_4"]["frames_per_s_wall"] > 0
        lad = d["config"]["precision_ladder"]["scoring_ms"]
        assert 0 < lad["f16x2"] < lad["bf16x3"] < lad["f32"]
        routed = d["config"]["precision_routing"]["models"]
        assert [m["share"] for m in routed] == [0.01, 0.1, 0.4]
        # (at this test's 30 000 frames the two launches of a routed model are launch-bound: no timing relation asserted)
        assert all(0 < m["states_f16x2"] < 3125 and m["scoring_ms"] > 0 and m["engine_path_scoring_plus_lna_ms"] > 0 for m in routed)
        assert "error" not in d["config"]["configs4"] and d["config"]["configs4"]["effective_precision"] == "f16x2"
        assert d["config"]["clustered"]["ms_per_million_frames"] > 0 and d["config"]["clustered"]["eval_ming"] == 0.25
        assert d["config"]["lna_check"]["max_code_difference"] <= 1
    if key == "gmm":
        assert d["config"]["workload"].startswith("configs[1]")
        assert "error" not in d["config"]["configs2"]
        assert d["config"]["configs2"]["lna_check"]["max_code_difference"] <= 1
    if key == "recipe":
        assert d["scaling"] == "strong" and d["config"]["recipe"]["utterances"] == 40


@pytest.mark.gpu
def test_two_ranks_go_through_the_whole_default_run():
    """What a one-GPU box can check of `--gpus N`: the file under torch.distributed.run with two ranks -- both on
    device 0 and the collectives over gloo (AASR_BENCH_SHARE_GPU / AASR_BENCH_BACKEND: RCCL refuses two ranks on one
    device) -- reaches the end of the default run, secondary measurements included, with every barrier matched, and
    rank 0 prints the one line with n_gpus = 2.  The line says that it is a rehearsal."""
    if not _have_device():
        pytest.skip("no HIP device on this host")
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update({"AASR_BENCH_SHARE_GPU": "1", "AASR_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0",
              "AASR_BENCH_RECIPE_SMALL": "1"})
    import socket
    with socket.socket() as sk:     # a port nobody holds right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--utts", "24", "--frames", "40000",
           "--steps", "2", "--warmup", "1", "--cpu-frames", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert "NOT a multi-GPU measurement" in d["config"]["rehearsal"]
    assert "error" not in d["config"]["configs1"] and "error" not in d["config"]["recipe_e2e"]
    assert d["config"]["recipe_e2e"]["what"].startswith("configs[3] in small: 512 utterances")
