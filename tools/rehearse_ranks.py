#!/usr/bin/env python
"""rehearse_ranks.py -- BASELINE configs[3] (the recipe workload) as 1 / 2 / 4 / 8 engine processes on ONE host.

The driver's 8-GPU node is not ours to launch, so what can be rehearsed on a 1-GPU box is the part of a
multi-rank recipe run that does not live on the device: N processes, each taking its Recipe::read slice
(aku/Recipe.cc:63-115) exactly as `bench.py --gpus N --workload recipe` does, sharing this host's CPU quota,
page cache and /dev/shm.  Two modes per rank count:

  shared  every rank runs the real path on the one GPU (they queue on one device and one PCIe link, so the
          aggregate is bounded by ONE device: this shows that nothing breaks or thrashes, not the scaling);
  stub    the ablation build's AASR_RECIPE_STUB=1: a block's kernels and its device -> host copy are skipped,
          file reads, uploads, the pinned result slots and the writer pools run as usual -- the host side of
          N ranks alone, which is what bounds an 8-GPU recipe run.

Every rank sizes its helper threads from usable_cores / N (aasr_set_host_share).  `--fixed-writers 16` pins the
round-2 constant instead, to show what the oversubscription cost.  Process group: gloo on 127.0.0.1 (RCCL refuses
two ranks on one device); model built on rank 0 and broadcast, as in bench.py.

    python tools/rehearse_ranks.py --ranks 1,2,4,8 --utts 4000 --out gpurun_out/ranks

writes <out>/r3_recipe_ranks_<N>.json (one per rank count, both modes inside).
"""
from __future__ import annotations

import argparse
import json
import os
import resource
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(a):
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world,
                            init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"])
    torch.cuda.set_device(0)                      # the one GPU of the box, shared by every rank
    import bench
    from aaltoasr_amd import capi, shard, synth
    capi.check(capi.lib().aasr_set_device(0))
    capi.set_host_share(world)
    names = ["mean", "var", "mix_off", "mix_idx", "mix_w"]
    model = dict(zip(names, synth.make_model(D=bench.DIM, G=bench.G, S=bench.S, comps=bench.COMPS))) if rank == 0 \
        else dict.fromkeys(names)
    if world > 1:
        model = shard.broadcast_model(model, src=0, device=None)
    gmm = capi.Gmm.from_arrays(*(model[k] for k in names))
    gmm.set_precision(3)

    def sync_all():
        torch.cuda.synchronize()
        dist.barrier()

    args = argparse.Namespace(recipe_dir=a.recipe_dir)
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    res = bench.run_recipe_workload(args, capi, synth, shard, gmm, world, rank, a.utts, a.steps, a.warmup, sync_all)
    t_all = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    timing = capi.recipe_last_timing(gmm)          # the last pass
    mine = {"rank": rank, "frames": res["frames"], "utterances": res["utterances"], "wall_s": res["wall_s"],
            "device_s": res["device_s"], "copy_s": res["copy_s"], "lna_bytes_per_step": res["lna_bytes_per_step"],
            "cpu_s_user": ru1.ru_utime - ru0.ru_utime, "cpu_s_sys": ru1.ru_stime - ru0.ru_stime,
            "last_pass": timing, "whole_call_s": t_all}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    if rank == 0:
        wall = max(r["wall_s"] for r in everyone)
        frames = sum(r["frames"] for r in everyone)
        out = {"ranks": world, "steps": a.steps, "utterances": a.utts, "frames_total": frames,
               "wall_s_per_step_slowest_rank": round(wall / a.steps, 4),
               "frames_per_s_aggregate": round(frames * a.steps / wall, 1),
               "lna_GB_per_step": round(sum(r["lna_bytes_per_step"] for r in everyone) / 1e9, 3),
               "file_write_GBps_aggregate": round(sum(r["lna_bytes_per_step"] for r in everyone) * a.steps / wall / 1e9, 2),
               "cpu_s_per_step_all_ranks": round(sum(r["cpu_s_user"] + r["cpu_s_sys"] for r in everyone) / (a.steps + a.warmup), 3),
               "device_busy_s_per_step_sum": round(sum(r["device_s"] for r in everyone) / a.steps, 4),
               "copy_busy_s_per_step_sum": round(sum(r["copy_s"] for r in everyone) / a.steps, 4),
               "writer_threads_per_rank": everyone[0]["last_pass"]["writer_threads"],
               "usable_cores": everyone[0]["last_pass"]["usable_cores"],
               "per_rank_last_pass": [r["last_pass"] for r in everyone]}
        print("REHEARSAL " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def run_world(n, mode, a):
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(n), "MASTER_PORT": str(_free_port()), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.pop("AASR_WRITER_THREADS", None)
    if a.fixed_writers:
        env["AASR_WRITER_THREADS"] = str(a.fixed_writers)
    if mode == "stub":
        env["AASR_LIBDIR"] = os.path.join(ROOT, "aaltoasr_amd", "lib_ablation")
        env["AASR_RECIPE_STUB"] = "1"
    cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--utts", str(a.utts), "--steps", str(a.steps),
           "--warmup", str(a.warmup), "--recipe-dir", a.recipe_dir]
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, err = procs[0].communicate(timeout=a.timeout)
    for p in procs[1:]:
        p.wait(timeout=a.timeout)
    for ln in out.splitlines():
        if ln.startswith("REHEARSAL "):
            return json.loads(ln[len("REHEARSAL "):])
    raise RuntimeError("rank 0 of %d (%s) printed no result:\n%s" % (n, mode, err[-3000:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--modes", default="stub,shared")
    ap.add_argument("--utts", type=int, default=4000)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--recipe-dir", default="")
    ap.add_argument("--fixed-writers", type=int, default=0)
    ap.add_argument("--timeout", type=int, default=900)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ranks"))
    ap.add_argument("--tag", default="")
    ap.add_argument("--worker", action="store_true")
    a = ap.parse_args()
    if a.worker:
        worker(a)
        return
    os.makedirs(a.out, exist_ok=True)
    base = {}
    for n in [int(x) for x in a.ranks.split(",")]:
        rec = {"what": "configs[3] rehearsal: %d engine processes on one host and ONE shared GPU, Recipe::read slices of a "
                       "%d-utterance recipe, WAV files -> 2-byte LNA files (see tools/rehearse_ranks.py)" % (n, a.utts),
               "fixed_writer_threads": a.fixed_writers or None, "modes": {}}
        for mode in a.modes.split(","):
            t0 = time.perf_counter()
            r = run_world(n, mode, a)
            r["launch_to_exit_s"] = round(time.perf_counter() - t0, 1)
            base.setdefault(mode, r["frames_per_s_aggregate"] if n == 1 else None)
            if base.get(mode):
                r["vs_one_rank"] = round(r["frames_per_s_aggregate"] / base[mode], 3)
            rec["modes"][mode] = r
            print("ranks %d %-6s %6.2f M frames/s aggregate, %5.1f GB/s of files, %d writers/rank, cpu %.1f s/step"
                  % (n, mode, r["frames_per_s_aggregate"] / 1e6, r["file_write_GBps_aggregate"],
                     r["writer_threads_per_rank"], r["cpu_s_per_step_all_ranks"]), flush=True)
        path = os.path.join(a.out, "r3_recipe_ranks_%d%s.json" % (n, a.tag))
        json.dump(rec, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
