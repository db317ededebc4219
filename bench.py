#!/usr/bin/env python
"""bench.py -- frames/sec of the acoustic-likelihood hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload gmm|full|recipe]

A "step" is one pass of the hot path over one batch of synthetic input:
  full   (BASELINE configs[2], the default `value` -- the metric's own "GMM log-lik + MFCC"
         workload): 1 h of 16 kHz int16 audio as 360 x 10 s utterances (449 280 frames), resident
         in HBM -> MFCC chain -> 50 000-Gaussian / 3 125-state x 16 diagonal HmmSet scoring ->
         2-byte LNA codes.  The default run also reports the per-stage split, the HBM rooflines
         of the feature chain and the LNA pass (roofline.stages), the fraction of LNA codes equal
         to the oracle's on a sampled utterance, the scoring stage under every arithmetic the engine
         has (config.precision_ladder: f16x2 / bf16x3 / f32) and under per-state precision routing
         (config.precision_routing: 1 / 10 / 40 % of the states over the f16x2 limits), the clustered
         pass pyrectool runs (config.clustered), configs[1] (config.configs1), configs[4]
         (config.configs4) and configs[3] at full size -- the 10 000-utterance recipe, WAV files in,
         2-byte and 4-byte LNA files out (config.recipe_e2e; a 256-utterance miniature when /dev/shm
         has no room) -- so one driver run carries all of them.
  gmm    (BASELINE configs[1]): 1 000 000 x 39 float32 frames, resident in HBM, against the same
         model -> [F x S] state log-likelihoods (k_gmm_diag_score_pl, two fp16 terms); no MFCC, no LNA.
  recipe (BASELINE configs[3]): a recipe of --utts (default 10 000) seeded utterances of
         U(2 s, 12 s) as WAV files, sliced over the ranks with Recipe::read's rule
         (aku/Recipe.cc:63-115), read -> features -> scoring -> 2-byte LNA files written; value =
         total frames / wall time of the slowest rank ("strong" scaling: the recipe is fixed),
         the device-only rate is reported beside it.

Multi-GPU: one process per GPU.  Under torch.distributed.run (WORLD_SIZE set) the rank takes its
place; without it `--gpus N` (N > 1) starts the N ranks itself by re-executing this file under
`python -m torch.distributed.run --nproc-per-node N`.  The model is built on rank 0 and broadcast
once over RCCL; there is no collective on the scoring path.  `n_gpus` in the line is the world
size RCCL reported, never the flag.

Prints ONE JSON line on rank 0.  The roofline block prices the dominant kernel in ALGORITHMIC
flops: 4*dim = 156 flop per frame x Gaussian pair against the ceiling of
the pipe it runs on -- default (`--precision f16x2`): the dense FP16 matrix peak 2500 TFLOP/s / 3
fp16 products per product; `--precision bf16x3`: the dense BF16 peak / 6; `--precision f32`: the
dense FP32 matrix peak, 157.3 TFLOP/s -- with the ratio to the FP32 matrix peak next to it.
A scoring call is two launches: k_frame_operand (the frame operand of the split-term kernel,
formed once per step: ~0.08 ms) and k_gmm_diag_score_pl; both are timed with HIP events on the
launch stream (roofline.scoring_call_ms, roofline.frame_operand_kernel_ms) and roofline.kernel_ms,
the figure the roofline prices, is their difference -- the scoring kernel's own duration, which
is what the rocprofv3 summary under profiles/ shows for it.  The cpu_baseline block times the
oracle's reference-shaped scalar double loop on this host (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3
BF16_MATRIX_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0
DIM = 39
G = 50000
S = 3125
COMPS = 16
NOMINAL_SCLK_MHZ = 2400.0   # the clock the datasheet peaks are quoted at
METRIC = "frames/sec GMM log-lik (39-d, 50k Gauss) + MFCC, 1/2/4/8 MI355X"   # BASELINE.json


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["gmm", "full", "recipe"],
                    default=os.environ.get("AASR_BENCH_WORKLOAD", "full"))
    ap.add_argument("--frames", type=int, default=1_000_000, help="frames per GPU per step (gmm workload, and the configs[1] figure of the default run)")
    ap.add_argument("--utts", type=int, default=0,
                    help="utterances: per GPU per step for `full` (default 360 x 10 s), in the whole "
                         "recipe for `recipe` (default 10 000 x U(2 s, 12 s))")
    ap.add_argument("--out-pitch", choices=["dense", "aligned"], default=os.environ.get("AASR_BENCH_OUT_PITCH", "aligned"),
                    help="gmm workload: output rows dense [F x S] or padded to whole 128-byte lines")
    ap.add_argument("--cpu-frames", type=int, default=20000, help="frames timed on the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="processes of the all-cores CPU baseline (-1 = one per usable core, 0 = skip)")
    ap.add_argument("--cpu-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--precision", choices=["f16x2", "bf16x3", "f32"], default=os.environ.get("AASR_BENCH_PRECISION", "f16x2"),
                    help="contraction arithmetic of the scoring kernel (all meet the 1e-4 parity bar; f16x2 is the "
                         "library default for models whose conditioning allows it)")
    ap.add_argument("--recipe-dir", default=os.environ.get("AASR_BENCH_RECIPE_DIR", ""),
                    help="where the recipe workload keeps its WAV inputs and LNA outputs "
                         "(default: /dev/shm when it has room, else the system temp directory)")
    ap.add_argument("--secondary", type=int, default=int(os.environ.get("AASR_BENCH_SECONDARY", "1")),
                    help="full / gmm workloads: also measure the other single-GPU config (configs[1] resp. configs[2]) "
                         "and a small recipe run (1 = yes)")
    return ap.parse_args()


# --------------------------------------------------------------------------- CPU baseline --

def cpu_worker(n_frames):
    """One process of the all-cores CPU baseline: the reference scales over cores by
    running independent phone_probs processes on recipe slices (-B n -I k,
    aku/Recipe.cc:63-115), so N copies of the single-thread loop is its native mode."""
    from aaltoasr_amd import synth
    from oracle import oracle as O
    om = O.DiagModel(*synth.make_model(D=DIM, G=G, S=S, comps=COMPS))
    cf = synth.make_frames(n_frames, DIM, seed=synth.SEED + 99).astype(np.float64)
    om.cpu_baseline(cf[:20])
    c0 = time.perf_counter()
    om.cpu_baseline(cf)
    print("CPUWORKER %d %.6f" % (n_frames, time.perf_counter() - c0), flush=True)


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU boxes expose 256 logical CPUs but schedule only a fraction of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_all_cores(n_procs, n_frames):
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(n_frames)]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for _ in range(n_procs)]
    rate, done = 0.0, 0
    for p in procs:
        out, _ = p.communicate()
        for ln in out.splitlines():
            if ln.startswith("CPUWORKER"):
                _, k, dt = ln.split()
                rate += int(k) / float(dt)
                done += 1
    return rate, done, time.perf_counter() - t0


def cpu_baseline(args, model):
    from aaltoasr_amd import synth
    from oracle import oracle as O
    O.build()
    om = O.DiagModel(*model)
    cf = synth.make_frames(args.cpu_frames, DIM, seed=synth.SEED + 99).astype(np.float64)
    om.cpu_baseline(cf[:50])
    c0 = time.perf_counter()
    om.cpu_baseline(cf)
    cdt = time.perf_counter() - c0
    cpu = {
        "value": round(args.cpu_frames / cdt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": "%d of the step's frames, same 50k-Gaussian model, oracle/aasr_oracle.c "
                  "orc_cpu_baseline_score (double, per-frame per-Gaussian exp, linear mixture sum, "
                  "float-cast normalisation) single thread, %.1f s" % (args.cpu_frames, cdt),
        "host": _cpu_model(), "host_cores": os.cpu_count(), "usable_cores": usable_cores(),
    }
    n_procs = usable_cores() if args.cpu_procs < 0 else args.cpu_procs
    if n_procs and n_procs > 1:
        per = max(200, args.cpu_frames // 10)
        rate, done, wall = cpu_all_cores(n_procs, per)
        if done:
            cpu["all_cores"] = {
                "value": round(rate, 1), "unit": "frames/s", "cores": done,
                "sample": "%d independent single-thread processes (the reference's -B/-I mode), %d frames "
                          "each, sum of the per-process rates, %.1f s wall incl. start-up" % (done, per, wall)}
    return cpu


# ------------------------------------------------------------------------ rank start-up --

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """`--gpus N` without a launcher: start the N ranks under torch.distributed.run and hand
    their output through.  Fails loudly when the node has fewer devices."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: this node exposes %d HIP device(s); there is no CPU fallback "
                         "and a smaller world is never reported as %d GPUs" % (args.gpus, have, args.gpus))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


# ------------------------------------------------------------------------------ recipe --

def _recipe_dir(args, need_bytes):
    if args.recipe_dir:
        os.makedirs(args.recipe_dir, exist_ok=True)
        return tempfile.mkdtemp(prefix="aasr_recipe_", dir=args.recipe_dir)
    for base in ("/dev/shm", tempfile.gettempdir()):
        try:
            if os.path.isdir(base) and os.access(base, os.W_OK) and shutil.disk_usage(base).free > 1.3 * need_bytes:
                return tempfile.mkdtemp(prefix="aasr_recipe_", dir=base)
        except OSError:
            pass
    raise RuntimeError("no directory with %.1f GB free for the recipe workload (give --recipe-dir)" % (need_bytes / 1e9))


def recipe_lengths(n_utts, sample_rate=16000):
    """Seeded utterance lengths U(2 s, 12 s) in samples."""
    from aaltoasr_amd import synth
    rng = np.random.default_rng(synth.SEED + 4000)
    return (rng.uniform(2.0, 12.0, n_utts) * sample_rate).astype(np.int64)


def write_recipe_inputs(workdir, lengths, first, count, sample_rate=16000):
    """WAV files of utterances [first, first + count): seeded slices of one 60-s pool of the
    synthetic signal (noise + three sinusoids) -- the audio content does not change the work."""
    import wave
    from aaltoasr_amd import synth
    pool = synth.make_audio(60 * sample_rate, seed=synth.SEED + 4001, sample_rate=sample_rate)
    rng = np.random.default_rng(synth.SEED + 4002 + first)
    for i in range(first, first + count):
        n = int(lengths[i])
        o = int(rng.integers(0, len(pool) - n))
        with wave.open(os.path.join(workdir, "u%05d.wav" % i), "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(sample_rate)
            w.writeframes(pool[o:o + n].astype("<i2").tobytes())


def run_recipe_e2e(args, capi, synth, shard, gmm, world, rank, sync_all, max_over_ranks, sum_over_ranks):
    """configs[3] inside the default run: the 10 000-utterance recipe (8.73 M frames; WAV files in, LNA files out, every
    rank its Recipe::read slice) once with 2-byte LNA files and once with the 4-byte files pyrectool asks for
    (pyrectool/rectool.py:655-666 passes --lnabytes=4) -- when the box has room for them in /dev/shm (55 / 109 GB per
    pass), else a 256-utterances-per-rank miniature.  One untimed small pass first (device buffers, pinned slots)."""
    feat = capi.Feat(synth.make_feature_config())
    n_full = 10000
    lengths_full = recipe_lengths(n_full)
    frames_full = int(sum(feat.eof_frame(int(n)) for n in lengths_full))
    need = frames_full * S * 4 + int(lengths_full.sum()) * 2
    roomy = False
    try:
        roomy = (not args.recipe_dir and os.access("/dev/shm", os.W_OK) and
                 shutil.disk_usage("/dev/shm").free > 1.25 * need) or \
                (args.recipe_dir and shutil.disk_usage(args.recipe_dir).free > 1.25 * need)
    except OSError:
        pass
    if os.environ.get("AASR_BENCH_RECIPE_SMALL") == "1":
        roomy = False
    # every rank must take the same branch
    roomy = max_over_ranks(0.0 if roomy else 1.0) == 0.0
    n_utts = n_full if roomy else 256 * world
    lengths = recipe_lengths(n_utts)
    frames_all = np.array([feat.eof_frame(int(n)) for n in lengths], np.int64)
    first, count = shard.rank_slice(n_utts, world, rank)
    my_frames = int(frames_all[first:first + count].sum())
    workdir = _recipe_dir(args, my_frames * S * 4 + int(lengths[first:first + count].sum()) * 2)
    out = {"what": "configs[3]%s: %d utterances U(2 s, 12 s) (%d frames), WAV files in %s -> LNA files, one pass each, "
                   "Recipe::read slices over %d rank(s)" % ("" if roomy else " in small", n_utts, int(frames_all.sum()),
                                                            os.path.dirname(workdir), world),
           "full_size": bool(roomy), "utterances": n_utts, "frames": int(frames_all.sum())}
    try:
        write_recipe_inputs(workdir, lengths, first, count)
        recipe = os.path.join(workdir, "r%d.recipe" % rank)
        with open(recipe, "w") as f:
            for i in range(n_utts):
                f.write("audio=%s lna=u%05d.lna\n" % (os.path.join(workdir, "u%05d.wav" % i), i))
        warm = os.path.join(workdir, "w%d.recipe" % rank)
        with open(warm, "w") as f:
            for i in range(first, first + min(count, 64)):
                f.write("audio=%s lna=u%05d.lna\n" % (os.path.join(workdir, "u%05d.wav" % i), i))

        def one(path, lnabytes, outdir, sliced):
            os.makedirs(outdir, exist_ok=True)
            return capi.run_recipe(feat, gmm, path, lnabytes=lnabytes, out_dir=outdir,
                                   num_batches=world if (sliced and world > 1) else 0,
                                   batch_index=rank + 1 if (sliced and world > 1) else 0)
        for nb in (2, 4):
            one(warm, nb, os.path.join(workdir, "lna", "warm%d" % nb), False)
            shutil.rmtree(os.path.join(workdir, "lna"), ignore_errors=True)
            sync_all()
            t0 = time.perf_counter()
            st = one(recipe, nb, os.path.join(workdir, "lna", "b%d" % nb), True)
            sync_all()
            wall = max_over_ranks(time.perf_counter() - t0)
            assert st.frames == my_frames and st.utterances == count, (st.frames, my_frames, st.utterances, count)
            tot = sum_over_ranks(float(my_frames))
            devs = max_over_ranks(st.seconds_device)
            out["lnabytes_%d" % nb] = {"frames_per_s_wall": round(tot / wall, 1), "wall_s": round(wall, 4),
                                       "frames_per_s_device_only": round(tot / max(devs, 1e-9), 1),
                                       "pcie_copy_out_s_rank0": round(st.seconds_copy_out, 4),
                                       "lna_GB_written_per_rank": round(my_frames * S * nb / 1e9, 3)}
            shutil.rmtree(os.path.join(workdir, "lna"), ignore_errors=True)
        # the keys the round-3 line carried, for the 2-byte pass
        out["frames_per_s_wall"] = out["lnabytes_2"]["frames_per_s_wall"]
        out["frames_per_s_device_only"] = out["lnabytes_2"]["frames_per_s_device_only"]
        out["wall_s"] = out["lnabytes_2"]["wall_s"]
        return out
    finally:
        shutil.rmtree(workdir, ignore_errors=True)


def run_recipe_workload(args, capi, synth, shard, gmm, world, rank, n_utts, steps, warmup, sync_all, workdir=None):
    """configs[3]: the rank's Recipe::read slice, files in, LNA files out.  Returns a dict
    with the local figures; the caller aggregates over ranks."""
    feat = capi.Feat(synth.make_feature_config())
    lengths = recipe_lengths(n_utts)
    frames_all = np.array([feat.eof_frame(int(n)) for n in lengths], np.int64)
    first, count = shard.rank_slice(n_utts, world, rank)
    my_frames = int(frames_all[first:first + count].sum())
    out_bytes = my_frames * S * 2
    in_bytes = int(lengths[first:first + count].sum()) * 2
    own = workdir is None
    if own:
        workdir = _recipe_dir(args, out_bytes * (steps + warmup) + in_bytes)
    try:
        write_recipe_inputs(workdir, lengths, first, count)
        # every rank holds the WHOLE recipe text and lets Recipe::read's -B/-I rule pick its slice,
        # exactly what N reference processes would do (missing files of other slices are never opened)
        recipe = os.path.join(workdir, "r%d.recipe" % rank)
        with open(recipe, "w") as f:
            for i in range(n_utts):
                f.write("audio=%s lna=u%05d.lna\n" % (os.path.join(workdir, "u%05d.wav" % i), i))
        # every pass writes NEW files into its own directory, as a recipe run does (replacing
        # existing files is a different and much slower file-system path: the old pages are freed
        # under the rename -- measured 10 GB/s whatever the thread count against 58 GB/s for fresh
        # files from 8 threads on the same tmpfs, tools/exp_tmpfs.py)
        passes = [0]

        def one():
            outdir = os.path.join(workdir, "lna", "pass%d" % passes[0])
            passes[0] += 1
            os.makedirs(outdir, exist_ok=True)
            return capi.run_recipe(feat, gmm, recipe, lnabytes=2, out_dir=outdir,
                                   num_batches=world if world > 1 else 0, batch_index=rank + 1 if world > 1 else 0)
        for _ in range(warmup):
            one()
        sync_all()
        t0 = time.perf_counter()
        dev_s = copy_s = 0.0
        for _ in range(steps):
            st = one()
            dev_s += st.seconds_device
            copy_s += st.seconds_copy_out
        sync_all()
        wall = time.perf_counter() - t0
        assert st.frames == my_frames and st.utterances == count, (st.frames, my_frames, st.utterances, count)
        return {"frames": my_frames, "utterances": count, "wall_s": wall, "device_s": dev_s, "copy_s": copy_s, "steps": steps,
                "dir": os.path.dirname(workdir) if own else workdir, "lna_bytes_per_step": out_bytes,
                "audio_bytes_per_step": in_bytes}
    finally:
        if own:
            shutil.rmtree(workdir, ignore_errors=True)


# -------------------------------------------------------------------------------- main --

def main():
    args = parse_args()
    if args.cpu_worker > 0:
        cpu_worker(args.cpu_worker)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)
    # stdout carries exactly one JSON line: anything native libraries print there (RCCL's version
    # banner, rocm-smi) is sent to stderr for the life of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if env_world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; refusing to "
                         "report one as the other" % (args.gpus, env_world))
    # AASR_BENCH_FORCE_DIST=1 takes the multi-rank code path (RCCL init, model broadcast, barrier,
    # max-over-ranks) at world size 1 too, so it can be exercised on a one-GPU box under torchrun
    distributed = env_world > 1 or os.environ.get("AASR_BENCH_FORCE_DIST") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: there is no CPU fallback for the product path")
    # Rehearsal switches (not what the driver runs): AASR_BENCH_SHARE_GPU=1 puts every rank on device 0 and
    # AASR_BENCH_BACKEND=gloo carries the few collectives over TCP -- RCCL refuses two ranks on one device, and a
    # one-GPU box is all there is to check that an N-rank run of this file goes through without a mismatched barrier
    backend = os.environ.get("AASR_BENCH_BACKEND", "nccl")
    if os.environ.get("AASR_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no device (%d visible)" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    world = 1
    if distributed:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=env_world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=env_world)
        world = dist.get_world_size()       # what RCCL actually connected

    from aaltoasr_amd import build, capi, shard, synth
    if rank == 0:
        build.build()
    if distributed:
        dist.barrier()
    capi.check(capi.lib().aasr_set_device(local_rank))

    # ---- model: built on rank 0 and broadcast once (RCCL over xGMI) -- the only
    # collective of the job; the reference has every process re-read the
    # .gk/.mc files instead (phone_probs.cc:96-110)
    names = ["mean", "var", "mix_off", "mix_idx", "mix_w"]
    if rank == 0:
        model = dict(zip(names, synth.make_model(D=DIM, G=G, S=S, comps=COMPS)))
    else:
        model = dict.fromkeys(names)
    if distributed:
        model = shard.broadcast_model(model, src=0, device=dev if backend == "nccl" else None)
    mean, var, off, idx, w = (model[k] for k in names)
    gmm = capi.Gmm.from_arrays(mean, var, off, idx, w)
    PREC = {"f16x2": 4, "bf16x3": 3, "f32": 0}[args.precision]   # AASR_PREC_F16X2 / AASR_PREC_BF16X3 / AASR_PREC_F32
    gmm.set_precision(PREC)
    if args.precision == "f16x2" and gmm.effective_precision() != 4:
        raise SystemExit("bench.py: the synthetic model was not packed for the f16x2 kernel (effective precision %d)"
                         % gmm.effective_precision())
    rows = gmm.expanded_rows
    stream = torch.cuda.current_stream()

    def sync_all():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not distributed:
            return x
        t = torch.tensor([x], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        if not distributed:
            return x
        t = torch.tensor([x], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    scaling = "weak"
    extra_cfg = {}
    if args.workload == "recipe":
        # ---- configs[3]: wall clock over files in -> LNA files out; strong scaling
        n_utts = args.utts or 10000
        res = run_recipe_workload(args, capi, synth, shard, gmm, world, rank, n_utts, args.steps, args.warmup, sync_all)
        elapsed = max_over_ranks(res["wall_s"])
        total_frames = sum_over_ranks(float(res["frames"]))
        dev_max = max_over_ranks(res["device_s"])
        F = res["frames"]
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_frames * args.steps / elapsed
        scaling = "strong"
        workload = ("configs[3]: recipe of %d synthetic utterances U(2 s, 12 s) (%d frames) sliced over %d rank(s) by "
                    "Recipe::read's rule, WAV files -> MFCC chain -> %d-Gaussian scoring -> 2-byte LNA files (%s), "
                    "wall clock incl. file reads, PCIe and file writes" % (n_utts, int(total_frames), world, G, res["dir"]))
        extra_cfg = {"recipe": {"utterances": n_utts, "frames_total": int(total_frames),
                                "frames_per_s_wall": round(value, 1),
                                "frames_per_s_device_only": round(total_frames * args.steps / max(dev_max, 1e-9), 1),
                                "wall_s_per_step": round(elapsed / args.steps, 4),
                                "device_s_per_step_slowest_rank": round(dev_max / args.steps, 4),
                                "pcie_copy_out_s_per_step_rank0": round(res["copy_s"] / args.steps, 4),
                                "lna_GB_written_per_step_rank0": round(res["lna_bytes_per_step"] / 1e9, 3)}}
        feat_runner = None

        def score_only():
            pass
    elif args.workload == "gmm":
        F = args.frames
        gen = torch.Generator(device=dev)
        gen.manual_seed(synth.SEED + 17 * rank)
        d_frames = torch.randn((F, DIM), generator=gen, device=dev, dtype=torch.float32)
        # output rows either dense (pitch S, what aasr_gmm_score hands to a host caller) or padded to
        # whole 128-byte lines (pitch S rounded up to 32 floats: the layout the device-resident chain
        # aasr_gmm_score_dev_pitched -> aasr_lna_encode_dev_pitched uses)
        pitch = (S + 31) // 32 * 32 if (args.out_pitch == "aligned" and gmm.score_pitch_ok()) else S
        d_out = torch.empty((F, pitch), device=dev, dtype=torch.float32)
        workload = "configs[1]: batched diag-GMM log-likelihood, %d x %d-d frames x %d Gaussians (%d states x %d), output row pitch %d floats; no MFCC and no LNA in the timed region (see config.configs2 for the full chain)" % (
            F, DIM, G, S, COMPS, pitch)

        if pitch == S:
            def step():
                gmm.score_dev(d_frames, d_out, stream)
        else:
            def step():
                gmm.score_dev_pitched(d_frames, d_out, pitch, stream)
        score_only = step
    else:
        from aaltoasr_amd import pipeline
        runner = pipeline.FullChainBench(gmm, n_utts=args.utts or 360, seconds=10.0, rank=rank, device=dev)
        F = runner.total_frames
        workload = ("configs[2]: full MFCC chain (FFT -> mel -> log -> DCT -> delta/delta-delta, CMS, normalisation) + %d-Gaussian "
                    "scoring + 2-byte LNA codes on 1 h of synthetic 16 kHz audio as %d x 10 s utterances (%d frames), "
                    "int16 samples resident in HBM" % (G, args.utts or 360, F))
        step = runner.step
        score_only = runner.score_only

    if args.workload != "recipe":
        for _ in range(args.warmup):
            step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync_all()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * F * args.steps / elapsed

    # ---- dominant-kernel time: HIP events on the launch stream, scoring only
    roofline = None
    dtype = "f32"
    if args.workload != "recipe":
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        kreps = max(1, args.steps)
        torch.cuda.synchronize()
        ev0.record(stream)
        for _ in range(kreps):
            score_only()
        ev1.record(stream)
        torch.cuda.synchronize()
        call_ms = ev0.elapsed_time(ev1) / kreps
        # a scoring call is k_frame_operand (the split-term frame operand, formed once per call) + the scoring kernel:
        # the roofline prices the kernel on ITS duration -- the call's minus the operand launch's, timed the same way
        fop_ms = -1.0
        try:
            d_fr = d_frames if args.workload == "gmm" else runner.d_fea
            fop_ms = gmm.frame_operand_ms(d_fr, max(3, kreps), stream)
        except Exception:
            pass
        k_ms = call_ms - fop_ms if 0 < fop_ms < 0.2 * call_ms else call_ms
        algo_flop = 4.0 * DIM * float(F) * float(rows)
        achieved = algo_flop / (k_ms * 1e-3) / 1e12
        if args.precision == "f32":
            kernel, peak, dtype = "k_gmm_diag_score_tracks<40,true>", FP32_MATRIX_PEAK_TFLOPS, "f32"
            peak_note = "dense FP32 matrix peak (v_mfma_f32_32x32x2_f32), MI355X_MICROARCH.md"
            products = 1.0
        elif args.precision == "bf16x3":
            # f32-accurate arithmetic: every f32 operand is carried as three bf16 terms (24 significant
            # bits), six bf16 matrix-core products per f32 product (terms below 2^-16 dropped), f32
            # accumulation; it meets the same 1e-4 parity bar as the plain f32 kernel (--precision f32).
            # The ceiling for ALGORITHMIC flops on the bf16 pipe is therefore the dense bf16 peak / 6.
            kernel, peak, dtype = ("k_gmm_diag_score_bf16x3<5,true,false,true,3>", BF16_MATRIX_PEAK_TFLOPS / 6.0,
                                   "f32 (3-term bf16 split on the matrix cores, f32 accumulate)")
            peak_note = ("dense BF16 matrix peak 2500 TFLOP/s / 6 bf16 products per f32-accurate product; "
                         "executed matrix flops = 6 * 160/156 * achieved")
            products = 6.0
        else:
            # two fp16 terms per operand (22 significant bits), three fp16 matrix-core products per product
            # (lo*lo dropped), f32 accumulation: same 1e-4 parity bar (state-level worst 3.4e-5 on 10^7 states,
            # tools/exp_fp16_split.py), chosen per model by its conditioning.  Ceiling for ALGORITHMIC flops:
            # the dense fp16 peak / 3.
            kernel, peak, dtype = ("k_gmm_diag_score_pl<5,true,false,true,2>", BF16_MATRIX_PEAK_TFLOPS / 3.0,
                                   "f32 (2-term fp16 split on the matrix cores, f32 accumulate)")
            peak_note = ("dense FP16 matrix peak 2500 TFLOP/s / 3 fp16 products per product; "
                         "executed matrix flops = 3 * 160/156 * achieved")
            products = 3.0
        roofline = {
            "bound": "mfma", "kernel": kernel, "achieved": round(achieved, 3),
            "peak": round(peak, 2), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None,
            "kernel_ms": round(k_ms, 4), "scoring_call_ms": round(call_ms, 4),
            "frame_operand_kernel_ms": round(fop_ms, 4) if fop_ms > 0 else None,
            # the same fraction priced on the whole scoring call (k_frame_operand is a launch of every split-term
            # call): `frac` is the dominant kernel's own, this one what a caller of aasr_gmm_score_dev sees
            "frac_of_scoring_call": round(algo_flop / (call_ms * 1e-3) / 1e12 / peak, 4),
            "kernel_ms_is": ("scoring call minus the separately timed k_frame_operand launch" if k_ms != call_ms
                             else "the scoring call (no separate operand launch timed)"),
            "algorithmic_flop_per_launch": algo_flop, "peak_note": peak_note,
            "frac_of_fp32_matrix_peak": round(achieved / FP32_MATRIX_PEAK_TFLOPS, 4),
        }
        # clock / power while the kernel runs: the bf16 matrix pipe draws the chip into its
        # power cap, so the nominal-clock peak above is not what the silicon offers
        obs = _observe_clock(lambda: [score_only() for _ in range(max(8, int(1500.0 / max(k_ms, 1e-3))))],
                             torch) if rank == 0 else None
        if obs:
            roofline.update(obs)
            if obs.get("sclk_mhz"):
                adj = peak * obs["sclk_mhz"] / NOMINAL_SCLK_MHZ
                roofline["frac_of_clock_adjusted_peak"] = round(achieved / adj, 4)
        if args.precision != "f32":
            probe = _pipe_probe(run=(rank == 0 and world == 1))
            if probe:
                executed = achieved * products * 160.0 / 156.0
                probe["executed_TFLOPs"] = round(executed, 1)
                probe["frac_of_probe"] = round(executed / probe["TFLOPs_" + probe["reference"]], 4)
                roofline["matrix_pipe_probe"] = probe
        td = _pmc_traffic(F, args.precision, aligned=(args.workload == "full" or args.out_pitch == "aligned"))
        if td:
            roofline["traffic"] = td["bytes"]
            roofline["traffic_detail"] = td

    # ---- secondary measurements of the default run (every rank takes part: they hold barriers)
    if args.workload == "full" and args.secondary:
        # the per-stage split of the timed workload and its HBM-bound stages priced against 8 TB/s
        try:
            split = runner.stage_split(max(3, min(args.steps, 10)))
            extra_cfg["stage_ms"] = split
            extra_cfg["hbm_bytes_written_per_frame"] = runner.bytes_written_per_frame()
            roofline["stages"] = _stage_rooflines(runner, split)
            if rank == 0:
                extra_cfg["lna_check"] = _lna_check(runner, gmm, mean, var, off, idx, w,
                                                    PREC)
        except Exception as e:
            extra_cfg["stage_ms"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            extra_cfg.update(_measure_precisions(torch, capi, synth, gmm, runner, (mean, var, off, idx, w), PREC,
                                                 with_models=(rank == 0 and world == 1)))
        except Exception as e:
            extra_cfg["precision_ladder"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0 and world == 1:   # (one-GPU figures: the other ranks of a multi-GPU run would idle ~40 s at the next barrier)
            try:
                extra_cfg["fitted_model"] = _measure_fitted_models(torch, capi, synth, pipeline, gmm, runner, dev, PREC)
            except Exception as e:
                extra_cfg["fitted_model"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            runner.release()
            torch.cuda.empty_cache()
            extra_cfg["configs1"] = _measure_configs1(torch, synth, gmm, rank, dev, stream, sync_all, max_over_ranks,
                                                      world, args)
        except Exception as e:
            extra_cfg["configs1"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0:
            try:
                torch.cuda.empty_cache()
                extra_cfg["clustered"] = _measure_clustered(torch, capi, synth, (mean, var, off, idx, w), dev, stream,
                                                            args.frames)
            except Exception as e:
                extra_cfg["clustered"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                torch.cuda.empty_cache()
                extra_cfg["configs4"] = _measure_configs4(torch, capi, synth, dev, stream)
            except Exception as e:
                extra_cfg["configs4"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if args.workload == "gmm" and args.secondary:
        try:
            del d_out
            torch.cuda.empty_cache()
            extra_cfg["configs2"] = _measure_configs2(torch, capi, synth, gmm, rank, dev, stream, sync_all,
                                                      max_over_ranks, world, mean, var, off, idx, w,
                                                      PREC)
        except Exception as e:  # the headline must survive a failure of the extras
            extra_cfg["configs2"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if args.workload in ("gmm", "full") and args.secondary:
        try:
            torch.cuda.empty_cache()
            extra_cfg["recipe_e2e"] = run_recipe_e2e(args, capi, synth, shard, gmm, world, rank, sync_all, max_over_ranks,
                                                     sum_over_ranks)
        except Exception as e:
            extra_cfg["recipe_e2e"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- CPU baseline (rank 0, N=1 only): oracle's reference-shaped loop
    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        cpu = cpu_baseline(args, (mean, var, off, idx, w))

    if rank == 0:
        cfg = {"workload": workload, "frames_per_gpu_per_step": F, "dim": DIM, "gaussians": G,
               "states": S, "components_per_state": COMPS,
               "sharding": "frames/utterances per rank, no collective on the scoring path",
               "rccl_world_size": world if distributed else None, "gpus_flag": args.gpus}
        cfg.update(extra_cfg)
        if backend != "nccl" or os.environ.get("AASR_BENCH_SHARE_GPU") == "1":
            cfg["rehearsal"] = ("NOT a multi-GPU measurement: %d ranks, backend %s, %s" % (
                world, backend, "all on device 0" if os.environ.get("AASR_BENCH_SHARE_GPU") == "1" else "one device each"))
        line = {
            "metric": METRIC,
            "metric_note": "value is the workload named in config.workload (default: configs[2], the metric's MFCC-inclusive chain); "
                           "the GMM-only rate of configs[1] is config.configs1.frames_per_s",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": cfg, "roofline": roofline, "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def _lna_check(runner, gmm, mean, var, off, idx, w, restore_precision=3):
    """Fraction of one utterance's 2-byte LNA codes (as the timed configs[2] step left them on the
    device) that equal the oracle's restatement of phone_probs on the same audio and model."""
    from oracle import oracle as O
    from aaltoasr_amd import capi as A
    O.build()
    u = 1
    ch = O.FeatureChain(runner.cfg_text)
    om = O.DiagModel(mean, var, off, idx, w)
    nfr = int(runner.frame_off[u + 1] - runner.frame_off[u])
    fea = ch.generate(runner.utts[u], 0, nfr)
    _, lik = om.score(fea, want_lik=True)
    _, by_ref = O.lna_encode(lik, True, 2)
    got = runner.d_bytes[int(runner.frame_off[u]):int(runner.frame_off[u + 1])].cpu().numpy()
    a = got.reshape(nfr, S, 2).astype(np.int32)
    b = np.asarray(by_ref).reshape(nfr, S, 2).astype(np.int32)
    ca, cb = a[..., 0] * 256 + a[..., 1], b[..., 0] * 256 + b[..., 1]
    out = {"utterance": u, "frames": nfr, "codes_equal_fraction": round(float((ca == cb).mean()), 6),
           "max_code_difference": int(np.abs(ca - cb).max()),
           "against": "oracle restatement of phone_probs (double), same audio and model"}
    # the same utterance with the engine in AASR_PREC_F64 (the reference's arithmetic in double): what
    # is left of the difference above when float rounding is taken away
    try:
        gmm.set_precision(1)
        data, n64 = A.run_utterance(runner.feat, gmm, runner.utts[u], lnabytes=2)
        c = np.frombuffer(data[5:], np.uint8).reshape(n64, S, 2).astype(np.int32)
        out["codes_equal_fraction_f64_mode"] = round(float(((c[..., 0] * 256 + c[..., 1]) == cb).mean()), 6)
    except Exception as e:
        out["codes_equal_fraction_f64_mode"] = "failed: %s" % e
    finally:
        gmm.set_precision(restore_precision)
    return out


def _time_scoring(torch, runner, gmm, reps=5):
    """ms of one scoring pass of `gmm` over the runner's resident feature frames (HIP events on the launch stream)."""
    def go():
        gmm.score_dev_pitched(runner.d_fea, runner.d_ll, runner.pitch, runner.stream)
    go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(runner.stream)
    for _ in range(reps):
        go()
    e1.record(runner.stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _time_engine_path(torch, runner, gmm, reps=5):
    """ms of frames -> 2-byte LNA codes through the engine's own score layout (aasr_gmm_score_lna_dev: what the recipe
    driver and aasr_run_utterance run after the feature chain) over the runner's resident feature frames."""
    F = runner.total_frames
    d_scr = torch.empty(gmm.score_scratch_floats(F), dtype=torch.float32, device=runner.d_fea.device)

    def go():
        gmm.score_lna_dev(runner.d_fea, d_scr, runner.d_bytes, True, 2, runner.stream)
    go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(runner.stream)
    for _ in range(reps):
        go()
    e1.record(runner.stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _measure_precisions(torch, capi, synth, gmm, runner, model, restore_precision, with_models=True):
    """The scoring stage of configs[2] in every arithmetic form the engine has (the any-model numbers next to the
    headline's), and per-state precision routing: the same model with 1 / 10 / 40 % of its states holding one Gaussian
    over the plain two-term layout's conditioning limits (kappa 250 / kappa2 80) -- those states become an engine part of
    their own in the slab-constant K layout (two fp16 terms, 6 slabs instead of 5), the rest keeps the plain layout."""
    import numpy as np
    out = {}
    ladder = {}
    for name, prec in (("f16x2", 4), ("bf16x3", 3), ("f32", 0)):
        gmm.set_precision(prec)
        ladder[name] = round(_time_scoring(torch, runner, gmm), 4)
    gmm.set_precision(restore_precision)
    base_engine = _time_engine_path(torch, runner, gmm)
    out["precision_ladder"] = {"what": "scoring stage of configs[2] (ms per %d frames x %d Gaussians) under each arithmetic: two fp16 "
                                       "terms (the default where a model's conditioning allows it), three bf16 terms (any model "
                                       "on the matrix path), plain f32 matrix instructions" % (runner.total_frames, G),
                               "scoring_ms": ladder}
    if not with_models:
        return out
    rng = np.random.default_rng(synth.SEED + 99)
    routing = []
    for share in (0.01, 0.10, 0.40):
        bad = sorted(rng.choice(S, max(1, int(round(share * S))), replace=False).tolist())
        g2 = capi.Gmm.from_arrays(*synth.push_states_over_the_f16_limits(model, bad))
        n16, moved = g2.precision_states()
        ms = _time_scoring(torch, runner, g2)
        ms_engine = _time_engine_path(torch, runner, g2)
        g2.set_precision(3)
        ms3 = _time_scoring(torch, runner, g2)
        g2.close()
        routing.append({"states_over_the_f16x2_limits": len(bad), "share": share, "states_f16x2": n16,
                        "states_moved_by_the_probe": moved, "scoring_ms": round(ms, 4),
                        "ratio_to_all_f16x2": round(ms / ladder["f16x2"], 4),
                        "target_1_plus_0.65_share": round(1.0 + 0.65 * share, 4),
                        "engine_path_scoring_plus_lna_ms": round(ms_engine, 4),
                        "engine_path_scoring_ms_minus_all_f16x2": round(ms_engine - base_engine, 4),
                        "engine_path_scoring_ratio": round((ladder["f16x2"] + ms_engine - base_engine) / ladder["f16x2"], 4),
                        "scoring_ms_whole_model_bf16x3": round(ms3, 4)})
    out["precision_routing"] = {"what": "per-state precision routing (aasr_gmm_precision_states): the configs[2] model with one Gaussian "
                                        "of a share of its states moved over the plain two-term layout's conditioning limits (to kappa2 = 100, "
                                        "kappa ~ 400); before round 4 ONE such Gaussian sent the whole model to the three-term kernel, "
                                        "rounds 4-5 scored those states with three bf16 terms, round 6 with two fp16 terms in the "
                                        "slab-constant K layout (an engine part of their own).  scoring_ms: the public score "
                                        "layout (columns = states: the parts' columns gathered back); engine_path_*: frames -> LNA codes on "
                                        "the engine's own layout (every part scores into its own column range, the LNA pass reads through a column map), "
                                        "what phone_probs / aasr_run_recipe run -- its ratio prices the scoring stage as all-f16x2 ms "
                                        "+ the extra ms of the whole path.  all-f16x2 engine path: %.4f ms" % base_engine,
                                "models": routing}
    return out


def _measure_fitted_models(torch, capi, synth, pipeline, gmm, runner, dev, precision):
    """The headline's robustness to the model's conditioning: 50 000 diagonal Gaussians FITTED (synth.fit_model: seeded
    tree clustering + mixture splitting, variances floored at 0.1 of the global variance like aku's estimate --minvar) to
    the engine's own 39-d features of one synthetic hour -- the stationary audio of the timed workload and a source-filter
    imitation of speech (silence / voiced / fricative segments: the modes trained models see) --, features brought to zero
    mean and unit variance per dimension as aku's normalization module does.  Such a model's Gaussians sit around their
    STATE's centre with variances down to the floor: around the pool's one pivot most of them break the two-term limits, so
    the engine sorts the states into pivot groups (engine parts).  Reported: the split of the states over the
    arithmetics, the pivot groups, ms of frames -> 2-byte LNA codes on the engine's own layout against the BASELINE
    model's on the same path, max |d ll| against the oracle on 64 frames (visible values)."""
    import time as _t
    from oracle import oracle as O
    base_ms = _time_engine_path(torch, runner, gmm)
    out = {"what": "synth.fit_model on %d frames of the engine's own features (360 distinct seeded 10-s utterances), "
                   "standardised per dimension; engine path = aasr_gmm_score_lna_dev, frames -> 2-byte LNA codes" % runner.total_frames,
           "baseline_model_engine_path_ms": round(base_ms, 4), "models": []}
    n_utts = len(runner.utts)
    sr = runner.feat.sample_rate
    n = len(runner.utts[0])
    for kind, mk in (("stationary", synth.make_audio), ("speechlike", synth.make_speechlike_audio)):
        t0 = _t.perf_counter()
        utts = [mk(n, seed=synth.SEED + 7000 + i, sample_rate=sr) for i in range(n_utts)]
        r2 = pipeline.FullChainBench(gmm, n_utts, n / sr, 0, dev, cfg_text=runner.cfg_text, utts=utts)
        r2.features_only()
        torch.cuda.synchronize()
        X = r2.d_fea.cpu().numpy()
        X = ((X - X.mean(0)) / X.std(0)).astype(np.float32)
        t_feat = _t.perf_counter() - t0
        t0 = _t.perf_counter()
        model = synth.fit_model(X, S=S, comps=COMPS)
        t_fit = _t.perf_counter() - t0
        t0 = _t.perf_counter()
        g2 = capi.Gmm.from_arrays(*model)
        g2.set_precision(precision)
        t_build = _t.perf_counter() - t0
        parts = g2.engine_parts()
        n16, moved = g2.precision_states()
        k1, k2 = synth.conditioning(model[0], model[1])
        r2.d_fea.copy_(torch.from_numpy(X))
        ms = _time_engine_path(torch, r2, g2)
        rng = np.random.default_rng(synth.SEED + 71)
        fi = np.sort(rng.choice(X.shape[0], 64, replace=False))
        sub = np.ascontiguousarray(X[fi])
        ref = O.DiagModel(*model).score(sub.astype(np.float64))
        got = g2.score(sub)
        vis = ref > -103.0
        err = np.abs(got - ref)
        best = ref.max(1, keepdims=True)
        win = vis & (ref > best - 36.0)
        # the same frames on the ENGINE's own layout (what the timed path runs: the parts' kernels): 4-byte LNA values
        # without normalisation are the state log-likelihoods as floats; compared where the reference's float likelihood
        # is a normal number (ll > ln 2^-126)
        eng = None
        try:
            d_sub = torch.from_numpy(sub).to(dev)
            d_scr = torch.empty(g2.score_scratch_floats(len(sub)), dtype=torch.float32, device=dev)
            d_b4 = torch.empty((len(sub), S * 4), dtype=torch.uint8, device=dev)
            g2.score_lna_dev(d_sub, d_scr, d_b4, False, 4, r2.stream)
            torch.cuda.synchronize()
            ll_eng = d_b4.cpu().numpy().view("<f4").reshape(len(sub), S).astype(np.float64)
            nrm = ref > -87.0
            e_eng = np.abs(ll_eng - ref)
            eng = {"max_abs_dll": float("%.3g" % e_eng[nrm].max()), "values": int(nrm.sum()),
                   "max_abs_dll_inside_the_2_byte_lna_window": float("%.3g" % e_eng[nrm & win].max()),
                   "share_within_1e-4": round(float((e_eng[nrm] <= 1e-4).mean()), 6)}
        except Exception as e:
            eng = {"error": "%s: %s" % (type(e).__name__, e)}
        entry = {"audio": kind, "states": S, "gaussians": G,
                 "one_pivot_conditioning": {"kappa_max": round(float(k1.max()), 1), "kappa2_max": round(float(k2.max()), 1),
                                            "states_within_the_f16x2_limits": int(((k1.reshape(S, COMPS).max(1) <= 250.0) &
                                                                                   (k2.reshape(S, COMPS).max(1) <= 80.0)).sum())},
                 "states_f16x2": n16, "share_f16x2": round(n16 / S, 4), "states_moved_by_the_probe": moved,
                 "engine_parts": parts, "engine_path_ms": round(ms, 4), "ratio_to_baseline_model": round(ms / base_ms, 4),
                 "max_abs_dll_vs_oracle_64_frames": float("%.3g" % err[vis].max()), "visible_values": int(vis.sum()),
                 "max_abs_dll_inside_the_2_byte_lna_window": float("%.3g" % err[win].max()),
                 "share_within_1e-4": round(float((err[vis] <= 1e-4).mean()), 6),
                 "public_layout_note": "aasr_gmm_score on these frames: the model's own one-pivot layouts where they cover it, "
                                       "the engine parts gathered back otherwise",
                 "engine_layout_vs_oracle_64_frames": eng,
                 "seconds": {"audio_and_features": round(t_feat, 1), "fit": round(t_fit, 1), "build": round(t_build, 2)}}
        # ... and the clustered pass on it (what pyrectool always runs: -C, --eval-ming 0.25): the parts' exact parts under
        # the same selection bits, merged through the column map
        try:
            C_ = 1000
            g2c = synth.make_clustering(model[0], C_, iters=2)
            g2.set_clustering(C_, [(i, int(c)) for i, c in enumerate(g2c)])
            g2.set_clustering_min_evals(0.0, 0.25)
            pitch = r2.d_ll.shape[1]

            def cstep():
                g2.score_dev_pitched(r2.d_fea, r2.d_ll, pitch, r2.stream)
            cstep()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(r2.stream)
            for _ in range(3):
                cstep()
            e1.record(r2.stream)
            torch.cuda.synchronize()
            cms = e0.elapsed_time(e1) / 3
            entry["clustered"] = {"clusters": C_, "eval_ming": 0.25, "ms_per_pass": round(cms, 3),
                                  "ms_per_million_frames": round(cms * 1e6 / X.shape[0], 3)}
        except Exception as e:
            entry["clustered"] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["models"].append(entry)
        g2.close()
        r2.release()
        del r2
        torch.cuda.empty_cache()
    return out


def _stage_rooflines(runner, split):
    """The two HBM-bound stages of configs[2] against the 8 TB/s HBM peak, in ALGORITHMIC bytes
    (SURVEY.md section 8d): feature chain 256 B of new samples in + 156 B of float features out per
    frame; LNA pass S * (4 in + B out) bytes per frame."""
    F = runner.total_frames
    out = []
    for name, kernels, bpf in (("features", "k_spectral_fused + k_temporal_fused + k_mean_subtract_tiled", 412),
                               ("lna", "k_state_norm_lna", runner.S * (4 + runner.lnabytes))):
        ms = split.get(name)
        if not ms:
            continue
        gbs = F * bpf / (ms * 1e-3) / 1e9
        e = {"stage": name, "kernels": kernels, "bound": "hbm", "algorithmic_bytes_per_frame": bpf,
             "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(gbs / HBM_PEAK_GBS, 4), "stage_ms": ms}
        if name == "features":
            v = _spectral_valu_roofline(F, ms)
            if v:
                e["valu_issue"] = v
        out.append(e)
    return out


def _spectral_valu_roofline(frames, stage_ms):
    """The ruler that fits k_spectral_fused (two thirds of the feature stage): it is bound by vector-instruction ISSUE, not by
    HBM -- the reference's arithmetic prescribes ~430 vector instructions per frame (float butterflies in KissFFT's order,
    serial float mel / power chains, double module buffers), a SIMD issues one vector instruction per 4 cycles, f32 or f64.
    Instruction count: the committed counter pass (profiles/*spectral_pmc*.txt, SQ_INSTS_VALU per launch of 449 280
    frames), scaled to this run's frames; the kernel's share of the stage: the committed kernel trace of the same tree
    (profiles/*full_chain*kernel_stats.csv); clock: nominal (the feature kernels do not reach the power cap)."""
    import csv
    import glob
    import re
    pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "*spectral_pmc*.txt")))
    ks = sorted(glob.glob(os.path.join(ROOT, "profiles", "*full_chain*kernel_stats.csv")))
    if not pm or not ks:
        return None
    m = re.search(r"SQ_INSTS_VALU\s+([0-9.e+]+)", open(pm[-1]).read())
    if not m:
        return None
    insts = float(m.group(1)) * frames / 449280.0
    t = {}
    for r in csv.DictReader(open(ks[-1])):
        for k in ("k_spectral_fused", "k_temporal_fused", "k_mean_subtract"):
            if k in r["Name"]:
                t[k] = float(r["AverageNs"]) * 1e-6
    if "k_spectral_fused" not in t or len(t) < 2:
        return None
    share = t["k_spectral_fused"] / sum(t.values())
    kernel_ms = stage_ms * share
    simds = 256 * 4
    slots_per_s = simds * NOMINAL_SCLK_MHZ * 1e6 / 4.0
    bound_ms = insts / slots_per_s * 1e3
    return {"kernel": "k_spectral_fused", "bound": "valu_issue", "vector_instructions_per_frame": round(insts / frames, 1),
            "issue_slots_per_s": slots_per_s, "bound_ms": round(bound_ms, 4), "kernel_ms": round(kernel_ms, 4),
            "kernel_share_of_stage": round(share, 3), "frac": round(bound_ms / kernel_ms, 4),
            "source": {"instructions": os.path.basename(pm[-1]), "kernel_share": os.path.basename(ks[-1])}}


def _measure_configs1(torch, synth, gmm, rank, dev, stream, sync_all, max_over_ranks, world, args, steps=5):
    """BASELINE configs[1] next to the headline: 1 000 000 resident frames x 50 000 Gaussians, scoring only."""
    F = args.frames
    gen = torch.Generator(device=dev)
    gen.manual_seed(synth.SEED + 17 * rank)
    d_frames = torch.randn((F, DIM), generator=gen, device=dev, dtype=torch.float32)
    pitch = (S + 31) // 32 * 32 if (args.out_pitch == "aligned" and gmm.score_pitch_ok()) else S
    d_out = torch.empty((F, pitch), device=dev, dtype=torch.float32)

    def step():
        if pitch == S:
            gmm.score_dev(d_frames, d_out, stream)
        else:
            gmm.score_dev_pitched(d_frames, d_out, pitch, stream)
    step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    ms = 1e3 * elapsed / steps
    flop = 4.0 * DIM * float(F) * float(gmm.expanded_rows)
    return {"workload": "configs[1]: batched diag-GMM log-likelihood only, %d x %d-d frames x %d Gaussians per rank, "
                        "output row pitch %d floats" % (F, DIM, G, pitch),
            "frames_per_gpu_per_step": F, "steps": steps, "ms_per_step": round(ms, 4),
            "frames_per_s": round(world * F * steps / elapsed, 1),
            "algorithmic_TFLOPs": round(flop / (ms * 1e-3) / 1e12, 2)}


def _measure_clustered(torch, capi, synth, model, dev, stream, frames, steps=3):
    """What production recognisers run (pyrectool passes -C ... --eval-ming 0.25, rectool.py:662-664): the configs[1]
    block scored with Gaussian clustering on -- 1000 clusters, int(0.25 G) Gaussians evaluated exactly per frame, the
    rest at their cluster centre's value (aku/Distributions.cc:2684-2722).  A fresh handle on rank 0."""
    C_ = 1000
    g2c = synth.make_clustering(model[0], C_, iters=2)
    g = capi.Gmm.from_arrays(*model)
    g.set_clustering(C_, [(i, int(c)) for i, c in enumerate(g2c)])
    g.set_clustering_min_evals(0.0, 0.25)
    F = frames
    gen = torch.Generator(device=dev)
    gen.manual_seed(synth.SEED + 23)
    d_frames = torch.randn((F, DIM), generator=gen, device=dev, dtype=torch.float32)
    pitch = (S + 31) // 32 * 32 if g.score_pitch_ok() else S
    d_out = torch.empty((F, pitch), device=dev, dtype=torch.float32)

    def step():
        if pitch == S:
            g.score_dev(d_frames, d_out, stream)
        else:
            g.score_dev_pitched(d_frames, d_out, pitch, stream)
    step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(steps):
        step()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    n_exact = g.cluster_exact_counts(min(F, 2000))
    g.close()
    return {"what": "configs[1] with Gaussian clustering: %d frames x %d Gaussians in %d clusters, --eval-minc 0 --eval-ming 0.25 "
                    "(centres, selection, masked scoring, merge), output row pitch %d floats" % (F, G, C_, pitch),
            "clusters": C_, "eval_ming": 0.25, "frames": F, "ms_per_pass": round(ms, 3),
            "ms_per_million_frames": round(ms * 1e6 / F, 3), "frames_per_s": round(F / ms * 1e3, 1),
            "clusters_evaluated_exactly_per_frame_mean": round(float(n_exact.mean()), 1)}


def _measure_configs4(torch, capi, synth, dev, stream, steps=5):
    """BASELINE configs[4] next to the headline (rank 0, no collective inside): 10 000 full-covariance Gaussians of 39
    dimensions (625 states x 16), 200 000 resident frames, state log-likelihoods out; the library's default arithmetic
    (two fp16 terms where the pool's conditioning allows it -- `effective_precision` says which rows ran)."""
    D4, G4, S4, F4 = 39, 10000, 625, 200000
    rng = np.random.default_rng(synth.SEED)
    mean = rng.standard_normal((G4, D4))
    a = rng.standard_normal((G4, D4, D4)) * 0.3
    cov = a @ a.transpose(0, 2, 1) + 0.1 * np.eye(D4)
    _, _, off, idx, w = synth.make_model(D=D4, G=G4, S=S4, comps=16)
    g4 = capi.Gmm.from_full(mean, cov, off, idx, w)
    try:
        d_fr = torch.randn((F4, D4), device=dev, dtype=torch.float32)
        d_out = torch.empty((F4, S4), device=dev, dtype=torch.float32)
        for _ in range(2):
            g4.score_dev(d_fr, d_out, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(steps):
            g4.score_dev(d_fr, d_out, stream)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        flop = float(D4 * (D4 + 3)) * F4 * G4          # d(d+3) flop per frame x Gaussian pair
        return {"workload": "configs[4]: %d full-covariance Gaussians x %d-d, %d states, %d resident frames, scoring only"
                            % (G4, D4, S4, F4),
                "frames_per_gpu_per_step": F4, "steps": steps, "ms_per_step": round(ms, 4),
                "frames_per_s": round(F4 / (ms * 1e-3), 1), "algorithmic_TFLOPs": round(flop / (ms * 1e-3) / 1e12, 2),
                "effective_precision": {0: "f32", 3: "bf16x3", 4: "f16x2"}.get(g4.effective_precision(), "?")}
    finally:
        g4.close()


def _measure_configs2(torch, capi, synth, gmm, rank, dev, stream, sync_all, max_over_ranks, world,
                      mean, var, off, idx, w, restore_precision=3):
    """BASELINE configs[2] next to a `--workload gmm` run: 360 x 10 s utterances per rank, MFCC chain +
    scoring + 2-byte LNA on the device; ms/step (max over ranks), per-stage split from HIP events,
    and on rank 0 the fraction of one utterance's LNA bytes that equal the oracle's."""
    from aaltoasr_amd import pipeline
    runner = pipeline.FullChainBench(gmm, n_utts=360, seconds=10.0, rank=rank, device=dev)
    steps = 5
    runner.step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.step()
    sync_all()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    split = runner.stage_split(steps)
    out = {"workload": "configs[2]: MFCC chain + %d-Gaussian scoring + 2-byte LNA, 360 x 10 s utterances per rank (%d frames), device resident" % (G, runner.total_frames),
           "frames_per_gpu_per_step": runner.total_frames, "steps": steps,
           "ms_per_step": round(1e3 * elapsed / steps, 4),
           "frames_per_s": round(world * runner.total_frames * steps / elapsed, 1),
           "stage_ms": split,
           "hbm_bytes_written_per_frame": runner.bytes_written_per_frame()}
    if rank == 0:
        out["lna_check"] = _lna_check(runner, gmm, mean, var, off, idx, w, restore_precision)
    return out


def _observe_clock(enqueue, torch):
    """Queues ~1.5 s of scoring launches and reads rocm-smi once while they run."""
    import re
    try:
        torch.cuda.synchronize()
        enqueue()
        time.sleep(0.4)
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True,
                             timeout=20).stdout
        torch.cuda.synchronize()
        res = {}
        m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
        if m:
            res["sclk_mhz"] = int(m.group(1))
        m = re.search(r"Package Power \(W\): ([0-9.]+)", out)
        if m:
            res["power_w"] = float(m.group(1))
        return res or None
    except Exception:
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        return None


def _pipe_probe(run=True):
    """The matrix pipe's own rate under the board's power cap: tools/mfma_peak (back-to-back
    v_mfma_f32_32x32x16_bf16 on register operands, no memory traffic; built by build() as
    aaltoasr_amd/lib/bin/mfma_peak) run on THIS box right after the timed region -- all-zero operands reach the
    datasheet rate, operands shaped like the three terms of a split do not, and the boxes of a pool differ by a few
    per cent.  Falls back to the figures recorded in profiles/r2_mfma_peak_probe.txt."""
    here = os.path.dirname(os.path.abspath(__file__))
    text, source = None, None
    exe = os.path.join(here, "aaltoasr_amd", "lib", "bin", "mfma_peak")
    if run and os.access(exe, os.X_OK):
        try:
            r = subprocess.run([exe, "40"], capture_output=True, text=True, timeout=120,
                               env=dict(os.environ, MFMA_PEAK_CLASSES="1"))
            if r.returncode == 0:
                text, source = r.stdout, "tools/mfma_peak run on this box after the timed region (40 ms launches)"
        except (OSError, subprocess.SubprocessError):
            pass
    if text is None:
        try:
            text = open(os.path.join(here, "profiles", "r2_mfma_peak_probe.txt")).read()
            source = "profiles/r2_mfma_peak_probe.txt (recorded on another box)"
        except OSError:
            return None
    rates = {}
    for line in text.splitlines():
        for key, tag in (("all-zero operands", "zeros"), ("random mantissas and exponents", "random"),
                         ("split-term classes, mixed order", "split_terms")):
            if line.startswith(key) and tag not in rates:
                rates[tag] = float(line[len(key):].split()[0])
    if "zeros" not in rates or ("split_terms" not in rates and "random" not in rates):
        return None
    out = {"TFLOPs_zero_operands": rates["zeros"], "TFLOPs_random_operands": rates.get("random"),
           "TFLOPs_split_term_operands": rates.get("split_terms"), "source": source}
    out["reference"] = "split_term_operands" if "split_terms" in rates else "random_operands"
    return out


def _pmc_traffic(frames, precision="f32", aligned=False):
    """HBM bytes per launch of k_gmm_diag_score from the committed rocprofv3 PMC
    passes (profiles/*_pmc*.json: FETCH_SIZE and WRITE_SIZE in KiB, collected in
    separate passes at 1 000 000 frames/launch; FETCH_SIZE doubled per
    MI355X_MICROARCH.md's gfx950 note), scaled to this launch's frame count.
    None when no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*gmm_%s_pmc*.json" % precision)))
    if not files and precision == "f32":
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*gmm_pmc*.json")))
    # passes taken with whole-line output rows carry "aligned" in the name
    picked = [f for f in files if ("aligned" in os.path.basename(f)) == aligned]
    files = picked or ([] if aligned else files)
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        fetch = d["FETCH_SIZE"]["mean_per_launch"] * 1024.0 * 2.0
        write = d["WRITE_SIZE"]["mean_per_launch"] * 1024.0
        base = float(d.get("frames_per_launch", 1_000_000))
        return {"bytes": round((fetch + write) * frames / base), "fetch_bytes": round(fetch * frames / base),
                "write_bytes": round(write * frames / base), "source": os.path.basename(files[-1]),
                "algorithmic_bytes": int(frames * (DIM * 4 + S * 4) + G * (2 * DIM + 1) * 4)}
    except Exception:
        return None


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
