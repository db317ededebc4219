#!/usr/bin/env python
"""bench.py -- frames/sec of the acoustic-likelihood hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload gmm|full]

A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM:
  gmm  (BASELINE configs[1]): 1 000 000 x 39 float32 frames against a
        50 000-Gaussian / 3 125-state x 16 diagonal HmmSet -> [F x S] state
        log-likelihoods (k_gmm_diag_score).
  full (BASELINE configs[2]): 1 h of 16 kHz int16 audio as 360 x 10 s
        utterances -> MFCC chain -> same scoring -> 2-byte LNA codes.
Multi-GPU (launched by torch.distributed.run, one rank per GPU): every rank
scores its own shard of frames/utterances, no collective on the scoring path;
value = total frames / max-over-ranks time ("weak" scaling).

Prints ONE JSON line on rank 0.  The roofline block prices the dominant
kernel (k_gmm_diag_score) in ALGORITHMIC flops: 4*dim = 156 flop per
frame x Gaussian pair against the dense FP32 matrix
peak of 157.3 TFLOP/s (MI355X_MICROARCH.md).  The cpu_baseline block times
the oracle's reference-shaped scalar double loop on this host.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3
BF16_MATRIX_PEAK_TFLOPS = 2500.0
DIM = 39
G = 50000
S = 3125
COMPS = 16
NOMINAL_SCLK_MHZ = 2400.0   # the clock the datasheet peaks are quoted at


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["gmm", "full"], default=os.environ.get("AASR_BENCH_WORKLOAD", "gmm"))
    ap.add_argument("--frames", type=int, default=1_000_000, help="frames per GPU per step (gmm workload)")
    ap.add_argument("--utts", type=int, default=360, help="10-s utterances per GPU per step (full workload)")
    ap.add_argument("--out-pitch", choices=["dense", "aligned"], default=os.environ.get("AASR_BENCH_OUT_PITCH", "aligned"),
                    help="gmm workload: output rows dense [F x S] or padded to whole 64-byte lines")
    ap.add_argument("--cpu-frames", type=int, default=20000, help="frames timed on the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="processes of the all-cores CPU baseline (-1 = one per usable core, 0 = skip)")
    ap.add_argument("--cpu-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--precision", choices=["bf16x3", "f32"], default=os.environ.get("AASR_BENCH_PRECISION", "bf16x3"),
                    help="contraction arithmetic of the scoring kernel (both meet the 1e-4 parity bar)")
    return ap.parse_args()


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
    import subprocess
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


def main():
    args = parse_args()
    if args.cpu_worker > 0:
        cpu_worker(args.cpu_worker)
        return
    # stdout carries exactly one JSON line: anything native libraries print there (RCCL's version
    # banner, rocm-smi) is sent to stderr for the life of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # AASR_BENCH_FORCE_DIST=1 takes the multi-rank code path (RCCL init, model broadcast, barrier,
    # max-over-ranks) at world size 1 too, so it can be exercised on a one-GPU box under torchrun
    distributed = world > 1 or os.environ.get("AASR_BENCH_FORCE_DIST") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from aaltoasr_amd import build, capi, synth
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
        from aaltoasr_amd import shard
        model = shard.broadcast_model(model, src=0, device=dev)
    mean, var, off, idx, w = (model[k] for k in names)
    gmm = capi.Gmm.from_arrays(mean, var, off, idx, w)
    gmm.set_precision(3 if args.precision == "bf16x3" else 0)   # AASR_PREC_BF16X3 / AASR_PREC_F32
    rows = gmm.expanded_rows

    stream = torch.cuda.current_stream()
    feat = None
    if args.workload == "gmm":
        F = args.frames
        gen = torch.Generator(device=dev)
        gen.manual_seed(synth.SEED + 17 * rank)
        d_frames = torch.randn((F, DIM), generator=gen, device=dev, dtype=torch.float32)
        # output rows either dense (pitch S, what aasr_gmm_score hands to a host caller) or padded to
        # whole 128-byte lines (pitch S rounded up to 32 floats: the layout the device-resident chain
        # aasr_gmm_score_dev_pitched -> aasr_lna_encode_dev_pitched uses)
        pitch = (S + 31) // 32 * 32 if (args.out_pitch == "aligned" and gmm.score_pitch_ok()) else S
        d_out = torch.empty((F, pitch), device=dev, dtype=torch.float32)
        workload = "configs[1]: batched diag-GMM log-likelihood, %d x %d-d frames x %d Gaussians (%d states x %d), output row pitch %d floats" % (
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
        runner = pipeline.FullChainBench(gmm, n_utts=args.utts, seconds=10.0, rank=rank, device=dev)
        F = runner.total_frames
        workload = "configs[2]: MFCC chain + %d-Gaussian scoring + 2-byte LNA, %d x 10 s synthetic 16 kHz utterances (%d frames)" % (
            G, args.utts, F)
        step = runner.step
        score_only = runner.score_only

    def sync_all():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * F * args.steps / elapsed

    # ---- dominant-kernel time: HIP events on the launch stream, scoring only
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    kreps = max(1, args.steps)
    torch.cuda.synchronize()
    ev0.record(stream)
    for _ in range(kreps):
        score_only()
    ev1.record(stream)
    torch.cuda.synchronize()
    k_ms = ev0.elapsed_time(ev1) / kreps
    algo_flop = 4.0 * DIM * float(F) * float(rows)
    achieved = algo_flop / (k_ms * 1e-3) / 1e12
    if args.precision == "f32":
        kernel, peak, dtype = "k_gmm_diag_score_tracks<40,true>", FP32_MATRIX_PEAK_TFLOPS, "f32"
        peak_note = "dense FP32 matrix peak (v_mfma_f32_32x32x2_f32), MI355X_MICROARCH.md"
    else:
        # every f32-accurate product is six bf16 matrix products (3-term split of
        # both operands, terms below 2^-16 dropped), so the ceiling for ALGORITHMIC
        # flops on the bf16 pipe is the dense bf16 peak / 6
        # f32-accurate arithmetic: every f32 operand is carried as three bf16 terms (24 significant
        # bits), six bf16 matrix-core products per f32 product, f32 accumulation; it meets the same
        # 1e-4 parity bar as the plain f32 kernel (--precision f32)
        kernel, peak, dtype = ("k_gmm_diag_score_bf16x3<5,true>", BF16_MATRIX_PEAK_TFLOPS / 6.0,
                               "f32 (3-term bf16 split on the matrix cores, f32 accumulate)")
        peak_note = ("dense BF16 matrix peak 2500 TFLOP/s / 6 bf16 products per f32-accurate product; "
                     "executed matrix flops = 6 * 160/156 * achieved")
    roofline = {
        "bound": "mfma", "kernel": kernel, "achieved": round(achieved, 3),
        "peak": round(peak, 2), "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": None,
        "kernel_ms": round(k_ms, 4), "algorithmic_flop_per_launch": algo_flop, "peak_note": peak_note,
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
    td = _pmc_traffic(F, args.precision, aligned=(args.workload == "full" or args.out_pitch == "aligned"))
    if td:
        roofline["traffic"] = td["bytes"]
        roofline["traffic_detail"] = td

    # ---- CPU baseline (rank 0, N=1 only): oracle's reference-shaped loop
    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        from oracle import oracle as O
        O.build()
        om = O.DiagModel(mean, var, off, idx, w)
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

    if rank == 0:
        line = {
            "metric": "frames/sec GMM log-lik (39-d, 50k Gauss) + MFCC",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": workload, "frames_per_gpu_per_step": F, "dim": DIM, "gaussians": G,
                       "states": S, "components_per_state": COMPS, "sharding": "frames/utterances per rank, no collective"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def _observe_clock(enqueue, torch):
    """Queues ~1.5 s of scoring launches and reads rocm-smi once while they run."""
    import re
    import subprocess
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
