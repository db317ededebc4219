"""build.py -- compiles libaasr.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m aaltoasr_amd.build [--force]

hipcc cross-compiles without a GPU.  The shared library lands in
aaltoasr_amd/lib/ (git-ignored, travels to the GPU box with the snapshot).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# AASR_BUILD_ABLATION=1 builds the library with the kernels' ablation branches and the recipe driver's device
# stub into its own directory (lib_ablation/, loaded with AASR_LIBDIR=...): the product library carries neither
LIBDIR = os.path.join(HERE, os.environ.get("AASR_BUILD_LIBDIR") or
                      ("lib_ablation" if (os.environ.get("AASR_BUILD_ABLATION") == "1" or os.environ.get("AASR_BUILD_DEFINES"))
                       else "lib"))   # AASR_BUILD_LIBDIR: the directory of an experiment build (several side by side)
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libaasr.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# -ffp-contract=off: the feature kernels restate float32/float64 arithmetic of
# the reference operation by operation; fused multiply-adds would change bits.
ABLATION = ["-DAASR_ABLATION=1"] if os.environ.get("AASR_BUILD_ABLATION") == "1" else []
if os.environ.get("AASR_BUILD_DEFINES"):   # experiment builds: extra -D flags, into the ablation directory
    ABLATION = ["-D" + d for d in os.environ["AASR_BUILD_DEFINES"].split(",")]
COMMON = ABLATION + ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall",
          "-Wno-unused-function", "-Wno-inline-asm", f"--offload-arch={ARCH}", "-I", os.path.join(HERE, "..", "include")]


def _sources():
    out = []
    for root, _dirs, files in os.walk(CSRC):
        for f in sorted(files):
            if f.endswith((".hip", ".cc")) and not f.startswith("main_"):
                out.append(os.path.join(root, f))
    return sorted(out)


def _headers_digest() -> str:
    h = hashlib.sha1()
    paths = [os.path.join(HERE, "..", "include", "aasr.h")]
    for root, _dirs, files in os.walk(CSRC):
        for f in sorted(files):
            if f.endswith(".h") or f.endswith(".hh"):
                paths.append(os.path.join(root, f))
    for p in sorted(paths):
        h.update(open(p, "rb").read())
    # flags without the checkout path: a snapshot of the tree elsewhere (the GPU box) must find its
    # objects up to date instead of recompiling everything
    h.update(" ".join(c for c in COMMON if not os.path.isabs(c)).encode())
    return h.hexdigest()


def _compile(src: str, digest: str, force: bool) -> str:
    rel = os.path.relpath(src, CSRC).replace(os.sep, "_")
    obj = os.path.join(OBJDIR, rel + ".o")
    stamp = obj + ".stamp"
    key = digest + hashlib.sha1(open(src, "rb").read()).hexdigest()
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == key:
        return obj
    cmd = [HIPCC] + COMMON + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    open(stamp, "w").write(key)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = _sources()
    digest = _headers_digest()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, digest, force), srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + \
              ["-Wl,-rpath,/opt/rocm/lib", "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    # command-line tools on top of the library
    bindir = os.path.join(LIBDIR, "bin")
    os.makedirs(bindir, exist_ok=True)
    for name, src in (("phone_probs", "aku/main_phone_probs.cc"),
                      ("aku_adapter_check", "aku/main_adapter_check.cc"),
                      ("feacat", "aku/main_feacat.cc"),
                      ("plugin_check", "aku/main_plugin_check.cc"),
                      ("acoustics_check", "decoder/main_acoustics_check.cc")):
        srcp = os.path.join(CSRC, src)
        exe = os.path.join(bindir, name)
        if force or not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(srcp), os.path.getmtime(LIB)):
            cmd = [HIPCC, "-O2", "-std=c++17", srcp, "-o", exe, "-L", LIBDIR, "-laasr",
                   "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("building %s failed:\n%s\n%s" % (name, r.stdout, r.stderr))
    # the stand-alone matrix-pipe probe (tools/mfma_peak): a measuring instrument for bench.py, no part of the library
    probe_src = os.path.join(os.path.dirname(HERE), "tools", "mfma_peak", "mfma_peak.hip")
    probe = os.path.join(bindir, "mfma_peak")
    if os.path.exists(probe_src) and (force or not os.path.exists(probe) or os.path.getmtime(probe) < os.path.getmtime(probe_src)):
        r = subprocess.run([HIPCC, f"--offload-arch={ARCH}", "-O3", "-o", probe, probe_src], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building mfma_peak failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
