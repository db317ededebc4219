#!/usr/bin/env python
"""kernel_notes.py -- register / scratch / LDS figures of the kernels in a compiled object, read from the code object's
notes (no GPU needed):

    python tools/kernel_notes.py [pattern] [object]      default object: aaltoasr_amd/lib/obj/gmm_score.hip.o

Prints one line per kernel whose demangled name contains `pattern`."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(obj, workdir):
    """The gfx950 code object embedded in a hipcc object file."""
    tmp = os.path.join(workdir, "x.o")
    with open(obj, "rb") as f, open(tmp, "wb") as g:
        g.write(f.read())
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp], check=True, capture_output=True)
    for n in os.listdir(workdir):
        if "amdgcn" in n:
            return os.path.join(workdir, n)
    raise RuntimeError("no device code object in " + obj)


def kernel_notes(obj):
    """{demangled kernel name: {vgpr, agpr, sgpr, spill_vgpr, spill_sgpr, scratch, lds}} of every kernel in `obj`."""
    with tempfile.TemporaryDirectory() as d:
        co = code_object(obj, d)
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                             text=True).stdout
    out = {}
    for blk in txt.split("- .agpr_count:")[1:]:
        def num(key):
            m = re.search(r"\.%s:\s+(\d+)" % key, blk)
            return int(m.group(1)) if m else 0
        agpr = int(re.match(r"\s*(\d+)", blk).group(1))
        m = re.search(r"\.name:\s+(\S+)", blk)
        if not m:
            continue
        out[m.group(1)] = dict(vgpr=num("vgpr_count"), agpr=agpr, sgpr=num("sgpr_count"),
                               spill_vgpr=num("vgpr_spill_count"), spill_sgpr=num("sgpr_spill_count"),
                               scratch=num("private_segment_fixed_size"), lds=num("group_segment_fixed_size"))
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return {re.sub(r"^void ", "", d).split("(")[0]: out[n] for n, d in zip(names, dem)}


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    obj = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "aaltoasr_amd", "lib", "obj", "gmm_score.hip.o")
    for name, k in sorted(kernel_notes(obj).items()):
        if pat in name:
            print("%-70s vgpr %3d agpr %3d sgpr %3d spill v %3d s %3d scratch %4d B" % (
                name[-70:], k["vgpr"], k["agpr"], k["sgpr"], k["spill_vgpr"], k["spill_sgpr"], k["scratch"]))
