"""Library hygiene: the product library reads only the environment variables INTEGRATION.md documents.  Experiment
switches (kernel variants, layouts, arithmetic A/B) live behind AASR_EXPERIMENT_ENV (csrc/common.h) and exist only in an
ablation build, so their names must not even appear in the product objects' strings."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCUMENTED = {"AASR_PREC", "AASR_PCGMM_AS_WRITTEN", "AASR_WRITER_THREADS", "AASR_RECIPE_TIMING", "AASR_LOCAL_RANKS",
              "AASR_F16_PROBE_TOL", "AASR_PG_PIVOT_COST", "AASR_PG_PIVOT_COST3"}
# names of the public header that show up in messages, not environment variables
NOT_ENV = re.compile(r"^AASR_(PREC_[A-Z0-9_]+|ERR_[A-Z_]+|OK)$")


def _names(path):
    out = subprocess.run(["strings", "-a", path], capture_output=True, text=True, check=True).stdout
    return set(re.findall(r"AASR_[A-Z0-9_]+", out))


def test_product_library_reads_documented_variables_only():
    from aaltoasr_amd import build
    lib = build.build()
    found = {n for n in _names(lib) if not NOT_ENV.match(n)}
    assert found <= DOCUMENTED, "undocumented AASR_ names in libaasr.so: %s" % sorted(found - DOCUMENTED)
    objdir = os.path.join(os.path.dirname(lib), "obj")
    for f in sorted(os.listdir(objdir)):
        if f.endswith(".o"):
            extra = {n for n in _names(os.path.join(objdir, f)) if not NOT_ENV.match(n)} - DOCUMENTED
            assert not extra, (f, sorted(extra))


def test_every_documented_variable_is_in_integration_md():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in sorted(DOCUMENTED):
        assert name in text, name


def test_sources_use_getenv_for_documented_variables_only():
    csrc = os.path.join(ROOT, "aaltoasr_amd", "csrc")
    bad = []
    for root, _d, files in os.walk(csrc):
        for f in files:
            if not f.endswith((".cc", ".hip", ".h", ".hh")) or f == "common.h":
                continue
            for m in re.finditer(r'(?<![A-Za-z_])getenv\("([A-Z0-9_]+)"\)', open(os.path.join(root, f)).read()):
                if m.group(1) not in DOCUMENTED:
                    bad.append((f, m.group(1)))
    assert not bad, bad
