"""Command-line grammar of the tools (aaltoasr_amd/csrc/aku/conf.hh) against the reference's own
parser, aku/conf.cc compiled in place (oracle/_ref/conf_ref): the same driver source is built on
both, run on the same command lines, and stdout / stderr / exit status must be identical.  The
option tables are phone_probs' and feacat's (aku/phone_probs.cc:60-82, aku/feacat.cc:50-63)."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(HERE, "oracle", "_ref", "conf_ref")
ENG = os.path.join(HERE, "oracle", "conf_engine")

PHONE_PROBS = ["usage: phone_probs [OPTION...]",
               ("h", "help", "", "", "display help", "s"),
               ("b", "base=BASENAME", "arg", "", "base filename for model files", "s"),
               ("g", "gk=FILE", "arg", "", "Gaussian kernels", "s"),
               ("m", "mc=FILE", "arg", "", "kernel indices for states", "s"),
               ("p", "ph=FILE", "arg", "", "HMM definitions", "s"),
               ("c", "config=FILE", "arg must", "", "feature configuration", "s"),
               ("r", "recipe=FILE", "arg must", "", "recipe file", "s"),
               ("o", "output-dir=DIR", "arg", "", "output directory (default: use filenames from recipe)", "s"),
               ("0", "lnabytes=INT", "arg", "2", "number of bytes for probabilities, 2 (default) or 4", "i"),
               ("a", "afname", "", "", "use audio file name", "s"),
               ("n", "no-overwrite", "", "", "prevent overwriting existing files", "s"),
               ("S", "speakers=FILE", "arg", "", "speaker configuration file", "s"),
               ("C", "clusters=FILE", "arg", "", "Gaussian clustering file", "s"),
               ("0", "eval-minc=FLOAT", "arg", "0", "minimum ratio of top clusters to evaluate", "d"),
               ("0", "eval-ming=FLOAT", "arg", "0.1", "minimum ratio of Gaussians to evaluate", "d"),
               ("0", "sort-recipe", "", "", "sort recipe lines, useful with adaptation", "s"),
               ("N", "no-normalization", "", "", "do not normalize the likelihoods", "s"),
               ("B", "batch=INT", "arg", "0", "number of batch processes with the same recipe", "i"),
               ("I", "bindex=INT", "arg", "0", "batch process index", "i"),
               ("i", "info=INT", "arg", "0", "info level", "i")]

FEACAT = ["usage: feacat [OPTION...] FILE",
          ("h", "help", "", "", "display help", "s"),
          ("c", "config=FILE", "arg must", "", "read feature configuration", "s"),
          ("w", "write-config=FILE", "arg", "", "write feature configuration", "s"),
          ("0", "raw-output", "", "", "raw float output", "s"),
          ("H", "header", "", "", "write a header (feature dim, 32 bits) in raw output", "s"),
          ("s", "start-frame=INT", "arg", "", "audio start frame", "i"),
          ("e", "end-frame=INT", "arg", "", "audio end frame", "i"),
          ("S", "speakers=FILE", "arg", "", "speaker configuration file", "s"),
          ("d", "speaker-id=NAME", "arg", "", "speaker ID", "s"),
          ("u", "utterance-id=NAME", "arg", "", "utterance ID", "s"),
          ("G", "gaussian-std=FLOAT", "arg", "", "Gaussian noise std added to features", "f")]


def _spec(tmp_path, table, name):
    p = str(tmp_path / name)
    with open(p, "w") as f:
        f.write(table[0] + "\n")
        for row in table[1:]:
            f.write("\t".join(row) + "\n")
    return p


def _run(exe, spec, words):
    r = subprocess.run([exe, spec] + words, capture_output=True, timeout=30)
    return r.returncode, r.stdout, r.stderr


CASES = [
    "-b m -c f.cfg -r x.recipe",
    "--base=m --config=f.cfg --recipe x.recipe --lnabytes=4 -aN",
    "-aNn -c f -r r",
    "-cr f.cfg x.recipe -b m",                       # grouped options, arguments in order
    "-hic 10 str",                                   # help wins after parsing
    "-i10 -c f -r r",                                # NOT info=10: three short options
    "-i -3 -c f -r r",                               # an option's argument may start with '-'
    "-c -r -r x",                                    # "-r" is the value of -c
    "-c f -r r -- -b m",                             # "--" ends option processing
    "-c f -r r - extra",                             # a lone hyphen is an argument
    "-c f -r r --afname=1",                          # value handed to an option without argument
    "--conf f -r r",                                 # no abbreviations
    "-c f -r r -x",                                  # unknown short option
    "-c f -r r --lnabytes",                          # missing argument at the end
    "-c f",                                          # required option missing
    "-r r",
    "",
    "-h",
    "--help",
    "-c f -r r -i 3x",                               # typed getter rejects trailing text
    "-c f -r r --eval-ming=0.25 --eval-minc 1e-2 -B 4 -I 2 --sort-recipe",
    "-c f -r r --eval-ming=abc",
    "-c a -c b -r r",                                # the later value wins
    "-c f -r r -i ''",
    "-c f -r r --info= ",
    "-c=f -r r",                                     # '=' is not special for short options
]


@pytest.mark.parametrize("table,name", [(PHONE_PROBS, "pp.spec"), (FEACAT, "fc.spec")])
def test_cli_grammar_matches_reference_parser(oracle, tmp_path, table, name):   # `oracle`: runs make -C oracle
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/conf_ref not built (no reference tree)")
    assert os.path.exists(ENG), "oracle/conf_engine missing: run `make -C oracle`"
    spec = _spec(tmp_path, table, name)
    for case in CASES:
        words = [w if w != "''" else "" for w in case.split(" ")] if case else []
        assert _run(ENG, spec, words) == _run(REF, spec, words), case
    # random command lines over the table's own vocabulary
    rng = np.random.default_rng(17)
    shorts = [r[0] for r in table[1:] if r[0] != "0"]
    longs = [r[1].split("=")[0] for r in table[1:]]
    vocab = (["-" + s for s in shorts] + ["--" + l for l in longs] + ["--" + l + "=v" for l in longs] +
             ["-" + a + b for a in shorts[:6] for b in shorts[3:9]] +
             ["x", "7", "-", "--", "-5", "0.5", "--bogus", "-Z", "file.wav", "--=", "-=", "--config=", "2"])
    agree = 0
    for _ in range(600):
        words = [str(w) for w in rng.choice(vocab, size=int(rng.integers(0, 9)))]
        a, b = _run(ENG, spec, words), _run(REF, spec, words)
        assert a == b, words
        agree += a[0] == 0
    assert agree > 20      # the sweep reaches successful parses too
