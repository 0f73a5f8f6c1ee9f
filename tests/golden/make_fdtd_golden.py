#!/usr/bin/env python3
"""Regenerates tests/golden/fdtd/*.csv.

What produced them: build/examples/fdtd_cpu, i.e. the reference's UNCHANGED examples/fdtd sources
(Kernel.hpp transition function, Parameters.hpp, CoefResolver, fdtd.cpp harness; read where they lie
under /root/reference, see examples/Makefile) compiled with g++ against this repository's
stencil::cpu backend.  The arithmetic of the transition function and of the set-up is therefore the
reference's own source text; only the sweep driver underneath is this repository's (and that driver
is pinned separately by the reference's self-checking known-answer test).  Only runs where
/root/reference exists; the CSV files are committed so that the tests need neither."""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXPERIMENT = json.load(open(os.path.join(HERE, "fdtd", "experiment.json")))


def main():
    binary = os.path.join(ROOT, "build", "examples", "fdtd_cpu")
    if not os.path.exists(binary):
        sys.exit("build/examples/fdtd_cpu is missing: make -C examples (needs /root/reference)")
    out = os.path.join(HERE, "fdtd")
    with tempfile.TemporaryDirectory() as tmp:
        cfg = os.path.join(tmp, "experiment.json")
        json.dump(EXPERIMENT, open(cfg, "w"))
        res = subprocess.run([binary, "-c", cfg, "-o", tmp], check=True, capture_output=True, text=True)
        open(os.path.join(out, "stdout.txt"), "w").write(
            "\n".join(l for l in res.stdout.splitlines() if not l.startswith("Walltime")) + "\n")
        for f in sorted(os.listdir(tmp)):
            if f.endswith(".csv"):
                open(os.path.join(out, f), "w").write(open(os.path.join(tmp, f)).read())
                print("wrote", f)


if __name__ == "__main__":
    main()
