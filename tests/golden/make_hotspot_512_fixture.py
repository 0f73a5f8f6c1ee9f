#!/usr/bin/env python3
"""Second reference-compiled pin for HotSpot.

Inputs: the reference's own data files examples/hotspot/data/{temp,power}_512 (Rodinia's 512 x 512 chip), stored here
gzip-compressed as `hotspot_temp_512.txt.gz` / `hotspot_power_512.txt.gz` (data, one float per line).
Expected output: what Rodinia's OpenMP HotSpot -- examples/hotspot/hotspot_openmp.cpp of the reference tree, compiled
UNCHANGED from where it lies by `make -C oracle ref` into oracle/_ref/hotspot_openmp -- writes for them after 2000
iterations, parsed into float32 and stored as `hotspot_512_rodinia_2000.npz`.  It prints six significant digits
(`%g`), so the fixture carries that rounding: the test's tolerance says so.

Run in the build container (where /root/reference exists):  python3 tests/golden/make_hotspot_512_fixture.py"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DATA = "/root/reference/examples/hotspot/data"
ITERATIONS = 2000


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "hotspot_openmp")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    for name in ("temp", "power"):
        with open(os.path.join(DATA, f"{name}_512"), "rb") as src, \
                gzip.GzipFile(os.path.join(HERE, f"hotspot_{name}_512.txt.gz"), "wb", compresslevel=9, mtime=0) as dst:
            shutil.copyfileobj(src, dst)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "out.txt")
        subprocess.check_call([exe, "512", "512", str(ITERATIONS), "4", os.path.join(DATA, "temp_512"),
                               os.path.join(DATA, "power_512"), out], stdout=subprocess.DEVNULL)
        temp = np.loadtxt(out, dtype=np.float64)[:, 1].reshape(512, 512).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "hotspot_512_rodinia_2000.npz"), temp=temp, iterations=ITERATIONS)
    print("wrote", temp.shape, float(temp.min()), float(temp.max()))


if __name__ == "__main__":
    sys.exit(main())
