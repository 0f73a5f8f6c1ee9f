#!/usr/bin/env python3
"""Host time to enqueue one ststhip_strip_advance (1000 generations) against the time the GPU needs for it: the strip
driver must stay ahead of the device.  STSTHIP_STRIP_DEBUG_BANDS=1 adds the band launches of a rank with neighbours."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stencilstream_amd import capi

capi.init(0)
p = capi.JacobiParams()
for i in range(5):
    p.coef[i] = 0.2
for rows in (1024, 2048, 4096, 8192):
    strip = capi.Strip("jacobi5general", p, np.float32(0).tobytes(), rows, 16384, 0, 1)
    strip.upload(0, np.random.default_rng(1).random((rows, 16384), dtype=np.float32))
    strip.advance(0, 1000, blocking=True)
    enq, tot = [], []
    for _ in range(4):
        t0 = time.perf_counter()
        strip.advance(0, 1000)
        t1 = time.perf_counter()
        strip.synchronize()
        t2 = time.perf_counter()
        enq.append(t1 - t0)
        tot.append(t2 - t0)
    launches, _ = strip.counters()
    print(f"rows {rows}: enqueue {min(enq) * 1e3:7.2f} ms, until done {min(tot) * 1e3:7.2f} ms per 1000 generations "
          f"({launches // 5} launches per call)", flush=True)
    strip.close()
