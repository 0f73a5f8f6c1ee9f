#!/usr/bin/env python3
"""What ONE rank of a multi-GPU run does per step, measured on one GPU: the middle strip of three (neighbours on both
sides: boundary bands, widened launches, the exchange's place in the stream order) driven by the native strip driver
with an exchange callback that moves nothing.  The rows next to the strip's ends are therefore wrong -- this is a
timing study of the compute side of a rank; tests/test_strip_native_gpu.py holds the correctness of the same driver
with real exchanges.  Gcell-updates/s per GPU = owned rows x width x generations / time.

usage: tools/bench_strip.py [--rows 2048 4096 8192] [--width 16384] [--generations 1000] [--exchange-every 1 2 4]
                            [--app jacobi5general] [--general]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stencilstream_amd import capi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[2048, 4096, 8192])
    ap.add_argument("--width", type=int, default=16384)
    ap.add_argument("--generations", type=int, default=1000)
    ap.add_argument("--exchange-every", type=int, nargs="+", default=[1])
    ap.add_argument("--general", action="store_true", help="coefficients that differ (the 9-flop kernel)")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    capi.init(0)
    p = capi.JacobiParams()
    for i, c in enumerate([0.2, 0.21, 0.19, 0.22, 0.18] if args.general else [0.2] * 5):
        p.coef[i] = c
    halo = np.float32(0).tobytes()
    # an exchange that moves nothing and costs nothing on the host: libc's sched_yield has a compatible C signature
    # for this purpose (it ignores its arguments and returns 0)
    import ctypes
    no_exchange = ctypes.cast(ctypes.CDLL(None).sched_yield, ctypes.c_void_p).value

    for rows in args.rows:
        for m in args.exchange_every:
            os.environ["STSTHIP_EXCHANGE_EVERY"] = str(m)
            strip = capi.Strip("jacobi5general", p, halo, 3 * rows, args.width, 1, 3, exchange_fn_address=no_exchange)
            init = torch.rand(rows, args.width, device="cuda")
            torch.cuda.synchronize()
            strip.upload_from_device(0, init.data_ptr(), init.numel() * 4)
            strip.synchronize()
            strip.advance(0, args.generations, blocking=True)
            best, enqueue = 1e9, 1e9
            for _ in range(args.reps):
                launches0, exchanges0 = strip.counters()
                t0 = time.perf_counter()
                strip.advance(0, args.generations, blocking=False)
                t1 = time.perf_counter()
                strip.synchronize()
                best = min(best, time.perf_counter() - t0)
                enqueue = min(enqueue, t1 - t0)
                launches1, exchanges1 = strip.counters()
            print(json.dumps({"rows_per_gpu": rows, "width": args.width, "generations": args.generations,
                              "exchange_every": m, "Gcell_updates_per_s_per_gpu": round(rows * args.width * args.generations / best / 1e9, 1),
                              "ms_per_step": round(best * 1e3, 3), "launches_per_step": launches1 - launches0,
                              "exchanges_per_step": exchanges1 - exchanges0, "host_enqueue_ms": round(enqueue * 1e3, 3), "kernel": "general" if args.general else "uniform"}), flush=True)
            strip.close()
            del init


if __name__ == "__main__":
    main()
