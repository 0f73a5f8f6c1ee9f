#!/usr/bin/env python3
"""How long the runtime's device allocations take (first and later ones of 1 GiB), and the first copy into one."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stencilstream_amd import capi
capi.init(0)
lib = capi.load()
GiB = 1 << 30
host = C.c_void_p()
capi.check(lib.ststhip_host_malloc(C.byref(host), GiB), "host")
C.memset(host, 1, GiB)
ptrs = []
for i in range(4):
    p = C.c_void_p()
    t0 = time.perf_counter()
    capi.check(lib.ststhip_malloc(C.byref(p), GiB), "malloc")
    t1 = time.perf_counter()
    print(f"ststhip_malloc #{i} of 1 GiB: {(t1 - t0) * 1e3:.2f} ms", flush=True)
    ptrs.append(p)
stream = C.c_void_p()
capi.check(lib.ststhip_default_stream(C.byref(stream)), "stream")
if os.environ.get("WARM"):
    small = C.c_void_p(); capi.check(lib.ststhip_host_malloc(C.byref(small), max(4096, int(os.environ["WARM"]))), "host")
    if os.environ.get("WARM_SRC") == "big":
        small = host
    t0 = time.perf_counter()
    capi.check(lib.ststhip_memcpy_h2d(ptrs[3], small, int(os.environ["WARM"]), stream), "h2d")
    capi.check(lib.ststhip_stream_synchronize(stream), "sync")
    print(f"warm-up h2d of {os.environ['WARM']} bytes: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
for i in range(3):
    t0 = time.perf_counter()
    capi.check(lib.ststhip_memcpy_h2d(ptrs[i], host, GiB, stream), "h2d")
    capi.check(lib.ststhip_stream_synchronize(stream), "sync")
    print(f"h2d of 1 GiB into buffer #{i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
t0 = time.perf_counter()
capi.check(lib.ststhip_memcpy_h2d(ptrs[0], host, GiB, stream), "h2d")
capi.check(lib.ststhip_stream_synchronize(stream), "sync")
print(f"h2d again into buffer #0: {(time.perf_counter() - t0) * 1e3:.2f} ms")
