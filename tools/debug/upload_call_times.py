"""Host time of each call of an upload in row blocks (is a hipMemcpyAsync of 128 MiB asynchronous?) and when the
blocks arrive.  -> a note in profiles/r04_stream_upload.txt"""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: F401  (one HIP runtime per process)

from stencilstream_amd import capi

capi.init(0)
lib = capi.load()
n = 1 << 30
host, dev = C.c_void_p(), C.c_void_p()
capi.check(lib.ststhip_host_malloc(C.byref(host), n), "host_malloc")
C.memset(host, 1, n)
capi.check(lib.ststhip_malloc(C.byref(dev), n), "malloc")
up = capi.upload_stream()
for rep in range(3):
    events = []
    t0 = time.perf_counter()
    marks = []
    for b in range(8):
        capi.check(lib.ststhip_memcpy_h2d(C.c_void_p(dev.value + b * (n // 8)), C.c_void_p(host.value + b * (n // 8)), n // 8,
                                          C.c_void_p(up)), "h2d")
        ev = C.c_void_p()
        lib.ststhip_event_create(C.byref(ev))
        lib.ststhip_event_record(ev, C.c_void_p(up))
        events.append(ev)
        marks.append((time.perf_counter() - t0) * 1e3)
    other = C.c_void_p()
    t1 = time.perf_counter()
    capi.check(lib.ststhip_malloc(C.byref(other), n), "malloc")
    t2 = time.perf_counter()
    arrive = []
    for ev in events:
        lib.ststhip_event_synchronize(ev)
        arrive.append((time.perf_counter() - t0) * 1e3)
    print(f"rep {rep}: copy calls returned at", " ".join(f"{m:.2f}" for m in marks), "ms; a 1 GiB malloc meanwhile",
          f"{(t2 - t1) * 1e3:.2f} ms; blocks arrived at", " ".join(f"{a:.2f}" for a in arrive), "ms")
    lib.ststhip_free(other)
    capi.events_destroy(events)
