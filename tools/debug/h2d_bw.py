import time, torch
n = 1 << 30
host = torch.empty(n, dtype=torch.uint8).pin_memory()
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for chunks in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(chunks)]
    step = n // chunks
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                dev[i * step:(i + 1) * step].copy_(host[i * step:(i + 1) * step], non_blocking=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"H2D 1 GiB in {chunks} chunk(s) on {chunks} stream(s): {dt*1e3:.2f} ms = {n/dt/1e9:.1f} GB/s", flush=True)
for chunks in (1, 4):
    streams = [torch.cuda.Stream() for _ in range(chunks)]
    step = n // chunks
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                host[i * step:(i + 1) * step].copy_(dev[i * step:(i + 1) * step], non_blocking=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"D2H 1 GiB in {chunks} chunk(s): {dt*1e3:.2f} ms = {n/dt/1e9:.1f} GB/s", flush=True)
