#!/usr/bin/env python3
"""Host-side cost of one ghost-row exchange as stencilstream_amd/dist.py issues it (grouped p2p through
torch.distributed on the RCCL backend), measured with a one-rank process group that sends to itself:
the data path is meaningless, the Python / launch overhead per call is what the multi-GPU driver pays
once per pass.  Also times the native path (ststhip_comm_exchange_rows) the same way."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    rows, width = 12, 16384
    a = torch.zeros(rows, width * 4, dtype=torch.uint8, device="cuda")
    b = torch.zeros_like(a)
    c = torch.zeros_like(a)
    d = torch.zeros_like(a)
    comm = torch.cuda.Stream()

    def exchange():
        ops = [dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0),
               dist.P2POp(dist.isend, c, 0), dist.P2POp(dist.irecv, d, 0)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    with torch.cuda.stream(comm):
        for _ in range(5):
            exchange()
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            exchange()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
    print(f"torch.distributed grouped p2p (4 ops of {a.numel()} bytes): host {host / n * 1e6:.1f} us per exchange, "
          f"{total / n * 1e6:.1f} us per exchange including the GPU side")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
