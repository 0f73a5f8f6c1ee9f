#!/bin/bash
# Flow check of the N > 1 path of bench.py where only one GPU exists: two ranks share cuda:0, ghost
# rows travel through host memory over gloo.  The number it prints is NOT a multi-GPU result.
cd "$(dirname "$0")/.."
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 2 --warmup 1 --rows-per-gpu ${ROWS:-8192} --debug-host-exchange "$@"
