#!/bin/bash
# A/B of two builds of libststhip.so on the same box, alternating so box-to-box and thermal drift
# cancel:  tools/ab_lib.sh build/ab/libststhip_head.so build/ab/libststhip_new.so [bench.py args]
# Prints Gcell-updates/s and the dominant kernel's ms for every run.
set -u
cd "$(dirname "$0")/.."
A=$1; B=$2; shift 2
keep=$(mktemp); cp stencilstream_amd/libststhip.so "$keep"
export STSTHIP_BENCH_MINIMAL=1
for round in 1 2 3; do
    for which in "$A" "$B"; do
        cp "$which" stencilstream_amd/libststhip.so
        python bench.py "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$which', d['value'], d['roofline'].get('kernel_ms'))"
    done
done
cp "$keep" stencilstream_amd/libststhip.so; rm -f "$keep"
