#!/bin/bash
# A/B of two builds of libststhip.so over every BASELINE configuration (tools/bench_apps.py), same box, alternating:
#   tools/ab_lib_apps.sh build/ab/libststhip_head.so build/ab/libststhip_new.so [apps...]
set -u
cd "$(dirname "$0")/.."
A=$1; B=$2; shift 2
keep=$(mktemp); cp stencilstream_amd/libststhip.so "$keep"
for round in 1 2; do
    for which in "$A" "$B"; do
        cp "$which" stencilstream_amd/libststhip.so
        python tools/bench_apps.py "$@" 2>/dev/null < /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$which', d['app'], d['Gcell_updates_per_s'])"
    done
done
cp "$keep" stencilstream_amd/libststhip.so; rm -f "$keep"
