#!/bin/bash
# the measurements quoted for the final build: bench line (twice), strips, convection, examples, hotspot timeline
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out/final
python3 bench.py > gpurun_out/final/bench1.json 2> gpurun_out/final/bench1.err
python3 bench.py > gpurun_out/final/bench2.json 2> gpurun_out/final/bench2.err
python3 tools/bench_strip.py --rows 2048 4096 8192 --exchange-every 1 2 4 --reps 3 > gpurun_out/final/strips.jsonl 2>/dev/null
bash tools/convection_bench.sh > gpurun_out/final/convection.txt 2>&1
bash tools/run_examples_bench.sh > gpurun_out/final/examples.txt 2>&1
bash tools/debug/trace_hotspot.sh > gpurun_out/final/hotspot_trace.txt 2>&1
python3 tools/bench_apps.py > gpurun_out/final/bench_apps.jsonl 2>/dev/null
echo done
