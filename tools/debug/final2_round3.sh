#!/bin/bash
# last GPU call of the round: bench profile (trace + PMC passes), two plain bench lines, grid sizes, config 5 on one GPU
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out/final2
bash tools/profile_bench_r03.sh r03_bench > gpurun_out/final2/profile.log 2>&1 && echo profile done
cd "$REPO"
python3 bench.py > gpurun_out/final2/bench1.json 2> gpurun_out/final2/bench1.err
python3 bench.py > gpurun_out/final2/bench2.json 2> gpurun_out/final2/bench2.err
bash tools/debug/sweep_sizes.sh
cp gpurun_out/sweep_sizes.txt gpurun_out/final2/
python3 bench.py --config5 --steps 3 --warmup 1 --no-legs --no-cpu-baseline > gpurun_out/final2/config5.json 2> gpurun_out/final2/config5.err
bash tools/run_examples_bench.sh > gpurun_out/final2/examples.txt 2>&1
echo done
