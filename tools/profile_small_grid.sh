#!/bin/bash
# Small grids are bound by the launch-to-launch latency of dependent kernels, not by the sweep itself: kernel
# duration (rocprofv3 --kernel-trace --stats) next to the wall time per launch of the unchanged Jacobi example.
# usage (GPU box): tools/profile_small_grid.sh [iterations]
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
EXE="$REPO/build/examples/jacobi_Jacobi5General_hip"
OUT="$REPO/gpurun_out/small_grid"; mkdir -p "$OUT"
N="${1:-40000}"
cd /tmp && export TMPDIR=/tmp
for wh in 256 1024; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$wh" -- "$EXE" $wh $wh $N /dev/null 0.2 0.2 0.2 0.2 0.2 > "$OUT/prof_$wh.log" 2>&1 < /dev/null
  stats=$(find "$OUT/prof_$wh" -name "*kernel_stats.csv" | head -1)
  echo "== ${wh}^2, $N generations: kernel statistics (ns)"; [ -n "$stats" ] && cut -c1-220 "$stats" | head -4
  find "$OUT/prof_$wh" -name "*kernel_trace.csv" -delete
  echo "== ${wh}^2, $N generations: application walltime without the profiler"
  "$EXE" $wh $wh $N /dev/null 0.2 0.2 0.2 0.2 0.2 < /dev/null | grep Walltime
done
