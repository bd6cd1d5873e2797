#!/bin/bash
# Run ON THE GPU BOX: kernel timeline of the config-5 fleet (rows contract) for ring depths K = 16 / 32.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for K in ${KS:-16 32}; do
  rm -rf /tmp/tr_$K
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr_$K -o t --output-format csv -- python "$REPO/tools/exp_hetero_trace.py" 640 $K float64 ahead > "$OUT/fleet_trace_K$K.log" 2>&1
  python "$REPO/tools/trace_timeline.py" /tmp/tr_$K 140 > "$OUT/fleet_timeline_K$K.txt" 2>&1
  find /tmp/tr_$K -name "*kernel_stats.csv" -exec cp {} "$OUT/fleet_kernel_stats_K$K.csv" \;
done
grep -h hetero "$OUT"/fleet_trace_K*.log
