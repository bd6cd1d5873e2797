#!/bin/bash
# Copy what tools/gpu_profile_r06.sh left under gpurun_out/prof_<tag>/ into profiles/<tag>/ (the tracked, judged place).
# usage: tools/copy_profiles.sh [tag]
set -eu
TAG=${1:-r06}
P=gpurun_out/prof_$TAG; D=profiles/$TAG
mkdir -p "$D"
cp $P/traffic.json $P/valu.json $P/traffic_fleet_*.json $P/traffic_general.json $P/csrc_hash.txt "$D"/
cp $P/summary.txt "$D"/rocprof_summary_$TAG.txt
cp $P/fleet_summary.txt "$D"/pmc_fleet_step_traffic.txt
cp $P/stats/bench_kernel_stats.csv "$D"/kernel_stats_$TAG.csv
tail -n 1 $P/stats.log > "$D"/bench_under_rocprof_driver_cmd.json
cp $P/bench_detail.json "$D"/bench_detail_under_rocprof.json
cp $P/general_summary.txt "$D"/pmc_general_path_traffic.txt
cat "$D"/csrc_hash.txt
