#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + stats and the HBM PMC passes for bench.py.
# Usage: tools/gpu_profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/*      (default args: the driver's command)
# PMC passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950: TCC has 4 slots,
# FETCH_SIZE takes 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel-trace only.
set -u
TAG=${1:-run}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="${*:---gpus 1 --steps 20 --warmup 5}"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- python "$REPO/bench.py" $ARGS > "$OUT/stats.log" 2> "$OUT/stats.err"
PMCARGS="$ARGS --no-cpu-baseline --hetero-steps 0 --no-closed-loop"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o bench --output-format csv -- python "$REPO/bench.py" $PMCARGS > "$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o bench --output-format csv -- python "$REPO/bench.py" $PMCARGS > "$OUT/pmc_write.log" 2>&1
cd "$REPO"
python tools/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
# stamp the workload the counters belong to (bench.py only reports `traffic` for a matching launch shape)
python - "$OUT/traffic.json" $ARGS <<'PY'
import json, sys
f, a = sys.argv[1], sys.argv[2:]
def arg(name, default):
    return int(a[a.index(name) + 1]) if name in a else default
d = json.load(open(f)); d["grids"] = arg("--grids", 100000); d["chunk"] = arg("--chunk", 64); d["bench_args"] = " ".join(a)
d["series"] = a[a.index("--series") + 1] if "--series" in a else "factorised"
d["note"] = ("by_launch_threads is keyed by kernel SPECIALISATION: the factorised (…,true>) and materialised (…,false>) forms of the "
             "fused kernels run in the same bench command (headline and 'other')")
json.dump(d, open(f, "w"), indent=1)
PY
cat "$OUT/summary.txt"
# keep the merged-back payload small: drop the raw per-dispatch traces, keep stats + summary
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
