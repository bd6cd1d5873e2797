#!/bin/bash
# Round 6, second GPU call: why the two-chain Gym step is host-bound (launchbench2: two threads inside the HIP runtime) and what the
# GPU side looks like (kernel trace of the 1 / 2 / 4-chain legs: durations, cadence, overlap); the new scenario-file GPU tests.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd "$REPO"
[ -x tools/bin/launchbench2 ] || hipcc --offload-arch=gfx950 -O3 -pthread tools/launchbench2.hip -o tools/bin/launchbench2
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d "$OUT/chain_trace" -o t --output-format csv -- python "$REPO/tools/exp_r6_two_chains.py" 100000 2048 trace > "$OUT/exp_two_chains_traced.txt" 2>&1
cd "$REPO"
grep -v amdgpu.ids "$OUT/exp_two_chains_traced.txt"
cp "$OUT/chain_trace/"*agent_info.csv "$OUT/" 2>/dev/null
python - "$OUT/chain_trace" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        rd = csv.DictReader(fh)
        print(rd.fieldnames)
        for j, r in enumerate(rd):
            if j < 3: print(r)
PY
python tools/summarize_chain_trace.py "$OUT/chain_trace" > "$OUT/chain_trace_summary.txt" 2>&1
cat "$OUT/chain_trace_summary.txt"
rm -rf "$OUT/chain_trace"
