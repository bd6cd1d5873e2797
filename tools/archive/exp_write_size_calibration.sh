#!/bin/bash
# WRITE_SIZE (rocprofv3 PMC) against a known byte count: tools/bin/fillbench writes 4 GiB per launch
# (8 / 16 B per lane, plain / non-temporal); MI355X_MICROARCH.md calls WRITE_SIZE uncalibrated.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ws_cal; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for c in WRITE_SIZE FETCH_SIZE; do
timeout 200 rocprofv3 --pmc $c --kernel-trace -d "$OUT/$c" -o b --output-format csv -- "$REPO/tools/bin/fillbench" > "$OUT/$c.log" 2>&1
done
python - "$OUT" <<'PY' | tee "$REPO/gpurun_out/write_size_calibration.txt"
import csv, glob, os, sys
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    acc = {}
    for f in glob.glob(os.path.join(sys.argv[1], c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            name = r["Kernel_Name"].split("(")[0]
            acc.setdefault(name, []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{c} {k:60s} {len(v):4d} launches  {sum(v) / len(v) * 1024 / 2**30:8.3f} GiB per launch (4 GiB written" + (", 4 GiB read" if "copy" in k else "") + ")")
PY
rm -rf "$OUT"
