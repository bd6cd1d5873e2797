#!/bin/bash
# Run ON THE GPU BOX: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the config-5 fleet step, per fleet_step_kernel launch.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_fleet
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d "$OUT/$c" -o b --output-format csv -- python "$REPO/tools/exp_hetero_trace.py" 256 8 > "$OUT/$c.log" 2>&1
done
cd "$REPO"
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "fleet_step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals.append((int(r.get("Grid_Size") or 0), float(r["Counter_Value"])))
    res[c] = vals
# launches with a window chunk have more workgroups than the chunk-less 8th step
sizes = sorted({s for s, _ in res["FETCH_SIZE"]})
print("launch sizes (threads):", sizes)
tot = {}
for c, vals in res.items():
    for s in sizes:
        v = [x for sz, x in vals if sz == s]
        if v:
            v.sort(); v = v[len(v) // 4: max(len(v) // 4 + 1, 3 * len(v) // 4)]
            tot[(c, s)] = sum(v) / len(v)
            print(f"{c} threads={s}: {tot[(c, s)]:.0f} KiB per launch")
small, big = sizes[0], sizes[-1]
b = lambda s: (2 * tot[("FETCH_SIZE", s)] + tot[("WRITE_SIZE", s)]) * 1024      # gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads
per8 = 7 * b(big) + b(small)
print(f"HBM bytes per launch: with chunk {b(big) / 1e6:.1f} MB, without {b(small) / 1e6:.1f} MB -> per fleet step (7 + 1 of 8): {per8 / 8 / 1e6:.1f} MB")
PY
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
