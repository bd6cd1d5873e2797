#!/bin/bash
# Run ON THE GPU BOX: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of config-5 fleet stepping (99 999 grids, H = 24,
# T = 8 760, factorised series), per fleet step: every mgx kernel dispatched from the first of the last STEPS fleet_step_kernel
# launches on (step launches + ring refills).   usage: gpu_pmc_fleet.sh [rows|views] [float64|float32]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
CONTRACT=${1:-rows}; DT=${2:-float64}; STEPS=1024
OUT=$REPO/gpurun_out/pmc_fleet_${CONTRACT}_${DT}
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d "$OUT/$c" -o b --output-format csv -- python "$REPO/tools/exp_fleet_prof.py" $CONTRACT $DT 3000 > "$OUT/$c.log" 2>&1
done
cd "$REPO"
python - "$OUT" $STEPS $CONTRACT $DT <<'PY' | tee "$OUT/summary.txt"
import csv, glob, json, os, sys
out, steps, contract, dt = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "mgx::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    step_idx = [j for j, r in enumerate(rows) if "fleet_step_kernel" in r["Kernel_Name"]]
    tail = rows[step_idx[-steps]:]
    by = {}
    for r in tail:
        name = r["Kernel_Name"].split("mgx::")[1].split("(")[0]
        n, v = by.get(name, (0, 0.0))
        by[name] = (n + 1, v + float(r["Counter_Value"]))
    for name, (n, v) in sorted(by.items()):
        print(f"{c:11s} {name:40s} {n:5d} launches  {v / n:12.0f} KiB per launch")
    tot[c] = sum(v for _, v in by.values())
# gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads (MI355X_MICROARCH.md)
per_step = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / steps
print(f"HBM bytes per fleet step ({contract}, {dt}): {per_step / 1e6:.1f} MB "
      f"(read {2 * tot['FETCH_SIZE'] * 1024 / steps / 1e6:.1f} + written {tot['WRITE_SIZE'] * 1024 / steps / 1e6:.1f})")
json.dump({"kernel": "fleet_step_kernel" + (" + obs_windows_k_kernel" if contract == "rows" else ""), "grids_per_gpu": 99999,
           "series": "factorised", "contract": contract, "dtype": dt, "obs_prefetch": 16, "hbm_bytes_per_fleet_step": per_step,
           "source": "tools/gpu_pmc_fleet.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bytes = (2*FETCH_SIZE + "
                     f"WRITE_SIZE)*1024 summed over every kernel of the last {steps} fleet steps (step launches + ring refills) / {steps}"},
          open(os.path.join(out, f"traffic_fleet_{contract}_{dt}.json"), "w"), indent=1)
PY
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
