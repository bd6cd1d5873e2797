#!/bin/bash
# Run ON THE GPU BOX (through gpurun) at the commit whose numbers are to be judged: everything bench.py quotes from profiles/, stamped
# with the source hash of the libmgx.so that ran (bench.py quotes a counter file only when its hash is the running library's).
#   1. rocprofv3 --kernel-trace --stats of the driver's command             -> stats + summary (kernel durations vs the bench line)
#   2. FETCH_SIZE / WRITE_SIZE passes of the same command (separate passes)  -> traffic.json
#   3. SQ counters (VALU issue) of the fused and the rule-based rollout     -> valu.json
#   4. FETCH / WRITE of the config-5 fleet, default rows, float64 + float32  -> traffic_fleet_rows_*.json   (ALL=1: + row-major, views)
#   5. FETCH / WRITE of the four general-path legs                           -> traffic_general.json
#   6. FETCH / WRITE of the single step WITH rows / with rows + log (own passes: same kernel and launch size as the plain step)
#                                                                            -> traffic.json keys "step_kernel<3,false>+rows[+log]"
# Usage: tools/gpu_profile_r06.sh [tag]          -> gpurun_out/prof_<tag>/*
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
HASH=$(cd "$REPO" && python -c "from pymgrid_amd import _lib; print(_lib.built_hash() or _lib.source_hash())")
echo "csrc_hash $HASH" > "$OUT/csrc_hash.txt"
ARGS="--gpus 1 --steps 20 --warmup 5"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- python "$REPO/bench.py" $ARGS --detail "$OUT/bench_detail.json" > "$OUT/stats.log" 2> "$OUT/stats.err"
PMCARGS="$ARGS --no-cpu-baseline --hetero-steps 0 --no-closed-loop --detail $OUT/bench_detail_pmc.json"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o bench --output-format csv -- python "$REPO/bench.py" $PMCARGS > "$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o bench --output-format csv -- python "$REPO/bench.py" $PMCARGS > "$OUT/pmc_write.log" 2>&1
cd "$REPO"
python tools/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
python - "$OUT/traffic.json" "$HASH" <<'PY'
import json, sys
f, h = sys.argv[1], sys.argv[2]
d = json.load(open(f)); d.update(grids=100000, chunk=64, bench_args="--gpus 1 --steps 20 --warmup 5", series="factorised", csrc_hash=h)
d["note"] = ("by_launch_threads is keyed by kernel SPECIALISATION and launch size; the headline (fused, factorised, two 50 000-grid shards) and "
             "the single-step kernel run in the same bench command")
json.dump(d, open(f, "w"), indent=1)
PY
# 6. the single step with observation rows / with rows + log: the same kernel specialisation at the same launch size as the plain step,
#    so each shape gets counter passes of its own (bench.py --legs <leg>) and a key of its own in traffic.json
for LEG in step_env_obs step_full; do
  cd /tmp
  LO=$OUT/leg_$LEG; mkdir -p "$LO"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace -d "$LO/$c" -o b --output-format csv -- python "$REPO/bench.py" --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --hetero-steps 0 --legs $LEG --detail /dev/null > "$LO/$c.log" 2>&1
  done
  cd "$REPO"
  python - "$LO" "$OUT/traffic.json" $LEG <<'PY'
import csv, glob, json, os, sys
lo, tj, leg = sys.argv[1], sys.argv[2], sys.argv[3]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(os.path.join(lo, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "mgx::step_kernel<3, false>" in r["Kernel_Name"] and 100000 <= int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0) < 145000:
                v.append(float(r["Counter_Value"]))
    v = sorted(v)[len(v) // 4: max(len(v) // 4 + 1, 3 * len(v) // 4)]
    vals[c] = sum(v) / len(v) if v else None
    print(leg, c, "launches averaged", len(v), "KiB per launch", vals[c])
if None not in vals.values():
    d = json.load(open(tj))
    key = "step_kernel<3,false>" + {"step_env_obs": "+rows", "step_full": "+rows+log"}[leg]
    d.setdefault("by_launch_threads", {})[key] = {"100096": {"fetch_size_kib": vals["FETCH_SIZE"], "write_size_kib": vals["WRITE_SIZE"],
                                                             "hbm_bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024}}
    json.dump(d, open(tj, "w"), indent=1)
PY
  rm -rf "$LO"
done
# 3. VALU issue: SQ counters of the two fused kernels (separate pass, kernel-trace only), the headline mode of each run alone
cd /tmp
SQARGS="--gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --hetero-steps 0 --no-side-modes --no-closed-loop --detail /dev/null"
for MODE in fused rbc; do
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d "$OUT/sq_$MODE" -o b --output-format csv -- python "$REPO/bench.py" $SQARGS --mode $MODE > "$OUT/sq_$MODE.log" 2>&1
done
# (the full-output form of the fused kernel: a side leg of the fused headline)
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d "$OUT/sq_rich" -o b --output-format csv -- python "$REPO/bench.py" --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --hetero-steps 0 --no-closed-loop --legs fused_rich --detail /dev/null > "$OUT/sq_rich.log" 2>&1
# (the general path's K-step launch: 100 000 grids, 32 steps per launch)
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d "$OUT/sq_kstep" -o b --output-format csv -- python "$REPO/tools/exp_r5_general_prof.py" kstep 512 > "$OUT/sq_kstep.log" 2>&1
# (three of a kind, and RuleBasedControl's list roll-out on the general path: 100 000 grids, 32 steps per launch)
for GL in kstep3 rbc; do
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d "$OUT/sq_g$GL" -o b --output-format csv -- python "$REPO/tools/exp_r5_general_prof.py" $GL 512 > "$OUT/sq_g$GL.log" 2>&1
done
cd "$REPO"
python - "$OUT" "$HASH" <<'PY'
import csv, glob, json, os, re, subprocess, sys
from collections import defaultdict
out, h = sys.argv[1], sys.argv[2]
def spec(k):
    m = re.search(r"mgx::([a-z_0-9]+)(<[^(]*>)?\(", k)
    return (m.group(1) + (m.group(2) or "")).replace(" ", "") if m else None
kern = {}
for mode in ("fused", "rbc", "rich", "kstep", "gkstep3", "grbc"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, f"sq_{mode}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            sp = spec(r["Kernel_Name"])
            if sp and sp.split("<")[0] in ("step_k_kernel", "rollout_kernel", "step_k_multi_small_kernel", "rollout_multi_small_kernel"):
                acc[sp][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for sp, d in acc.items():
        if mode == "rich" and sp in kern:           # (the headline's kernel runs in this pass too: its own pass above is the one kept)
            continue
        mean = {c: sum(v) / len(v) for c, v in d.items()}
        waves = mean.get("SQ_WAVES", 0.0)
        # launches of 50 000 grids (two shards): 64 grids per wave, 192..256 grids per workgroup of 4 waves
        kern[sp] = {"valu_active_cycles_per_launch": 4.0 * mean.get("SQ_ACTIVE_INST_VALU", 0.0), "valu_insts_per_launch": mean.get("SQ_INSTS_VALU"),
                    "waves_per_launch": waves, "wave_cycles_per_launch": 4.0 * mean.get("SQ_WAVE_CYCLES", 0.0),
                    "grids_per_launch": 100000 if "kstep" in mode or mode == "grbc" else 50000, "steps_per_launch": 32 if "kstep" in mode or mode == "grbc" else 64,
                    "valu_insts_per_wave_step": (mean.get("SQ_INSTS_VALU", 0.0) / waves / (32 if "kstep" in mode or mode == "grbc" else 64)) if waves else None,
                    "launches_averaged": len(next(iter(d.values())))}
sclk = None
try:
    txt = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
    card = next(iter(json.loads(txt[txt.index("{"):]).values()))
    m = re.search(r"(\d+)Mhz", str([v for k, v in card.items() if "sclk clock speed" in k][0]))
    sclk = int(m.group(1)) if m else None
except Exception:
    pass
json.dump({"csrc_hash": h, "kernels": kern, "sclk_mhz": 2400 if sclk is None or sclk < 1000 else sclk,
           "source": "tools/gpu_profile_r06.sh: rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES (one pass, "
                     "kernel-trace only) of bench.py --steps 4 --warmup 1 --no-side-modes --mode fused / rbc: means per 50 000-grid launch; SQ_* "
                     "cycle counters tick in quad-cycles (x 4); sclk_mhz: the clock the VALU peak is priced at (idle reading replaced by 2400)"},
          open(os.path.join(out, "valu.json"), "w"), indent=1)
print(json.dumps(kern, indent=1))
PY
# 4. the config-5 fleet
CONTRACTS="rows"; [ "${ALL:-0}" = 1 ] && CONTRACTS="rows rows_rowmajor views"
for CONTRACT in $CONTRACTS; do for DT in float64 float32; do
  cd /tmp
  FO=$OUT/fleet_${CONTRACT}_${DT}; mkdir -p "$FO"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace -d "$FO/$c" -o b --output-format csv -- python "$REPO/tools/exp_fleet_prof.py" $CONTRACT $DT 3000 32 > "$FO/$c.log" 2>&1
  done
  cd "$REPO"
  python - "$FO" 1024 $CONTRACT $DT "$HASH" "$OUT" <<'PY' | tee -a "$OUT/fleet_summary.txt"
import csv, glob, json, os, sys
out, steps, contract, dt, h, top = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
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
        print(f"{contract} {dt} {c:11s} {name:40s} {n:5d} launches  {v / n:12.0f} KiB per launch")
    tot[c] = sum(v for _, v in by.values())
per_step = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / steps
print(f"HBM bytes per fleet step ({contract}, {dt}): {per_step / 1e6:.1f} MB "
      f"(read {2 * tot['FETCH_SIZE'] * 1024 / steps / 1e6:.1f} + written {tot['WRITE_SIZE'] * 1024 / steps / 1e6:.1f})")
json.dump({"kernel": "fleet_step_kernel_v" + (" + obs_windows_k_kernel" if contract.startswith("rows") else ""), "grids_per_gpu": 99999,
           "series": "factorised", "contract": contract, "dtype": dt, "obs_prefetch": 32, "hbm_bytes_per_fleet_step": per_step, "csrc_hash": h,
           "source": "tools/gpu_profile_r06.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bytes = (2*FETCH_SIZE + "
                     f"WRITE_SIZE)*1024 summed over every kernel of the last {steps} fleet steps (step launches + ring refills) / {steps}"},
          open(os.path.join(top, f"traffic_fleet_{contract}_{dt}.json"), "w"), indent=1)
PY
  rm -rf "$FO"
done; done
# 5. the general path, one leg per pass
for LEG in single kstep kstep3 gymrows; do
  cd /tmp
  GO=$OUT/general_$LEG; mkdir -p "$GO"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace -d "$GO/$c" -o b --output-format csv -- python "$REPO/tools/exp_r5_general_prof.py" $LEG 512 > "$GO/$c.log" 2>&1
  done
done
cd "$REPO"
python - "$OUT" "$HASH" <<'PY' | tee "$OUT/general_summary.txt"
import csv, glob, json, os, sys
top, h = sys.argv[1], sys.argv[2]
res = {}
# what "one launch" of bench.py's leg is, which kernel marks it, and over how many of them (the tail of the run) the bytes are averaged
legs = {"single": ("single_steps", "step_multi_kernel", 256, 1), "kstep": ("k_step_launches", "step_k_multi_small_kernel", 8, 1),
        "kstep3": ("k_step_3_of_a_kind", "step_k_multi_small_kernel", 8, 1),
        "gymrows": ("gym_steps_rows_h24", "step_multi_kernel", 256, 1)}
for leg, (key, marker, n_tail, _) in legs.items():
    tot, per = {}, {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = []
        for f in glob.glob(os.path.join(top, f"general_{leg}", c, "**", "*counter_collection.csv"), recursive=True):
            rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "mgx::" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        idx = [j for j, r in enumerate(rows) if marker in r["Kernel_Name"]]
        if len(idx) < n_tail:
            print(f"{leg}: only {len(idx)} {marker} launches in the {c} pass"); break
        tail = rows[idx[-n_tail]:]
        by = {}
        for r in tail:
            name = r["Kernel_Name"].split("mgx::")[1].split("(")[0]
            n, v = by.get(name, (0, 0.0))
            by[name] = (n + 1, v + float(r["Counter_Value"]))
        for name, (n, v) in sorted(by.items()):
            print(f"{leg:8s} {c:11s} {name:46s} {n:5d} launches  {v / n:12.0f} KiB per launch")
        tot[c] = sum(v for _, v in by.values())
    else:
        b = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / n_tail
        print(f"{leg}: HBM bytes per {marker} launch (everything launched beside it included): {b / 1e6:.1f} MB")
        res[key] = {"hbm_bytes_per_launch": b, "grids_per_gpu": 100000, "launches_averaged": n_tail}
res.update(csrc_hash=h, source="tools/gpu_profile_r06.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/exp_r5_general_prof.py "
                               "single | kstep | kstep3 | gymrows; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 over every mgx kernel of the tail of the run / "
                               "the marker kernel's launches (a K-step launch = 32 env-steps; gymrows: step launches + ring refills per step)")
json.dump(res, open(os.path.join(top, "traffic_general.json"), "w"), indent=1)
PY
cat "$OUT/summary.txt" | tail -30
# keep the merged-back payload small: drop the raw per-dispatch traces, keep stats + summaries
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*counter_collection.csv" -size +1M -delete
rm -rf "$OUT"/general_*/FETCH_SIZE "$OUT"/general_*/WRITE_SIZE
