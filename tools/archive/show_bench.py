#!/usr/bin/env python3
"""Print the interesting numbers of a bench.py JSON line."""
import json
import sys

d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
r = d["roofline"]
print(f"headline {d['value'] / 1e9:.2f} G env-steps/s  {d['ms_per_step'] * 1e3:.2f} us/round  frac {r['frac']:.3f}  "
      f"cadence {r['avg_launch_us']:.2f} us  kernel {r.get('kernel_avg_duration_us')}  traffic {r.get('traffic')}")
for k, v in (d.get("other") or {}).items():
    if "error" in v:
        print(f"  {k:42s} FAILED: {v['error']}")
        continue
    q = v["roofline"]
    print(f"  {k:42s} {v['value'] / 1e9:8.2f} G  {v['ms_per_step'] * 1e3:8.2f} us/round  frac {q['frac']:.3f}  "
          f"launch {q['avg_launch_us']:.2f} us  kernel {q.get('kernel_avg_duration_us')}")
c = d.get("closed_loop_policy_gym_steps")
if c:
    print(f"  closed loop (policy on device, obs rows): " + (f"FAILED {c['error']}" if "error" in c else
          f"{c['us_per_step']:.1f} us/step  {c['value'] / 1e9:.2f} G env-steps/s"))
    for k in ("float64", "float32_io_rotating_outputs", "float32_io_auto_reset_168_step_episodes"):
        if k in c and "error" not in c:
            print(f"    {k:40s} {c[k]['us_per_step']:.1f} us/step = policy kernels {c[k]['policy_kernels_alone_us']:.1f} + env.step "
                  f"{c[k]['env_step_alone_us']:.1f} (each alone)")
h = d.get("hetero_h24_gym_steps")
if h and "error" in h:
    print(f"  hetero FAILED: {h['error']}")
elif h:
    for k in ("float64_rows", "float32_rows", "float64_views", "float32_views"):
        if k in h:
            q = h[k]["roofline"]
            print(f"  hetero {k}: {h[k]['us_per_step']:.1f} us/step wall  {q['avg_launch_us']:.1f} us gpu  {h[k]['value'] / 1e9:.2f} G  "
                  f"frac {q['frac']:.3f}  traffic {q.get('traffic')}")
c = d.get("cpu_baseline")
if c and "error" in c:
    print(f"  cpu baseline FAILED: {c['error']}")
elif c:
    print(f"  cpu {c['value'] / 1e6:.0f} M on {c['cores']} threads, 1 thread {c['value_1thread'] / 1e6:.1f} M, quota {c.get('cgroup_cpu_quota')}")
print("  per rank:", [round(x / 1e9, 2) for x in d.get("per_rank_env_steps_per_s", [])])
