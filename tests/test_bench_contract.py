"""bench.py contract: the one-rank JSON line on a GPU box, and the N > 1 path -- `python bench.py --gpus N` launches its own N
ranks (torch.distributed.run, one rank per GPU) -- exercised here with two ranks through the gloo backend (MGX_DIST_BACKEND /
MGX_FORCE_LOCAL_RANK test overrides: on CPU for the launch path alone, on the single GPU for the whole bench).  The real
multi-GPU runs over RCCL are the driver's."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--grids", "20000", "--rows", "700", "--steps", "3", "--warmup", "1", "--chunk", "32", "--launches-per-step", "2",
         "--hetero-steps", "16", "--cpu-seconds", "1", "--prewarm", "0.05"]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"}
MAX_LINE = 4096          # bench.MAX_LINE: the driver keeps a bounded tail of stdout (round 4's 22-KB line did not parse)


def _line(out):
    """The contract: stdout is exactly ONE non-empty line, a JSON record shorter than MAX_LINE."""
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out[-2000:]
    assert len(lines[0]) < MAX_LINE, len(lines[0])
    return json.loads(lines[0])


def _fake_leg(us=5.123456789, frac=0.51234567, lat=None, traffic=1.0e8):
    rf = {"bound": "hbm", "achieved": 4098.7654321, "peak": 8000.0, "unit": "GB/s", "frac": frac, "frac_wall": frac, "traffic": traffic,
          "traffic_source": "profiles/r05/traffic.json", "algorithmic_bytes_per_launch": 98765432.1, "kernel": "k" * 60, "avg_launch_us": us,
          "launches": 1234, "bytes_per_env_step": 42.9375}
    if lat is not None:
        rf["frac_of_latency_model"] = lat
    return {"value": 1.23456789e10, "us_per_step": us, "roofline": rf, "note": "x" * 400}


def test_compact_line_stays_parseable_whatever_the_legs_hold():
    """The stdout record is built from the full record by bench.compact_line: with EVERY leg present (--all-legs, long floats, long
    notes) it stays under MAX_LINE, keeps the contract's keys, and sheds legs -- never the headline -- if it had to."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.MAX_LINE == MAX_LINE
    other = {n: _fake_leg(lat=0.61234 if "single" in n else None) for n in (
        "single_step_launches_one_call", "single_step_launches_python_loop", "single_step_launches_python_loop_with_rows",
        "rbc_rollout_on_device", "fused_launches_one_stream", "fused_launches_materialised", "fused_launches_materialised_one_stream",
        "rbc_rollout_materialised")}
    hetero = {f"{dt}_{c}": _fake_leg() for dt in ("float64", "float32") for c in ("rows", "rows_rowmajor", "views")}
    hetero.update(grids_per_gpu=99999, workload="w" * 300)
    general = {k: _fake_leg() for k in ("single_steps", "k_step_launches", "gym_steps_rows_h24")}
    detail = {"metric": "microgrid env-steps/sec", "value": 1.2345678901234e11, "unit": "env-steps/s", "n_gpus": 8, "steps": 20, "warmup": 5,
              "ms_per_step": 0.8412345678, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
              "config": {"workload": "100000 generated 4-module grids (genset+battery+load+pv) per GPU, T=8760, H=0, normalised random actions "
                                     "(BASELINE configs[2])", "env_steps_per_step": 1024, "steps_per_launch": 64, "grids_per_gpu": 100000,
                         "grids_total": 800000, "mode": "fused", "series": "factorised", "backend": "nccl",
                         "parallelism": "grids sharded x8 ranks, no data-path collective; 2 shard streams per GPU", "step": "s" * 200},
              "roofline": dict(_fake_leg()["roofline"], round_us={"n": 640, "what": "y" * 500}), "roofline_valu": {"frac": 0.4512345},
              "csrc_hash": "0123456789abcdef", "per_rank_env_steps_per_s": [1.54321098765e10] * 8,
              "cpu_baseline": {"value": 3.0261e8, "unit": "env-steps/s", "cores": 16, "kind": "port", "value_1thread": 2.12e7,
                               "host_logical_cpus": 384, "sample": "z" * 400, "sample_short": "first 65536 grids x 64 steps, ~10 s, oracle/mgx_oracle.c"},
              "other": other, "hetero_h24_gym_steps": hetero, "general_path_2g2b1grid": general,
              "closed_loop_policy_gym_steps": {"us_per_step": 31.4159, "loop": "l" * 900},
              "metrics_allreduce": {"sum_last_reward": -1.2345678901234567e9, "mean_soc": 0.61234567890123, "collective_backend": "nccl"}}
    text = bench.compact_line(detail, "fused", "bench_detail.json")
    d = json.loads(text)
    assert len(text) < MAX_LINE and KEYS <= set(d)
    assert d["value"] == detail["value"] and d["ms_per_step"] == detail["ms_per_step"] and d["n_gpus"] == 8
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel", "avg_launch_us"}
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample", "value_1thread"}
    assert len(d["per_rank_env_steps_per_s"]) == 8 and d["csrc_hash"] == "0123456789abcdef"
    assert {"single_step_launches_python_loop", "config5_float64_rows", "config5_float32_rows", "general_single_step", "general_k_step",
            "general_gym_rows_h24"} <= set(d["legs"])
    assert d["legs"]["single_step_launches_python_loop"] == {"us": 5.123, "frac": 0.5123, "lat": 0.6123, "t/a": 1.012}
    # a record that would not fit sheds legs, never the headline
    detail["other"].update({f"extra_leg_{j}_{'n' * 40}": _fake_leg() for j in range(60)})
    text = bench.compact_line(detail, "fused", "bench_detail.json")
    d = json.loads(text)
    assert len(text) < MAX_LINE and KEYS <= set(d) and "config5_float64_rows" in d["legs"] and not any(k.startswith("extra") for k in d["legs"])


def test_gpus_n_launches_its_own_ranks_cpu():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts two ranks itself and they form one process
    group of size 2 (gloo here; nccl = RCCL on a GPU node)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MGX_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert d == {"launch_check": True, "n_gpus": 2, "backend": "gloo", "ranks": [0, 1],
                 "metrics_allreduce": {"sums": [3.0, 2.0], "ok_on_every_rank": True, "collective_backend": "gloo", "error": None}}


def test_gpus_mismatch_fails_loudly():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert r.returncode != 0 and "--gpus 2" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_bench_one_rank_json_line(device, tmp_path):
    det = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--detail", det] + SMALL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    rf = d["roofline"]
    assert set(rf) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel", "avg_launch_us"}
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["launches"] == 3 * 2                                    # every timed step is --launches-per-step full-K launches per shard stream
    assert rf["kernel"] == "step_k_kernel<3,8,double,false,true>" and abs(rf["bytes_per_env_step"] - (158 / 32 + 40)) < 0.01
    assert d["config"]["env_steps_per_step"] == 64 and d["config"]["series"] == "factorised"
    assert abs(d["value"] - 20000 * 3 * 64 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample", "value_1thread"} and d["cpu_baseline"]["kind"] == "port"
    assert "workload" in d["config"] and d["scaling"] == "weak" and d["higher_is_better"] is True
    legs = d["legs"]
    assert {"single_step_launches_one_call", "single_step_launches_python_loop", "single_step_launches_python_loop_with_rows",
            "config5_float64_rows", "config5_float32_rows", "general_single_step", "general_k_step", "general_gym_rows_h24"} <= set(legs)
    assert all("error" not in v and v["us"] > 0 and v["frac"] > 0 for v in legs.values()), legs
    assert legs["single_step_launches_python_loop"]["lat"] > 0 and legs["general_single_step"]["lat"] > 0
    from pymgrid_amd import _lib
    assert d["csrc_hash"] == _lib.source_hash()
    assert rf["traffic"] is None or "STALE" not in str(rf["traffic_source"])          # a stale counter file is never quoted
    # the full record beside it
    full = json.load(open(det))
    assert full["value"] == d["value"] and full["roofline"]["concurrent_streams"] == 2 and full["roofline"]["series"] == "factorised"
    ru = full["roofline"]["round_us"]
    assert ru["n"] == 6 * 2 and ru["min"] <= ru["median"] <= ru["max"]
    assert full["roofline_valu"]["bound"] == "valu"
    gp = full["general_path_2g2b1grid"]
    assert gp["single_steps"]["roofline"]["bytes_per_env_step"] == 8 * (7 + 27 + 6 + 2 * 2 + 2) + 2 * 8 + 8 and gp["gym_steps_rows_h24"]["obs_dim"] == 162
    assert full["other"]["single_step_launches_python_loop"]["roofline"]["bound"] == "latency"
    assert "bench_detail " in r.stderr


@pytest.mark.gpu
def test_bench_all_legs_still_one_short_line(device, tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--all-legs", "--no-cpu-baseline", "--detail", str(tmp_path / "d.json")] + SMALL,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert {"rbc_rollout_on_device", "fused_launches_one_stream", "fused_launches_materialised", "config5_float64_rows_rowmajor",
            "config5_float32_views", "closed_loop_policy"} <= set(d["legs"]) and d["cpu_baseline"] is None
    assert all("error" not in v for v in d["legs"].values()), d["legs"]
    assert d["legs"]["config5_float64_views"]["lat"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_self_launched(device):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MGX_DIST_BACKEND="gloo", MGX_FORCE_LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-side-modes"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["grids_total"] == 40000 and d["value"] > 0
    assert len(d["per_rank_env_steps_per_s"]) == 2 and d["cpu_baseline"]["kind"] == "port"


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_gpu_equal_one_rank(device):
    """The 8-GPU launch path without an 8-GPU node: `python bench.py --gpus 8` (self-launching, one rank per "GPU") with every rank
    pinned to the one device and gloo standing in for RCCL -- 8 ranks x 12 500 grids == BASELINE configs[3]'s sharding at 1/10
    scale.  The record parses, carries 8 per-rank rates, and -- batch and actions being functions of the GLOBAL grid index -- the
    all-reduced metrics equal those of ONE rank stepping the same 100 000 grids (to the re-association of a float64 sum)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MGX_DIST_BACKEND="gloo", MGX_FORCE_LOCAL_RANK="0")
    common = ["--rows", "300", "--steps", "2", "--warmup", "1", "--chunk", "32", "--launches-per-step", "2", "--hetero-steps", "0",
              "--no-side-modes", "--no-cpu-baseline", "--prewarm", "0", "--detail", ""]
    r8 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--grids", "12500"] + common, capture_output=True,
                        text=True, timeout=1200, env=env)
    assert r8.returncode == 0, (r8.stdout + r8.stderr)[-3000:]
    d8 = _line(r8.stdout)
    assert d8["n_gpus"] == 8 and d8["config"]["grids_total"] == 100000 and d8["config"]["backend"] == "gloo"
    assert len(d8["per_rank_env_steps_per_s"]) == 8 and all(v > 0 for v in d8["per_rank_env_steps_per_s"]) and d8["value"] > 0
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--grids", "100000"] + common, capture_output=True,
                        text=True, timeout=600)
    assert r1.returncode == 0, (r1.stdout + r1.stderr)[-3000:]
    d1 = _line(r1.stdout)
    m8, m1 = d8["metrics"], d1["metrics"]
    assert m8["backend"] == "gloo"
    assert abs(m8["sum_last_reward"] - m1["sum_last_reward"]) <= 1e-12 * abs(m1["sum_last_reward"]), (m8, m1)
    assert abs(m8["mean_soc"] - m1["mean_soc"]) <= 1e-12 * abs(m1["mean_soc"]), (m8, m1)


@pytest.mark.gpu
def test_bench_two_ranks_under_torchrun(device):
    env = dict(os.environ, MGX_DIST_BACKEND="gloo", MGX_FORCE_LOCAL_RANK="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-side-modes", "--no-cpu-baseline"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["grids_total"] == 40000 and d["value"] > 0 and d["cpu_baseline"] is None


@pytest.mark.gpu
def test_two_gpus_first_contact_over_rccl(device):
    """Fires the first time a box shows two GPUs (skips on the one-GPU boxes of this pool): `python bench.py --gpus 2` launches one
    rank per GPU over RCCL (backend "nccl"), the ranks see each other, THE collective of the engine -- the metrics all-reduce --
    comes through on RCCL itself (no gloo fallback), and a short bench line carries two per-rank rates."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the multi-GPU path is covered by the gloo tests; this one needs >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "MGX_DIST_BACKEND", "MGX_FORCE_LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert d["backend"] == "nccl" and d["ranks"] == [0, 1], d
    m = d["metrics_allreduce"]
    assert m["sums"] == [3.0, 2.0] and m["ok_on_every_rank"] and m["collective_backend"] == "nccl" and m["error"] is None, m
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-side-modes"] + SMALL, capture_output=True,
                       text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["backend"] == "nccl" and len(d["per_rank_env_steps_per_s"]) == 2
    assert d["metrics"]["backend"] == "nccl", d["metrics"]
