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
SMALL = ["--grids", "20000", "--rows", "700", "--steps", "6", "--warmup", "2", "--chunk", "32", "--hetero-steps", "16",
         "--cpu-seconds", "1", "--prewarm", "0.05"]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"}


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def gp_rows_ok(d):
    """the general path's Gym step with whole rows off prefetched rings (round 4)"""
    g = d["general_path_2g2b1grid"].get("gym_steps_rows_h24")
    return g is not None and g["value"] > 0 and g["roofline"]["frac"] > 0 and g["obs_dim"] == 162


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
def test_bench_one_rank_json_line(device):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["value"] > 0
    rf = d["roofline"]
    assert set(rf) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and rf["bound"] == "hbm"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["launches"] == 6 and rf["series"] == "factorised" and "frac_wall" in rf     # every timed round is one full-K launch
    assert rf["concurrent_streams"] == 2                                                  # ... per shard stream
    assert rf["kernel"] == "step_k_kernel<3,4,double,false,true>" and abs(rf["bytes_per_env_step"] - (158 / 32 + 40)) < 1e-9
    assert abs(d["value"] - 20000 * 6 * 32 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
    assert "workload" in d["config"] and d["scaling"] == "weak" and d["higher_is_better"] is True
    for k in ("float64_rows", "float32_rows", "float64_rows_colmajor", "float32_rows_colmajor", "float64_views", "float32_views"):
        assert set(d["hetero_h24_gym_steps"][k]["roofline"]) >= {"bound", "achieved", "peak", "frac", "frac_wall", "traffic"}, k
    assert {"fused_launches_one_stream", "fused_launches_materialised", "fused_launches_materialised_one_stream", "rbc_rollout_materialised",
            "single_step_launches_one_call", "rbc_rollout_on_device", "single_step_launches_python_loop"} <= set(d["other"])
    assert all("error" not in v for v in d["other"].values())
    assert d["other"]["fused_launches_materialised"]["roofline"]["concurrent_streams"] == 2
    # round 4: the spread of the timed rounds, which kernels ran, the issue-side roofline block, the general path and the server
    ru = rf["round_us"]
    assert ru["n"] == 6 * rf["concurrent_streams"] and ru["min"] <= ru["median"] <= ru["max"] and ru["min"] <= ru["mean"] <= ru["max"]
    assert gp_rows_ok(d)
    from pymgrid_amd import _lib
    assert d["csrc_hash"] == _lib.source_hash() and d["roofline_valu"]["bound"] == "valu"
    assert rf["traffic"] is None or "STALE" not in str(rf["traffic_source"])          # a stale counter file is never quoted
    gp = d["general_path_2g2b1grid"]
    assert "error" not in gp and gp["single_steps"]["roofline"]["bytes_per_env_step"] == 8 * (7 + 27 + 6 + 2 * 2 + 2) + 2 * 8 + 8
    assert gp["k_step_launches"]["value"] > 0 and d["resident_step_server"]["steps"] == 600


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
    assert d["metrics_allreduce"]["collective_backend"] == "nccl", d["metrics_allreduce"]
