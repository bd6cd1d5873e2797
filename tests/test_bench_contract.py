"""bench.py contract on a GPU box: the one-rank JSON line, and the N > 1 launch path (torchrun, one rank per GPU) exercised
with two ranks on the single GPU through the gloo backend (MGX_DIST_BACKEND / MGX_FORCE_LOCAL_RANK test overrides) -- the
real multi-GPU runs over RCCL are the driver's."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--grids", "20000", "--rows", "700", "--steps", "192", "--warmup", "64", "--hetero-steps", "16", "--cpu-seconds", "1"]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"}


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_one_rank_json_line(device):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 192 and d["value"] > 0
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["roofline"]["bound"] == "hbm"
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-12
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
    assert "workload" in d["config"] and d["scaling"] == "weak" and d["higher_is_better"] is True


@pytest.mark.gpu
def test_bench_two_ranks_launch_path(device):
    env = dict(os.environ, MGX_DIST_BACKEND="gloo", MGX_FORCE_LOCAL_RANK="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["grids_total"] == 40000 and d["value"] > 0 and d["cpu_baseline"] is None
