"""The C ABI from a consumer that is neither Python nor torch: tests/c_abi/demo.cpp (HIP runtime + include/mgx.h) is compiled
against libmgx.so and the CPU oracle and must report zero mismatches."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "demo.cpp")
EXE = os.path.join(ROOT, "tests", "c_abi", "_build", "demo")


def build_demo(SRC=SRC, EXE=EXE):
    from oracle import oracle as orc
    from pymgrid_amd import _lib
    _lib.build()
    orc.build()
    deps = [SRC, _lib.LIB_PATH, orc._LIB_PATH, os.path.join(ROOT, "include", "mgx.h")]
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(d) for d in deps):
        return EXE
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", SRC, "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "oracle"), "-L" + os.path.dirname(_lib.LIB_PATH), "-lmgx",
                    "-L" + os.path.dirname(orc._LIB_PATH), "-lmgx_oracle",
                    "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath," + os.path.dirname(orc._LIB_PATH), "-o", EXE],
                   check=True)
    return EXE


def test_c_abi_consumer_builds():
    """hipcc compiles and links the consumer against the shipped header and library (no GPU needed)."""
    assert os.path.exists(build_demo())


@pytest.mark.gpu
def test_c_abi_consumer_matches_the_oracle():
    res = subprocess.run([build_demo()], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "0 mismatches" in res.stdout


# SURVEY 8(b)'s list beyond create / reset / step / step_k: reset with observations, observe, expand_discrete, check_discrete,
# step_discrete, metrics, the bound Gym step (mgx_env_*), fleet_step, generate_columns -- tests/c_abi/demo2.cpp
SRC2 = os.path.join(ROOT, "tests", "c_abi", "demo2.cpp")
EXE2 = os.path.join(ROOT, "tests", "c_abi", "_build", "demo2")


def test_c_abi_consumer_2_builds():
    assert os.path.exists(build_demo(SRC2, EXE2))


@pytest.mark.gpu
def test_c_abi_consumer_2_matches_the_oracle():
    res = subprocess.run([build_demo(SRC2, EXE2)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert ": 0 mismatches" in res.stdout


# the general path (several modules of a kind) from C++: mgx_step, mgx_step_k, mgx_expand_lists, mgx_step_lists (ABI minor 2),
# mgx_rollout_lists against the oracle's multi-instance restatement -- tests/c_abi/demo3.cpp
SRC3 = os.path.join(ROOT, "tests", "c_abi", "demo3.cpp")
EXE3 = os.path.join(ROOT, "tests", "c_abi", "_build", "demo3")


def test_c_abi_consumer_3_builds():
    assert os.path.exists(build_demo(SRC3, EXE3))


@pytest.mark.gpu
def test_c_abi_consumer_3_matches_the_oracle():
    res = subprocess.run([build_demo(SRC3, EXE3)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert ": 0 mismatches" in res.stdout
