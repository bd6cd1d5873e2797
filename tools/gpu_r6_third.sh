#!/bin/bash
# Round 6, third GPU call: kernel arguments in DEVICE memory (HIP_FORCE_DEV_KERNARG=1: the runtime writes the kernarg block over the
# BAR instead of leaving it in host memory, where the kernel's first scalar loads pay a PCIe round trip) x one / two launch chains.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd "$REPO"
for KA in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$KA" | tee -a "$OUT/exp_dev_kernarg.txt"
  HIP_FORCE_DEV_KERNARG=$KA timeout 600 python tools/exp_r6_two_chains.py 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/exp_dev_kernarg.txt"
  HIP_FORCE_DEV_KERNARG=$KA timeout 120 tools/bin/launchbench2 2>&1 | tee -a "$OUT/exp_dev_kernarg.txt"
done
