#!/bin/bash
# GPU experiment: does a deliberate start offset between the two shard streams reproduce the good (66-68 us) rounds?
for st in 0 10 20 35 50 0 35; do
  echo "stagger $st us:"
  for rep in 1 2; do
    MGX_FORK_STAGGER_US=$st python bench.py --gpus 1 --steps 20 --warmup 5 --no-side-modes --no-cpu-baseline --hetero-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('   20 rounds: frac %.3f cadence %.2f us wall %.2f us/round kernel %.1f' % (r['frac'], r['avg_launch_us'], d['ms_per_step']*1e3, r['kernel_avg_duration_us']))"
  done
  MGX_FORK_STAGGER_US=$st python bench.py --gpus 1 --steps 512 --warmup 64 --no-side-modes --no-cpu-baseline --hetero-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('  512 rounds: frac %.3f cadence %.2f us wall %.2f us/round' % (r['frac'], r['avg_launch_us'], d['ms_per_step']*1e3))"
done
