#!/bin/bash
# config 5: ring refills as chunks inside the step launches vs whole rings ahead on the prefetch streams, ring depths, row types
# (profiles/r02/exp_fleet_refill_ahead.txt; the stream-priority / LDS-cap variants in that log were experiment builds)
mkdir -p gpurun_out
{
for mode in chunks ahead; do
  for K in 8 16; do
    python tools/exp_hetero_trace.py 2048 $K float64 $mode
  done
done
python tools/exp_hetero_trace.py 2048 4 float64 ahead
python tools/exp_hetero_trace.py 2048 32 float64 ahead
python tools/exp_hetero_trace.py 2048 8 float32 chunks
python tools/exp_hetero_trace.py 2048 16 float32 ahead
PER=33344 python tools/exp_hetero_trace.py 2048 16 float64 ahead
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/exp_fleet_refill.txt
