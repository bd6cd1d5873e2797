#!/bin/bash
# config 5: ring refills as chunks inside the step launches vs whole rings ahead on the prefetch streams
mkdir -p gpurun_out
{
python tools/exp_hetero_trace.py 2048 8 float64 chunks
python tools/exp_hetero_trace.py 2048 16 float64 ahead
MGX_PREFETCH_PRIORITY=low python tools/exp_hetero_trace.py 2048 16 float64 ahead
MGX_PREFETCH_PRIORITY=high python tools/exp_hetero_trace.py 2048 16 float64 ahead
MGX_PREFETCH_PRIORITY=low python tools/exp_hetero_trace.py 2048 8 float64 ahead
MGX_WIN_GROUP=8 python tools/exp_hetero_trace.py 2048 16 float64 ahead
python tools/exp_hetero_trace.py 2048 12 float64 ahead
python tools/exp_hetero_trace.py 2048 24 float64 ahead
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/exp_fleet_refill2.txt
