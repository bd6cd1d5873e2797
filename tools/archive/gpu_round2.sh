#!/bin/bash
# Run ON THE GPU BOX (through gpurun): GPU test suite + the driver's bench command, results under gpurun_out/r02/.
mkdir -p gpurun_out/r02
python -m pytest tests -x -q -m gpu 2>&1 | tail -15
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02/bench_driver_cmd.json 2> gpurun_out/r02/bench_driver_cmd.err
tail -c 2000 gpurun_out/r02/bench_driver_cmd.err
python tools/show_bench.py gpurun_out/r02/bench_driver_cmd.json
