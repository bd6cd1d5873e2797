#!/bin/bash
# Run ON THE GPU BOX (through gpurun): new-ABI tests + the driver's bench command, results under gpurun_out/r02/.
mkdir -p gpurun_out/r02
python -m pytest tests/test_abi_v3.py tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -30
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02/bench_driver_cmd.json 2> gpurun_out/r02/bench_driver_cmd.err
tail -c 3000 gpurun_out/r02/bench_driver_cmd.err
python tools/show_bench.py gpurun_out/r02/bench_driver_cmd.json
