#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scalar.py -x -q -m gpu -k "plan or joint or decimal or fft or random_grids" > gpurun_out/t32.log 2>&1; tail -3 gpurun_out/t32.log
for n in 100 120 200 240 300 360 400 480 500; do
  timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 10 --set tile_plans=0 --set tile_plans=1 --set tile_plans=0 --set tile_plans=1 2>&1 | cut -c1-330
done | tee gpurun_out/ab_tile_plans.jsonl
