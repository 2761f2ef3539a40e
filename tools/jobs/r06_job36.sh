#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
timeout 900 python -m pytest tests/test_gpu_scalar.py -x -q -m gpu -k "plan or decimal" > gpurun_out/t36.log 2>&1; tail -3 gpurun_out/t36.log
for n in 200 300 400; do
  timeout 300 python tools/ab_grid.py --grid $n,$n,$n --mode porous --steps 10 --set tile_plans=0 --set tile_plans=1 --set tile_plans=0 --set tile_plans=1 2>&1 | cut -c1-300
done
