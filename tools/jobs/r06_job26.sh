#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scalar.py -x -q -m gpu -k "joint or decimal or fft or random_grids" > gpurun_out/t26.log 2>&1; tail -3 gpurun_out/t26.log
for n in 100 200 300 400; do
  timeout 300 python tools/ab_grid.py --grid $n,$n,$n --mode porous --steps 10 --set joint_x=0 --set joint_x=1 --set joint_x=0 --set joint_x=1 2>&1 | cut -c1-330
done
