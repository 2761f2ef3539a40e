#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fft or random_grids or decimal" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_fullsize_oracle.py -x -q -m gpu -k "decimal" 2>&1 | tail -3
for n in 480 500 600; do
  timeout 400 python tools/ab_grid.py --grid $n,$n,$n --steps 5 --set joint_x=0 --set joint_x=1 --set joint_x=0 --set joint_x=1 2>&1 | cut -c1-330
done
