#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slab.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_fg.py -x -q -m gpu 2>&1 | tail -4
for n in 100 120 200 240 300 360 400 480 500 600 75 125 225; do
  timeout 400 python tools/ab_grid.py --grid $n,$n,$n --steps 10 2>&1 | cut -c1-300
done | tee gpurun_out/landscape_joint.jsonl
