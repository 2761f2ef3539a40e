#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_suite2.log 2>&1
echo "suite rc=$?"; grep -E "passed|failed|error" gpurun_out/gpu_suite2.log | tail -3
for n in 96 112 144 160 192 224 288 320 384 448 576 640 768; do
  timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 10 2>&1 | cut -c1-330
done | tee gpurun_out/landscape_p2k.jsonl
