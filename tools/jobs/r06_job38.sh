#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
for n in 96 144 160 192 224 288 320 384 448; do
  echo -n "mixed "; timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 10 2>&1 | cut -c1-330
  echo -n "tiles "; FG_SMOOTH_MIXED=1 timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 10 2>&1 | cut -c1-330
done | tee gpurun_out/ab_mixed_vs_tiles.jsonl
FG_SMOOTH_MIXED=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fft_forward_inverse or basic_scheme" 2>&1 | tail -2
