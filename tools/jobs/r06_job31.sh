#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
for rep in 1 2; do
  echo -n "generic "; timeout 300 python tools/ab_grid.py --grid 200,200,200 --steps 10 2>&1 | cut -c1-300
  echo -n "plan    "; FG_XPLAN=1 timeout 300 python tools/ab_grid.py --grid 200,200,200 --steps 10 2>&1 | cut -c1-300
done
FG_XPLAN=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "joint and 200" 2>&1 | tail -2
