#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
for n in 112 384 576 640 768 896; do
  echo -n "mixed "; FG_FORCE_MIXED=1 timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 5 2>&1 | cut -c1-400
  echo -n "tiles "; timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 5 2>&1 | cut -c1-400
done
