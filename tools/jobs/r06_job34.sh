#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
for rep in 1 2; do for n in 300 360 400; do
  echo -n "three-pass "; timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 10 2>&1 | cut -c1-300
  echo -n "two-pass   "; FG_PLAN_HI=20 timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 10 2>&1 | cut -c1-300
done; done
