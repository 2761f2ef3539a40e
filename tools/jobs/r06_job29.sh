#!/bin/bash
# old / new library alternating in one job: tile kernels' index arithmetic without run-time integer divisions
cd "$(dirname "$0")/../.." || exit 1
cp fibergen_amd/libfibergen_amd.so /tmp/new.so; cp tools/build/libfibergen_amd_old.so /tmp/old.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fft or random_grids or decimal or joint" > gpurun_out/t29.log 2>&1; tail -2 gpurun_out/t29.log
for rep in 1 2; do for v in old new; do
  cp /tmp/$v.so fibergen_amd/libfibergen_amd.so
  for n in 100 200 300 400 500; do echo -n "$v "; timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 10 2>&1 | cut -c1-300; done
done; done | tee gpurun_out/ab_nodiv.jsonl
cp /tmp/new.so fibergen_amd/libfibergen_amd.so
