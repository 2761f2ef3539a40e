#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slab.py -x -q -m gpu -k "fft_forward_inverse or basic_scheme or estimators or slab" > gpurun_out/t41.log 2>&1; tail -2 gpurun_out/t41.log
for n in 384 320; do timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 10 2>&1 | cut -c1-330; done
timeout 300 python tools/ab_grid.py --grid 384,200,320 --steps 10 2>&1 | cut -c1-330
