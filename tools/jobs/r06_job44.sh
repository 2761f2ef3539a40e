#!/bin/bash
# soak on the round's final library: the randomised combinations with more seeds (tile-kernel lengths 3 000, the whole fuzz file 1 000)
cd "$(dirname "$0")/../.." || exit 1
FG_FUZZ_SEEDS=3000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "tile_kernel_lengths" > gpurun_out/soak_tile.log 2>&1; grep -E "passed|failed" gpurun_out/soak_tile.log | tail -2
FG_FUZZ_SEEDS=1000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "not tile_kernel_lengths" > gpurun_out/soak_all.log 2>&1; grep -E "passed|failed" gpurun_out/soak_all.log | tail -2
