#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "plan_kernel or built_for_one_plan" > gpurun_out/t33.log 2>&1; tail -3 gpurun_out/t33.log
