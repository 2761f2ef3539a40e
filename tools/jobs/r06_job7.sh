mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r06/gpu_suite.log 2>&1; tail -15 gpurun_out/r06/gpu_suite.log
