mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_parity.py -k "estimator or chunked or staged" -x -q > gpurun_out/r06/t3.log 2>&1; tail -15 gpurun_out/r06/t3.log
timeout 1500 python -m pytest tests/test_gpu_slab.py -k "estimators or cg_matches" -x -q > gpurun_out/r06/t3b.log 2>&1; tail -15 gpurun_out/r06/t3b.log
timeout 300 python tools/transfer_bench.py 256 > gpurun_out/r06/transfer_bench2.txt 2>&1; cat gpurun_out/r06/transfer_bench2.txt
