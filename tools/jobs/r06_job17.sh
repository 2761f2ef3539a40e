mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scalar.py -k "fft_forward or stage or full_run or basic or green or scalar or heat or porous" -x -q > gpurun_out/r06/t17.log 2>&1; tail -3 gpurun_out/r06/t17.log
timeout 900 python -m pytest tests/test_gpu_fullsize_oracle.py -k decimal -x -q 2>&1 | tail -2
for n in 100 200 300; do timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 10 2>&1 | cut -c1-300; done
