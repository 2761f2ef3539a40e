mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_parity.py -k "fft_forward or stage or full_run or basic or green" -x -q > gpurun_out/r06/t16.log 2>&1; tail -3 gpurun_out/r06/t16.log
for n in 100 120 200 240 300; do
  timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 10 2>&1 | cut -c1-300
done
