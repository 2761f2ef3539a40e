mkdir -p gpurun_out/r06
O=gpurun_out/r06/grid_size_landscape.jsonl; : > $O
for n in 64 96 100 120 128 144 160 192 200 224 240 256 288 300 320 384 400 448 480 500; do
  timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 5 >> $O 2>&1
done
cat $O | cut -c1-400
python tools/setup_cost.py 128 > gpurun_out/r06/setup_cost_128.txt 2>&1; tail -12 gpurun_out/r06/setup_cost_128.txt
python tools/concurrent_cases_probe.py 128 > gpurun_out/r06/concurrent_128.txt 2>&1; tail -5 gpurun_out/r06/concurrent_128.txt
python tools/ceff_bench.py 128 > gpurun_out/r06/ceff_128.txt 2>&1; tail -3 gpurun_out/r06/ceff_128.txt
timeout 900 python -m pytest tests/test_gpu_viscosity.py -k "nunan" -x -q > gpurun_out/r06/t4.log 2>&1; tail -5 gpurun_out/r06/t4.log
timeout 900 python -m pytest tests/test_gpu_voxelize.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r06/t4b.log 2>&1; tail -5 gpurun_out/r06/t4b.log
