mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_parity.py -k "fft_forward or stage or full_run or basic or green" -x -q > gpurun_out/r06/t8.log 2>&1; tail -5 gpurun_out/r06/t8.log
timeout 1500 python -m pytest tests/test_gpu_scalar.py tests/test_gpu_slab.py -x -q --durations=8 > gpurun_out/r06/t8b.log 2>&1; tail -14 gpurun_out/r06/t8b.log
timeout 900 python -m pytest tests/test_gpu_fullsize_oracle.py -k decimal -x -q > gpurun_out/r06/t8c.log 2>&1; tail -3 gpurun_out/r06/t8c.log
O=gpurun_out/r06/grid_size_landscape_smooth_v3.jsonl; : > $O
for n in 100 120 200 240 300 400 480 500; do
  for fx in 1 0; do
  timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 5 --set fuse_x=$fx >> $O 2>&1
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r06/grid_size_landscape_smooth_v3.jsonl'):
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    n=d['grid'][0]; print(d['variant'], n, d['it_s'], "Gvox/s %.2f"%(n**3*d['it_s']/1e9), d['stages_us'])
PY
