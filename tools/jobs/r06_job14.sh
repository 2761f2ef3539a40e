mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_parity.py -k "fft_forward or stage or full_run or basic or green" -x -q > gpurun_out/r06/t14.log 2>&1; tail -3 gpurun_out/r06/t14.log
O=gpurun_out/r06/grid_size_landscape_rounds.jsonl; : > $O
for n in 100 120 200 240 300 360 400 480 500 600; do
  timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 5 >> $O 2>&1
done
python - <<'PY'
import json
for l in open('gpurun_out/r06/grid_size_landscape_rounds.jsonl'):
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    n=d['grid'][0]; print(n, d['it_s'], "Gvox/s %.2f"%(n**3*d['it_s']/1e9), d['stages_us'])
PY
