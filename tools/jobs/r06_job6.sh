mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_parity.py -k "fft_forward or stage or full_run or basic" -x -q > gpurun_out/r06/t6.log 2>&1; tail -8 gpurun_out/r06/t6.log
O=gpurun_out/r06/grid_size_landscape_smooth_v2.jsonl; : > $O
for mode in 1 2; do
for n in 96 100 120 144 160 192 200 224 240 288 300 320 384 400 448 480 500; do
  FG_FFT_SMOOTH=$mode timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 5 2>&1 | sed "s/^{/{\"smooth\": $mode, /" >> $O
done
done
python - <<'PY'
import json
for l in open('gpurun_out/r06/grid_size_landscape_smooth_v2.jsonl'):
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    n=d['grid'][0]; print(d['smooth'], n, d['it_s'], "Gvox/s %.2f"%(n**3*d['it_s']/1e9), d['stages_us'])
PY
