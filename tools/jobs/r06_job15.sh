mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_parity.py -k "fft_forward or stage or full_run or basic or green" -x -q > gpurun_out/r06/t15.log 2>&1; tail -3 gpurun_out/r06/t15.log
O=gpurun_out/r06/ab_wide_tiles.jsonl; : > $O
for n in 100 120 144 200 240 250; do
  for w in 0 1 0 1; do
  FG_SMOOTH_WIDE=$w timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 10 2>&1 | sed "s/^{/{\"wide\": $w, /" >> $O
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r06/ab_wide_tiles.jsonl'):
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    n=d['grid'][0]; print(d['wide'], n, d['it_s'], d['stages_us'])
PY
