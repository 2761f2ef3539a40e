mkdir -p gpurun_out/r06
O=gpurun_out/r06/ab_smooth_occ3.jsonl; : > $O
for n in 200 300 400 500; do
  timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 5 2>&1 | sed "s/^{/{\"occ3\": 0, /" >> $O
  FG_SMOOTH_OCC3=1 timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 5 2>&1 | sed "s/^{/{\"occ3\": 1, /" >> $O
done
cut -c1-330 $O
timeout 300 python tools/transfer_bench.py 256 2>&1 | tail -4 | cut -c1-400
