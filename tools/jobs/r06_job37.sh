#!/bin/bash
# final library of the round: whole GPU suite + smoke, kernel statistics and counters at 200^3 / 400^3, the driver's bench line, landscape
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/gpu_suite.log 2>&1
echo "suite rc=$?"; grep -E "passed|failed|error" gpurun_out/gpu_suite.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for n in 200 400; do
  out=gpurun_out/prof_r06d_$n
  rm -rf "$out"; mkdir -p "$out"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- \
    python3 bench.py --n $n --mixing voigt --steps 20 --warmup 3 --repeats 3 --sustain-s 0.5 --also "" --slab-members 0 \
    --no-cpu-baseline --no-live-traffic > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
  echo "n=$n stats rc=$?"
  cp $(find "$out/stats" -name "*kernel_stats.csv" | head -1) "$out/kernel_stats.csv" 2>/dev/null
  head -7 "$out/kernel_stats.csv" | cut -c1-220
done
tools/pmc_pass.sh r06d_200_fetch 200 voigt FETCH_SIZE
tools/pmc_pass.sh r06d_200_write 200 voigt WRITE_SIZE
python3 tools/traffic_csv.py gpurun_out/pmc_r06d_200_fetch gpurun_out/pmc_r06d_200_write > gpurun_out/prof_r06d_200/pmc_hbm_traffic.csv
grep -i "smooth" gpurun_out/prof_r06d_200/pmc_hbm_traffic.csv
timeout 900 python bench.py > gpurun_out/bench_r06_v6.json 2> gpurun_out/bench_r06_v6.err; echo "bench rc=$?"
python3 tools/bench_brief.py gpurun_out/bench_r06_v6.json 2>/dev/null | head -5
for n in 100 120 150 180 200 240 250 300 360 400 480 500 600 75 125 225; do
  timeout 400 python tools/ab_grid.py --grid $n,$n,$n --steps 10 2>&1 | cut -c1-300
done > gpurun_out/landscape_plans.jsonl; wc -l gpurun_out/landscape_plans.jsonl
