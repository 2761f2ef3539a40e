#!/bin/bash
# after the split of the tile kernels into translation units of their own: parity subset, kernel statistics at 200^3 / 300^3, bench line
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scalar.py -x -q -m gpu -k "joint or decimal or fft or random_grids or estimators" > gpurun_out/t27.log 2>&1; tail -2 gpurun_out/t27.log
for n in 200 300; do
  out=gpurun_out/prof_r06c_$n
  rm -rf "$out"; mkdir -p "$out"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- \
    python3 bench.py --n $n --mixing voigt --steps 20 --warmup 3 --repeats 3 --sustain-s 0.5 --also "" --slab-members 0 \
    --no-cpu-baseline --no-live-traffic > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
  echo "n=$n stats rc=$?"
  cp $(find "$out/stats" -name "*kernel_stats.csv" | head -1) "$out/kernel_stats.csv" 2>/dev/null
  head -8 "$out/kernel_stats.csv" | cut -c1-200
done
tools/pmc_pass.sh r06c_200_fetch 200 voigt FETCH_SIZE
tools/pmc_pass.sh r06c_200_write 200 voigt WRITE_SIZE
python3 tools/traffic_csv.py gpurun_out/pmc_r06c_200_fetch gpurun_out/pmc_r06c_200_write > gpurun_out/prof_r06c_200/pmc_hbm_traffic.csv
cat gpurun_out/prof_r06c_200/pmc_hbm_traffic.csv | grep -i "smooth\|u_tile"
timeout 900 python bench.py > gpurun_out/bench_r06_v5.json 2> gpurun_out/bench_r06_v5.err; echo "bench rc=$?"
python3 tools/bench_brief.py gpurun_out/bench_r06_v5.json 2>/dev/null | head -30
