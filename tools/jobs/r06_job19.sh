#!/bin/bash
# kernel statistics of the passes at 128^3 (configs[1]), 200^3 and 300^3 with the final library
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
for n in 128 200 300; do
  out=gpurun_out/prof_r06b_$n
  rm -rf "$out"; mkdir -p "$out"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- \
    python3 bench.py --n $n --mixing voigt --steps 20 --warmup 3 --repeats 3 --sustain-s 0.5 --also "" --slab-members 0 \
    --no-cpu-baseline --no-live-traffic > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
  echo "n=$n stats rc=$?"
  cp $(find "$out/stats" -name "*kernel_stats.csv" | head -1) "$out/kernel_stats.csv" 2>/dev/null
  head -12 "$out/kernel_stats.csv"
done
tools/pmc_pass.sh r06b_128_fetch 128 voigt FETCH_SIZE
tools/pmc_pass.sh r06b_128_write 128 voigt WRITE_SIZE
python3 tools/traffic_csv.py gpurun_out/pmc_r06b_128_fetch gpurun_out/pmc_r06b_128_write > gpurun_out/prof_r06b_128/pmc_hbm_traffic.csv
cat gpurun_out/prof_r06b_128/pmc_hbm_traffic.csv
