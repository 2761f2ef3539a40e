#!/bin/bash
# Collect the per-round profile set on a GPU box (each profiler call bounded by its own timeout).
# usage: tools/profile_round.sh <tag, e.g. r02_256> [n=256] [mixing=voigt]
tag=$1; n=${2:-256}; mix=${3:-voigt}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- \
  python3 bench.py --n "$n" --mixing "$mix" --steps 20 --warmup 3 --repeats 3 --sustain-s 0.5 --also "" --slab-members 0 \
  --no-cpu-baseline --no-live-traffic > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
echo "stats rc=$?"
cp $(find "$out/stats" -name "*kernel_stats.csv" | head -1) "$out/kernel_stats.csv" 2>/dev/null
tools/pmc_pass.sh ${tag}_fetch "$n" "$mix" FETCH_SIZE
tools/pmc_pass.sh ${tag}_write "$n" "$mix" WRITE_SIZE
python3 tools/traffic_csv.py gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write > "$out/pmc_hbm_traffic.csv"
