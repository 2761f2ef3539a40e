#!/bin/bash
# Collect the per-round profile set on a GPU box (each profiler call bounded by its own timeout).
# usage: tools/profile_round.sh <tag, e.g. r01b> [n=512]
tag=$1; n=${2:-512}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- \
  python3 bench.py --n "$n" --steps 20 --warmup 3 --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
echo "stats rc=$?"
tools/pmc_pass.sh ${tag}_fetch "$n" FETCH_SIZE
tools/pmc_pass.sh ${tag}_write "$n" WRITE_SIZE
tools/pmc_pass.sh ${tag}_l2 "$n" TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write gpurun_out/pmc_${tag}_l2 > "$out/pmc_summary.txt"
