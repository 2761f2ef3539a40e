#!/bin/bash
# Kernel-trace statistics of one extra bench configuration (bounded by its own timeout).
# usage: tools/profile_extra.sh <tag> <bench.py arguments ...>
tag=$1; shift
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- \
  python3 bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
echo "stats $tag rc=$?"
