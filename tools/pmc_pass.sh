#!/bin/bash
# One rocprofv3 counter pass over a short bench run, bounded by its own timeout.
# usage: tools/pmc_pass.sh <tag> <n> <mixing> <counter> [<counter> ...]
# Output: gpurun_out/pmc_<tag>/ (csv).  Counter passes are kept separate from --stats runs.
tag=$1; n=$2; mix=$3; shift 3
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
rm -rf "$out"
timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out" -- \
  python3 bench.py --n "$n" --mixing "$mix" --steps 3 --warmup 1 --repeats 1 --sustain-s 0.05 --also "" --slab-members 0 --no-live-traffic \
  --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
echo "pmc $tag rc=$?"
