#!/bin/bash
# round 5: the per-round profile set (rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE passes) for the bench workload and
# the north-star target configuration, then a fuzz soak on the final library and the suite's per-file times
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tools/profile_round.sh r05_256 256 voigt
tools/profile_round.sh r05_512lam 512 laminate
FG_FUZZ_SEEDS=${1:-600} timeout 2400 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r05_fuzz_soak.log 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r05_fuzz_soak.log
