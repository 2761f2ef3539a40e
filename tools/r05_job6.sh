#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out/r05_job6; mkdir -p $o
cat /proc/loadavg
( time timeout 900 python3 -m pytest tests/test_gpu_distributed.py -m gpu -x -q -p no:cacheprovider --durations=8 ) > $o/dist_direct.log 2>&1; tail -14 $o/dist_direct.log
cat /proc/loadavg
( time FG_TEST_LAUNCHER=torchrun timeout 900 python3 -m pytest tests/test_gpu_distributed.py -m gpu -x -q -p no:cacheprovider --durations=8 ) > $o/dist_torchrun.log 2>&1; tail -14 $o/dist_torchrun.log
cat /proc/loadavg; cat /sys/fs/cgroup/cpu.stat | grep thrott
