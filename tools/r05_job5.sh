#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
o=gpurun_out/r05_job5; mkdir -p $o
{ echo "== nproc"; nproc; echo "== cpu.max"; cat /sys/fs/cgroup/cpu.max 2>&1; echo "== v1 quota"; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>&1; echo "== cpu.stat"; cat /sys/fs/cgroup/cpu.stat /sys/fs/cgroup/cpu/cpu.stat 2>&1 | head -12; echo "== cpuset"; cat /sys/fs/cgroup/cpuset.cpus.effective /sys/fs/cgroup/cpuset/cpuset.cpus 2>&1; echo "== affinity"; python3 -c "import os;print(len(os.sched_getaffinity(0)))"; echo "== lscpu"; lscpu | head -25; echo "== numa"; numactl -H 2>&1 | head -12; echo "== mem"; free -g | head -3; echo "== loadavg"; cat /proc/loadavg; } > $o/cpu_info.txt 2>&1
timeout 900 python3 bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "default rc=$?"
tail -3 $o/bench_default.err
timeout 1200 python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_cg_fused.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider > $o/pytest_subset.log 2>&1; echo "pytest rc=$?"
tail -3 $o/pytest_subset.log
cat $o/cpu_info.txt
