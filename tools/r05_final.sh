#!/bin/bash
# round 5: the validation set of the final state (one job): smoke, the full GPU suite, the driver's bench command, the N > 1 line
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
o=gpurun_out/r05_final; mkdir -p $o
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $o/smoke.log
( time timeout 1500 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $o/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $o/pytest_gpu.log | tail -3
timeout 900 python3 bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "default rc=$?"
timeout 1200 python3 bench.py --gpus 2 --dist-backend nccl-one-gpu --steps 10 --repeats 3 > $o/bench_gpus2.json 2> $o/bench_gpus2.err; echo "gpus2 rc=$?"; tail -3 $o/bench_gpus2.err
