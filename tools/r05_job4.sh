#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
o=gpurun_out/r05_job4; mkdir -p $o
timeout 600 python3 bench.py --mode viscosity --also "" --no-cpu-baseline --slab-members 0 > $o/bench_viscosity.json 2> $o/bench_viscosity.err; echo "viscosity rc=$?"
timeout 900 python3 bench.py --gpus 2 --dist-backend nccl-one-gpu --steps 10 --repeats 3 --also-slab "" > $o/bench_gpus2.json 2> $o/bench_gpus2.err; echo "gpus2 rc=$?"
timeout 900 python3 bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "default rc=$?"
timeout 1200 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=40 > $o/pytest_durations.log 2>&1; echo "pytest rc=$?"
tail -60 $o/pytest_durations.log
