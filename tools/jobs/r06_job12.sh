mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r06/gpu_suite_final.log 2>&1; tail -25 gpurun_out/r06/gpu_suite_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r06/bench_default_v2.json 2> gpurun_out/r06/bench_default_v2.err; echo "bench rc $?"; tail -c 300 gpurun_out/r06/bench_default_v2.json
