#!/bin/bash
# the round's last full job: whole GPU suite + smoke + the driver's bench line on the final library
cd "$(dirname "$0")/../.." || exit 1
timeout 1700 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/gpu_suite.log 2>&1
echo "suite rc=$?"; grep -E "passed|failed|error" gpurun_out/gpu_suite.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/bench_r06_final.json 2> gpurun_out/bench_r06_final.err; echo "bench rc=$?"
python3 tools/bench_brief.py gpurun_out/bench_r06_final.json 2>/dev/null | head -5
