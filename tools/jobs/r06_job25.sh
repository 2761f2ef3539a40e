#!/bin/bash
# the whole GPU suite + smoke with the library as built
cd "$(dirname "$0")/../.." || exit 1
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/gpu_suite.log 2>&1
echo "suite rc=$?"
grep -E "passed|failed|error" gpurun_out/gpu_suite.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
