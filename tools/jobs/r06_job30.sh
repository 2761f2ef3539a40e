#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
for n in 400 300; do
  tools/pmc_pass.sh sq3_$n $n voigt SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS
  echo "== $n"; python3 tools/pmc_summary.py gpurun_out/pmc_sq3_$n | grep -E "k_smooth"
done
