#!/bin/bash
# where the tile kernels' time goes: SQ counters at 400^3 (tile kernels) beside 256^3 (power-of-two kernels)
cd "$(dirname "$0")/../.." || exit 1
for n in 400 256; do
  tools/pmc_pass.sh sq1_$n $n voigt SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
  tools/pmc_pass.sh sq2_$n $n voigt SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
  tools/pmc_pass.sh sq3_$n $n voigt SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM
  for t in sq1 sq2 sq3; do echo "== $t $n"; python3 tools/pmc_summary.py gpurun_out/pmc_${t}_$n | grep -E "k_smooth|k_xfused|k_strided|k_zpass|k_u_tile" ; done
done 2>&1 | tee gpurun_out/sq_counters.txt
