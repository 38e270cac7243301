#!/bin/bash
# SQ / TCC counters of conv_bank_fwd_k (forward + input gradient launches of tools/bench_conv_bank.py), one pass per group
bash "$(dirname "$0")/pmc_kernel.sh" conv_bank "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
  "SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" -- python "$(dirname "$0")/bench_conv_bank.py" 5
