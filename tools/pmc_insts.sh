#!/bin/bash
# Dynamic instruction counts of the recurrent kernels (SQ_INSTS_* per dispatch) -> per wave and decoder step, the input of the
# issue-floor table profiles/r03_attn_issue_floor.txt (tools/issue_floor.py).  Run on the GPU box: bash tools/pmc_insts.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/insts; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES --kernel-trace -d $O/p1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode > $O/p1.log 2>&1
tail -2 $O/p1.log | cut -c1-200
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace -d $O/p2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode > $O/p2.log 2>&1
cd $R
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES; do python tools/rocprof_pmc.py $O/p1 $c 40 2>&1 | grep "counter\|attn_cluster\|lstm_cluster\|lstm_bwd_mfma\|lstm_fwd_mfma\|attn_param" | cut -c1-160; done > $O/insts.txt
for c in SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY; do python tools/rocprof_pmc.py $O/p2 $c 40 2>&1 | grep "counter\|attn_cluster\|lstm_cluster\|lstm_bwd_mfma\|lstm_fwd_mfma\|attn_param" | cut -c1-160; done >> $O/insts.txt
cat $O/insts.txt
