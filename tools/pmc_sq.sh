#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O/p1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/p1.log 2>&1
tail -2 $O/p1.log | cut -c1-200
cd $R
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do python tools/rocprof_pmc.py $O/p1 $c 6 2>&1 | grep "counter\|attn_cluster\|gemm_kernelILi1ELi0ELb1ELb1ELb1\|lstm_cluster_bwd" | cut -c1-150; done
