#!/bin/bash
# SQ counters of the fused attention and small-row GEMM kernels of one train step (one --pmc pass, kernel trace only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sqf; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O/p1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode > $O/p1.log 2>&1
cd $R
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do python tools/rocprof_pmc.py $O/p1 $c 60 2>&1 | grep "counter\|flash\|gemm_rows\|small_attn" | cut -c1-150; done | tee $O/summary.txt
