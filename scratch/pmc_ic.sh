#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ic; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_IFETCH --kernel-trace -d $O/p1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/p1.log 2>&1
tail -1 $O/p1.log | cut -c1-120
cd $R
for c in SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_IFETCH; do python tools/rocprof_pmc.py $O/p1 $c 8 2>&1 | grep "counter\|attn_cluster" | cut -c1-140; done
