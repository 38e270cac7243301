#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $cnt --kernel-trace -d $O/pmc_$cnt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$cnt.log 2>&1
done
cd $R
timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -q -k "single_launch" 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline 2>&1 | cut -c1-260
