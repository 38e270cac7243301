#!/bin/bash
# final round measurements: run from the repo root on the GPU box
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $cnt --kernel-trace -d $O/pmc_$cnt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --time-all-kernels > $O/pmc_$cnt.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
cd $R
SATT_CHUNKS=1 SATT_CMAX=4 timeout 400 python tools/prof_attn.py > $O/attn_phases.txt 2>&1
SATT_TRACE_ONLY=1 SATT_TRACE=1 SATT_CHUNKS=1 SATT_CMAX=4 timeout 400 python tools/prof_attn.py 2>&1 | tail -11 > $O/attn_trace_fwd.txt
SATT_TRACE_ONLY=1 SATT_TRACE_BWD=1 SATT_CHUNKS=1 SATT_CMAX=4 timeout 400 python tools/prof_attn.py 2>&1 | tail -17 > $O/attn_trace_bwd.txt
timeout 200 python tools/phase_marks.py 2>&1 | tail -17 > $O/phase_marks.txt
timeout 200 python tools/bench_infer.py --steps 200 > $O/infer.json 2> $O/infer.err
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/gpu_tests.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
