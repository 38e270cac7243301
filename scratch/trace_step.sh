#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tr; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
cd $R; python tools/rocprof_timeline.py $O/trace 3 > $O/timeline.txt 2>&1; tail -1 $O/timeline.txt
