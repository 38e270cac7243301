# kernel timeline of one train step around the head backward (rocprofv3 kernel trace -> tools/step_timeline.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-decode > $O/trace.log 2>&1 < /dev/null
cd $R
python tools/step_timeline.py $O/trace 4 > $O/timeline.txt 2>&1
grep -n "loss_fused\|flash_bwd\|attn_cluster_bwd" $O/timeline.txt | head
