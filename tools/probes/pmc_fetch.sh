R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fetch; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $cnt --kernel-trace -d $O/$cnt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode > $O/$cnt.log 2>&1
done
cd $R
for cnt in FETCH_SIZE WRITE_SIZE; do python tools/rocprof_pmc.py $O/$cnt $cnt 12 | cut -c1-150; done
