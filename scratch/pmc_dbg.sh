#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcdbg; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
mk() { cat > /tmp/run_$1.py <<PY
import faulthandler, sys, runpy, torch
faulthandler.dump_traceback_later(40, exit=True)
sys.path.insert(0, '$R')
import importlib
pkg = importlib.import_module('self-attention-tacotron_amd')
from importlib import import_module
ops = import_module('self-attention-tacotron_amd.ops'); eng = import_module('self-attention-tacotron_amd.engine')
dev = torch.device('cuda:0')
s = eng.Engine._device_streams(dev)
print('streams', s, flush=True)
main = torch.cuda.current_stream()
print('probe', [ops.streams_run_concurrently(main, x) for x in s], flush=True)
sys.argv = ['bench.py', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'] + '$2'.split()
runpy.run_path('$R/bench.py', run_name='__main__')
PY
}
mk a "--chunked-attention"; mk b ""
for v in a b; do
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/p$v -- python /tmp/run_$v.py > $O/log_$v.txt 2>&1
echo "== $v"; grep -v "^W2026\|^E2026" $O/log_$v.txt | tail -12
done
