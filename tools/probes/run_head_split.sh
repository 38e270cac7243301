mkdir -p gpurun_out/v6
for cfg in "0 2" "2 1" "2 2" "2 3" "2 4" "0 2"; do set -- $cfg
  SATT_HEAD_SPLIT_LOW=$1 SATT_HEAD_SPLIT_RELEASE=$2 timeout 200 python bench.py --no-cpu-baseline --no-decode 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('low=$1 release=$2: ms/step %.3f median %.3f attn bwd launch %.3f' % (b['ms_per_step'], b['ms_per_step_median'], b['roofline']['launch_ms']))" | tee -a gpurun_out/v6/sweep.txt
done
SATT_HEAD_SPLIT_LOW=2 timeout 300 python -m pytest tests/test_flash_gpu.py -m gpu -x -q -k "split" 2>&1 | tail -3 | tee gpurun_out/v6/tests.log
