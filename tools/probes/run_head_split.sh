mkdir -p gpurun_out/v5
timeout 300 python -m pytest tests/test_flash_gpu.py -m gpu -x -q -k "split" 2>&1 | tail -3 | tee gpurun_out/v5/tests.log
for low in 0 1 0 1; do
  SATT_HEAD_SPLIT_LOW=$low timeout 200 python bench.py --no-cpu-baseline --no-decode 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('low tiles on side stream=$low: ms/step %.3f median %.3f attn bwd launch %.3f' % (b['ms_per_step'], b['ms_per_step_median'], b['roofline']['launch_ms']))" | tee -a gpurun_out/v5/sweep.txt
done
for m in vctk; do for low in 0 1; do
  SATT_HEAD_SPLIT_LOW=$low timeout 200 python bench.py --no-cpu-baseline --no-decode --model $m 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$m low=$low: ms/step %.3f median %.3f' % (b['ms_per_step'], b['ms_per_step_median']))" | tee -a gpurun_out/v5/sweep.txt
done; done
timeout 100 python tools/phase_marks.py 2>&1 | tail -19 > gpurun_out/v5/phases.txt; cat gpurun_out/v5/phases.txt
