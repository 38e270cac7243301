mkdir -p gpurun_out/v8
timeout 300 python -m pytest tests/test_flash_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/v8/tests.log
python tools/flash_time.py 2>&1 | tail -6 | tee gpurun_out/v8/flash_time.txt
for bf in 1 0; do
  SATT_FLASH_BF16=$bf timeout 200 python bench.py --no-cpu-baseline --no-decode 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('flash bf16 copies=$bf: ms/step %.3f median %.3f attn bwd launch %.3f' % (b['ms_per_step'], b['ms_per_step_median'], b['roofline']['launch_ms']))" | tee -a gpurun_out/v8/sweep.txt
done
