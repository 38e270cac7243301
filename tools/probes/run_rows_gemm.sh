mkdir -p gpurun_out/v3
timeout 300 python -m pytest tests/test_gemm_tile_gpu.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/v3/gemm_tests.log
for off in 1 0; do
  SATT_NO_ROWS_GEMM=$off timeout 200 python bench.py --no-cpu-baseline --no-decode 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows gemm off=$off: ms/step %.3f median %.3f attn bwd launch %.3f' % (b['ms_per_step'], b['ms_per_step_median'], b['roofline']['launch_ms']))" | tee -a gpurun_out/v3/sweep.txt
done
