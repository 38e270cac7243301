mkdir -p gpurun_out/v10
for n in 1 2 3 4 1; do
  SATT_HEAD_SPLIT_CHUNKS=$n timeout 200 python bench.py --no-cpu-baseline --no-decode 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunks in suffix $n: ms/step %.3f median %.3f attn bwd launch %.3f' % (b['ms_per_step'], b['ms_per_step_median'], b['roofline']['launch_ms']))" | tee -a gpurun_out/v10/sweep.txt
done
