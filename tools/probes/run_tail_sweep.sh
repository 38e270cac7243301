mkdir -p gpurun_out/v4
for tail in 6,3 6,4 7,4 6,5 7,5 8,6; do
  timeout 200 python bench.py --no-cpu-baseline --no-decode --tail $tail 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tail $tail: ms/step %.3f median %.3f attn bwd launch %.3f' % (b['ms_per_step'], b['ms_per_step_median'], b['roofline']['launch_ms']))" | tee -a gpurun_out/v4/sweep.txt
done
