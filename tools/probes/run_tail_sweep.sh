mkdir -p gpurun_out/v9
for tail in "" 6,4 7,4 6,5 7,5 8,6 5,3 7,3; do
  SATT_TAIL_FWD=$tail timeout 200 python bench.py --no-cpu-baseline --no-decode 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('forward tail [$tail]: ms/step %.3f median %.3f' % (b['ms_per_step'], b['ms_per_step_median']))" | tee -a gpurun_out/v9/sweep.txt
done
