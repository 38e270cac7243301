timeout 300 python -m pytest tests/test_flash_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 100 python tools/flash_time.py 2>&1 | tail -6
for rep in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --no-decode --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['kernel_ms_per_step'])"
done
timeout 100 python tools/phase_marks.py 2>&1 | grep "head\|loss"
