for cfg in "6 3,4" "6 4,8" "7 4,8" "7 5,16" "8 5,16" "6 3,8" "7 4,16"; do
  set -- $cfg
  timeout 200 python bench.py --no-cpu-baseline --no-decode --steps 40 --warmup 5 --chunks $1 --tail $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chunks $1 tail $2', round(d['ms_per_step'],3), d['kernel_ms_per_step'])"
done
