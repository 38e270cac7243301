for rep in 1 2 3; do
for v in sumsfirst rowsfirst rowsfirst_newdw; do
  SATT_LIB_PATH=tools/probes/libsatt_$v.so timeout 200 python bench.py --no-cpu-baseline --no-decode --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), d['kernel_ms_per_step'])"
done
done
