for rep in 1 2 3; do
for v in foldlds foldagpr; do
  SATT_LIB_PATH=tools/probes/libsatt_$v.so timeout 200 python bench.py --no-cpu-baseline --no-decode --steps 40 --warmup 5 --time-all-kernels 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), {k:v for k,v in d['kernel_ms_per_step'].items() if 'attn_rnn' in k})"
done
done
