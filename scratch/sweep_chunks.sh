#!/bin/bash
for c in 2 4 8 16; do
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --chunks $c 2>/dev/null > /tmp/b_$c.json
  python - "$c" <<'PY'
import json, sys
c = sys.argv[1]
d = json.load(open("/tmp/b_%s.json" % c))
print("chunks", c, round(d["ms_per_step"], 2), round(d["value"]), d["kernel_ms_per_step"])
PY
done
