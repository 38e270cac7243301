#!/bin/bash
# tools/ab_bench.sh <libA.so> <libB.so> [...]: bench.py with each library in turn, twice over (same box, interleaved: boxes and
# clocks differ between gpurun calls, so only numbers of ONE call compare).  Prints ms/step and the attention-kernel launch times.
for rep in 1 2; do
  for lib in "$@"; do
    SATT_LIB_PATH=$lib python bench.py --no-cpu-baseline --no-decode --steps 40 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %.3f ms/step  %s' % ('$lib', d['ms_per_step'], d['kernel_ms_per_step']))"
  done
done
