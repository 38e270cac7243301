#!/bin/bash
python -m pytest tests/test_model_gpu.py -q -x -k "fused_lstm_input" 2>&1 | tail -5
python -m pytest tests/test_ops_gpu.py -q -x -k "lstm" 2>&1 | tail -2
for rep in 1 2; do
SATT_FUSE_XG_STEPS=0 python bench.py --no-cpu-baseline --no-decode --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fuse 0  %.3f' % d['ms_per_step'])"
SATT_FUSE_XG_STEPS=16 python bench.py --no-cpu-baseline --no-decode --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fuse 16 %.3f' % d['ms_per_step'])"
SATT_FUSE_XG_STEPS=32 python bench.py --no-cpu-baseline --no-decode --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fuse 32 %.3f' % d['ms_per_step'])"
SATT_FUSE_XG_STEPS=64 python bench.py --no-cpu-baseline --no-decode --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fuse 64 %.3f' % d['ms_per_step'])"
done
python tools/phase_marks.py 2>&1 | grep "loop fwd\|total"
