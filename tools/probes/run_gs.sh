#!/bin/bash
python -m pytest tests/test_model_gpu.py tests/test_inference_gpu.py -q -x 2>&1 | tail -3
python -m pytest tests/test_pinned_gpu.py -q -x 2>&1 | tail -2
bash tools/ab_bench.sh tools/probes/libsatt_base.so self-attention-tacotron_amd/libsatt_hip.so 2>&1
for lib in tools/probes/libsatt_base.so self-attention-tacotron_amd/libsatt_hip.so; do
SATT_LIB_PATH=$lib python bench.py --chunks 1 --no-cpu-baseline --no-decode --steps 20 --time-all-kernels 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib standalone fwd', d['kernel_ms_per_step']['attn_rnn_fwd'])"
done
