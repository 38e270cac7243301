#!/bin/bash
python -m pytest tests/test_ops_gpu.py -q -x -k "attn or saf or param" 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py tests/test_pinned_gpu.py -q -x 2>&1 | tail -3
bash tools/ab_bench.sh tools/probes/libsatt_base.so self-attention-tacotron_amd/libsatt_hip.so 2>&1
python tools/phase_marks.py 2>&1 | tail -18
