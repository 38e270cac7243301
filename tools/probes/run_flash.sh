#!/bin/bash
python -m pytest tests/test_flash_gpu.py -q -x 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py -q -x -k "bf16 or production or full_size or identical" 2>&1 | tail -3
python tools/flash_time.py 2>&1 | tail -6
bash tools/ab_bench.sh tools/probes/libsatt_base.so self-attention-tacotron_amd/libsatt_hip.so 2>&1
python tools/phase_marks.py 2>&1 | grep "head\|total"
