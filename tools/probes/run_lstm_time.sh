#!/bin/bash
python -m pytest tests/test_ops_gpu.py -q -k "lstm" -x 2>&1 | tail -3
SATT_LIB_PATH=tools/probes/libsatt_base.so python tools/lstm_time.py 2>&1 | tail -3
python tools/lstm_time.py 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py -q -x 2>&1 | tail -3
bash tools/ab_bench.sh tools/probes/libsatt_base.so self-attention-tacotron_amd/libsatt_hip.so 2>&1
timeout 1200 python tools/chunk_sweep.py --wide > gpurun_out/wide_sweep.txt 2>&1
