#!/bin/bash
python -m pytest tests/test_ops_gpu.py -q -k "lstm" -x 2>&1 | tail -3
for rep in 1 2; do
SATT_LIB_PATH=tools/probes/libsatt_base.so python tools/lstm_time.py 2>&1 | tail -3
python tools/lstm_time.py 2>&1 | tail -3
done
python -m pytest tests/test_model_gpu.py -q -x 2>&1 | tail -3
bash tools/ab_bench.sh tools/probes/libsatt_base.so self-attention-tacotron_amd/libsatt_hip.so 2>&1
