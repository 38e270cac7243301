#!/bin/bash
# encoder-LSTM single-barrier kernels: tests, A/B bench against the two-barrier kernels of the previous commit, phase marks, tail sweep
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -q -k "lstm" -x 2>&1 | tail -5
python -m pytest tests/test_model_gpu.py -q -x 2>&1 | tail -3
bash tools/ab_bench.sh tools/probes/libsatt_base.so self-attention-tacotron_amd/libsatt_hip.so 2>&1 | tee gpurun_out/lstm_ab.txt
python tools/phase_marks.py 2>&1 | tail -20 | tee gpurun_out/lstm_phases.txt
timeout 900 python tools/chunk_sweep.py --tdiv 2>&1 | tee gpurun_out/tdiv_sweep.txt
