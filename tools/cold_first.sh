#!/bin/bash
# the FIRST GPU process of a fresh gpurun box = one "first process on a fresh box" cold trial of the persistent decode kernel (the one
# r5 event happened in that position); appended to gpurun_out/cold/first_<epoch>.jsonl (one file per box: gpurun merges by file name)
mkdir -p gpurun_out/cold
python tools/decode_cold.py b1 --dump gpurun_out/cold --tag firstbox 2>/dev/null | grep '^{' > gpurun_out/cold/first_$(date +%s).jsonl
