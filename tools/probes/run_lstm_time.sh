#!/bin/bash
SATT_LIB_PATH=tools/probes/libsatt_lprof.so python tools/lstm_time.py 2>&1 | grep -A1 "fwd.*bwd" | head -2
