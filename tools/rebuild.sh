#!/bin/bash
# rebuild libsatt_hip.so in-tree (any cwd)
cd "$(dirname "$0")/.." && python -c "
import importlib.util
spec=importlib.util.spec_from_file_location('b','self-attention-tacotron_amd/csrc/build.py'); m=importlib.util.module_from_spec(spec); spec.loader.exec_module(m); m.build(verbose=False)"
