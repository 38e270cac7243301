#!/bin/bash
# tools/build_variant.sh <name> <file.hip> [-DFLAG ...]: a variant of libsatt_hip.so in which ONE source is recompiled with extra
# flags (timing experiments, profile marks); the other objects come from the regular in-tree build.  Output:
# tools/probes/libsatt_<name>.so - use with SATT_LIB_PATH=tools/probes/libsatt_<name>.so python bench.py ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/self-attention-tacotron_amd/csrc
name=$1; src=$2; shift 2
python -c "import sys; sys.path.insert(0,'$R'); import __graft_entry__ as g; g.build()" > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc "$@" -c $C/$src -o /tmp/variant_$name.o
objs=""
for f in gemm gemm_tile flash small_attn elementwise highway lstm lstm_cluster attn_rnn attn_cluster decode api; do
  if [ "$f.hip" == "$src" ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs $C/build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/tools/probes/libsatt_$name.so
echo $R/tools/probes/libsatt_$name.so
