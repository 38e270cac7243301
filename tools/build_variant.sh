#!/bin/bash
# tools/build_variant.sh <name> "<file.hip> [file2.hip ...]" [-DFLAG ...]: a variant of libsatt_hip.so in which the named sources
# are recompiled with extra flags (timing experiments, profile marks); the other objects come from the regular in-tree build.
# Output: tools/probes/libsatt_<name>.so - use with SATT_LIB_PATH=tools/probes/libsatt_<name>.so python bench.py ...
# (tools/probes/*.so is listed in .gpurunignore: build the variants ON the GPU box, inside the gpurun command.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/self-attention-tacotron_amd/csrc
name=$1; srcs=$2; shift 2
python -c "import sys; sys.path.insert(0,'$R'); import __graft_entry__ as g; g.build()" > /dev/null
for s in $srcs; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc "$@" -c $C/$s -o /tmp/variant_${name}_${s%.hip}.o &
done
wait
objs=""
for f in $(python -c "import re;print(' '.join(x[:-4] for x in re.search(r'SOURCES = \[(.*?)\]', open('$C/build.py').read(), re.S).group(1).replace('\"','').replace(',',' ').split()))"); do
  if [[ " $srcs " == *" $f.hip "* ]]; then objs="$objs /tmp/variant_${name}_$f.o"; else objs="$objs $C/build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/tools/probes/libsatt_$name.so
echo $R/tools/probes/libsatt_$name.so
