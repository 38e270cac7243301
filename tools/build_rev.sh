#!/bin/bash
# tools/build_rev.sh <git-rev> [name]: libsatt_hip.so of another revision -> ab_libs/libsatt_<name>.so (A/B partner of the in-tree
# build for tools/ab_bench.sh; ab_libs/ travels to the GPU box, is git-ignored, delete it after the experiment)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); rev=$1; name=${2:-$1}
W=/tmp/satt_rev_$name; rm -rf $W; mkdir -p $W/pkg/csrc $W/include
for f in $(git -C $R ls-tree --name-only $rev self-attention-tacotron_amd/csrc/ | grep -E "\.(hip|h)$"); do git -C $R show $rev:$f > $W/pkg/csrc/$(basename $f); done
git -C $R show $rev:include/satt_hip.h > $W/include/satt_hip.h
cd $W/pkg/csrc
srcs=$(python - <<PY
import re
print(" ".join(re.search(r'SOURCES = \[(.*?)\]', open("$R/self-attention-tacotron_amd/csrc/build.py").read(), re.S).group(1).replace('"','').replace(',',' ').split()))
PY
)
for s in $srcs; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -c $s -o ${s%.hip}.o & done; wait
mkdir -p $R/ab_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(for s in $srcs; do echo ${s%.hip}.o; done) -o $R/ab_libs/libsatt_$name.so
echo $R/ab_libs/libsatt_$name.so
