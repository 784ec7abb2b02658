#!/bin/bash
# builds tools/exp/lib_<name>.so: the library with k_tu.hip compiled with extra flags (A/B experiments in one gpurun call)
# usage: build_var.sh name -DFLAG=...
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
python -m libde265_amd.build > /dev/null
B=libde265_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $* -c libde265_amd/csrc/k_tu.hip -o /tmp/k_tu_$name.o
objs=$(ls $B/*.o | grep -v k_tu.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/exp/lib_$name.so $objs /tmp/k_tu_$name.o
echo built tools/exp/lib_$name.so
