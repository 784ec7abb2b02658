#!/bin/bash
# builds tools/exp/lib_<name>.so with k_tu.hip AND host.hip compiled with extra flags (macros both sides must agree on)
# usage: build_var2.sh name -DFLAG=...
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
python -m libde265_amd.build > /dev/null
B=libde265_amd/csrc/build
for f in k_tu host; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $* -c libde265_amd/csrc/$f.hip -o /tmp/${f}_$name.o &
done
wait
objs=$(ls $B/*.o | grep -v "/k_tu.o\|/host.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/exp/lib_$name.so $objs /tmp/k_tu_$name.o /tmp/host_$name.o
echo built tools/exp/lib_$name.so
