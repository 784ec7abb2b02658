#!/bin/bash
# builds tools/exp/lib_<name>.so with k_lf.hip AND host.hip compiled with extra flags (tile shape of k_lf_tile lives in kernels.h)
# usage: build_var_lf.sh name -DLF_TW=.. -DLF_TH=..
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
python -m libde265_amd.build > /dev/null
B=libde265_amd/csrc/build
for f in k_lf host; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $* -c libde265_amd/csrc/$f.hip -o /tmp/${f}_$name.o &
done
wait
objs=$(ls $B/*.o | grep -v -E "k_lf.o|host.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/exp/lib_$name.so $objs /tmp/k_lf_$name.o /tmp/host_$name.o
echo built tools/exp/lib_$name.so
