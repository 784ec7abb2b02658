#!/bin/bash
# builds tools/exp/lib_<name>.so with ONE source file (k_tu.hip, k_lf.hip, k_mc.hip) compiled with extra flags
# usage: build_var_file.sh name file.hip -DFLAG=...
set -e
cd "$(dirname "$0")/../.."
name=$1; file=$2; shift; shift
python -m libde265_amd.build > /dev/null
B=libde265_amd/csrc/build
obj=$(basename $file .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $* -c libde265_amd/csrc/$file -o /tmp/${obj}_$name.o
objs=$(ls $B/*.o | grep -v "/$obj.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/exp/lib_$name.so $objs /tmp/${obj}_$name.o
echo built tools/exp/lib_$name.so
