#!/bin/bash
# builds tools/exp/lib_<name>.so with SEVERAL source files compiled with extra flags (macros both sides must agree on)
# usage: build_var2.sh name "k_tu host" -DFLAG=...      (default file list when the 2nd argument starts with -: "k_tu host")
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
files="k_tu host"
case "$1" in -*) ;; *) files=$1; shift;; esac
python -m libde265_amd.build > /dev/null
B=libde265_amd/csrc/build
pat=""
for f in $files; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $* -c libde265_amd/csrc/$f.hip -o /tmp/${f}_$name.o &
  pat="$pat\|/$f.o"
done
wait
objs=$(ls $B/*.o | grep -v "${pat#\\|}")
vobjs=""; for f in $files; do vobjs="$vobjs /tmp/${f}_$name.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/exp/lib_$name.so $objs $vobjs
echo built tools/exp/lib_$name.so
