#!/bin/bash
# builds tools/exp/lib_abl.so: the library with k_run's timing-only ablation switches (DE265HIP_DEBUG bits) compiled in
set -e
cd "$(dirname "$0")/../.."
python -m libde265_amd.build > /dev/null
B=libde265_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DD265_ABLATE $* -c libde265_amd/csrc/k_tu.hip -o /tmp/k_tu_abl.o
objs=$(ls $B/*.o | grep -v k_tu.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/exp/lib_abl.so $objs /tmp/k_tu_abl.o
echo built tools/exp/lib_abl.so
