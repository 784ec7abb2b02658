#!/bin/bash
export DE265HIP_TUNING=1      # the library reads its DE265HIP_* switches only in a process that sets this (csrc/env.h)
# SURVEY 8(f1)+(f2)+(f3) end to end on the GPU box: the SAME libde265 binary (oracle/_ref/f1_dec) decodes the SAME synthetic
# 4K Main10 / 1080p bitstreams (oracle/_ref/f2_writer) (a) entirely on the host CPU, 1 and N worker threads, and (b) with every
# reconstruction call offloaded to the MI355X, synchronous and pipelined (F1_PIPELINE=n: n host worker threads between parser and device).  Pictures/s of the decode loop, no
# output file, no hash check.   usage: tools/exp/e2e_stream_bench.sh [out_dir]
set -e
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/e2e}; mkdir -p "$OUT"; TMP=$(mktemp -d)
LIB=$PWD/libde265_amd/libde265_hip.so; DEC=oracle/_ref/f1_dec; WR=oracle/_ref/f2_writer
NT=${NT:-16}
run() { # label env... -- stream
  local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local best=0
  for rep in 1 2 3; do
    r=$(env F1_TIMING=1 F1_CHECK_HASH=0 "${envs[@]}" timeout -k 5 300 $DEC "$1" 2>/dev/null | tail -1 | awk '{print $3}')
    best=$(python3 -c "print(max($best, ${r:-0}))")
  done
  printf "%-34s %8.2f pictures/s\n" "$label" "$best" | tee -a "$OUT/e2e.txt"
}
: > "$OUT/e2e.txt"
for cfg in "4k10_B_wpp:gop=B pics=96 w=3840 h=2160 bits=10 log2ctb=6 wpp=1 md5=0 seed=31" \
           "4k10_B_wpp_dense:gop=B pics=64 w=3840 h=2160 bits=10 log2ctb=6 wpp=1 md5=0 dens=90 max_level=60 seed=32" \
           "1080p8_B_wpp:gop=B pics=128 w=1920 h=1080 log2ctb=6 wpp=1 md5=0 seed=33"; do
  name=${cfg%%:*}; args=${cfg#*:}
  $WR out=$TMP/$name.bin $args
  echo "== $name ($args; $(stat -c %s $TMP/$name.bin) bytes)" | tee -a "$OUT/e2e.txt"
  run "cpu libde265, 1 thread" -- $TMP/$name.bin
  run "cpu libde265, $NT threads" F1_THREADS=$NT -- $TMP/$name.bin
  run "hip sync, 1 thread" F1_MODE=hip F1_HIP_LIB=$LIB -- $TMP/$name.bin
  run "hip pipelined(1), 1 thread" F1_MODE=hip F1_HIP_LIB=$LIB F1_PIPELINE=1 -- $TMP/$name.bin
  run "hip pipelined(2), 4 threads" F1_MODE=hip F1_HIP_LIB=$LIB F1_PIPELINE=2 F1_THREADS=4 -- $TMP/$name.bin
  run "hip pipelined(2), $NT threads" F1_MODE=hip F1_HIP_LIB=$LIB F1_PIPELINE=2 F1_THREADS=$NT -- $TMP/$name.bin
  run "hip pipelined(4), $NT threads" F1_MODE=hip F1_HIP_LIB=$LIB F1_PIPELINE=4 F1_THREADS=$NT -- $TMP/$name.bin
  run "hip pipelined(6), $NT threads" F1_MODE=hip F1_HIP_LIB=$LIB F1_PIPELINE=6 F1_THREADS=$NT -- $TMP/$name.bin
done
rm -rf "$TMP"
