#!/bin/bash
# How many 4K Main10 streams does ONE MI355X serve when several decoder processes share it?  N processes of the patched libde265
# (oracle/_ref/f1_dec), each with T parse threads, decode the same f2_writer stream at once: aggregate pictures/s, CPU-only vs
# MI355X offload (pipelined).  At most 6 processes use the GPU (pool rule).   usage: tools/exp/e2e_multi.sh [T] [out]
cd "$(dirname "$0")/../.."
T=${1:-16}; OUT=${2:-gpurun_out/e2e_multi.txt}; mkdir -p "$(dirname "$OUT")"; TMP=$(mktemp -d)
LIB=$PWD/libde265_amd/libde265_hip.so; DEC=oracle/_ref/f1_dec
oracle/_ref/f2_writer out=$TMP/s.bin gop=B pics=96 w=3840 h=2160 bits=10 log2ctb=6 wpp=1 md5=0 seed=31
: > "$OUT"
for mode in cpu hip; do
  for N in 1 2 4 6; do
    pids=()
    for i in $(seq 1 $N); do
      if [ $mode = hip ]; then env F1_TIMING=1 F1_CHECK_HASH=0 F1_MODE=hip F1_HIP_LIB=$LIB F1_PIPELINE=4 F1_THREADS=$T timeout -k 5 200 $DEC $TMP/s.bin > $TMP/o_$i.txt 2>/dev/null &
      else env F1_TIMING=1 F1_CHECK_HASH=0 F1_THREADS=$T timeout -k 5 200 $DEC $TMP/s.bin > $TMP/o_$i.txt 2>/dev/null & fi
      pids+=($!)
    done
    for p in "${pids[@]}"; do wait $p; done
    tot=$(cat $TMP/o_*.txt | awk '/pictures\/s/ {s+=$3} END {printf "%.1f", s}')
    printf "%-4s %d process(es) x %2d threads: %8s pictures/s in total\n" $mode $N $T "$tot" | tee -a "$OUT"
    rm -f $TMP/o_*.txt
  done
done
rm -rf "$TMP"
