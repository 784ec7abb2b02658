#!/bin/bash
# where the host side of the HIP-backed libde265 goes (F1_PROFILE=1), 4K Main10 B stream; GPU box
cd "$(dirname "$0")/../.."
TMP=$(mktemp -d); LIB=$PWD/libde265_amd/libde265_hip.so; DEC=oracle/_ref/f1_dec
oracle/_ref/f2_writer out=$TMP/s.bin gop=B pics=${PICS:-24} w=3840 h=2160 bits=10 log2ctb=6 wpp=1 md5=0 seed=31
for t in 0 16; do
  for pl in 0 1 6; do
    echo "== threads $t pipeline $pl"
    F1_PROFILE=1 F1_TIMING=1 F1_CHECK_HASH=0 F1_MODE=hip F1_HIP_LIB=$LIB F1_PIPELINE=$pl F1_THREADS=$t DE265HIP_BUILD_TIMING=${BT:-0} timeout -k 5 120 $DEC $TMP/s.bin 2>&1 | tail -5
  done
done
rm -rf $TMP
