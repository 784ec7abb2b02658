#!/bin/bash
# CPU decode vs HIP-backed decode of a stream with a dropped picture, for a few stream shapes
run() {
  name=$1; k=$2; shift 2
  oracle/_ref/f2_writer out=/tmp/w.bin "$@" > /dev/null
  python tools/drop_picture.py /tmp/w.bin /tmp/s.bin $k > /dev/null
  F1_CHECK_HASH=0 oracle/_ref/f1_dec /tmp/s.bin /tmp/cpu.yuv > /tmp/cpu.txt 2>&1
  F1_CHECK_HASH=0 F1_MODE=hip F1_HIP_LIB=libde265_amd/libde265_hip.so oracle/_ref/f1_dec /tmp/s.bin /tmp/hip.yuv > /tmp/hip.txt 2>&1
  if cmp -s /tmp/cpu.yuv /tmp/hip.yuv; then echo "$name: identical ($(head -1 /tmp/cpu.txt))"; else echo "$name: DIFFER ($(head -1 /tmp/cpu.txt) / $(head -1 /tmp/hip.txt)) $(cmp /tmp/cpu.yuv /tmp/hip.yuv | head -1)"; grep -i warn /tmp/cpu.txt | sort | uniq -c | head -3; fi
}
run B8 1 gop=B pics=9 bits=8 w=416 h=240 seed=6
run B10 1 gop=B pics=9 bits=10 w=416 h=240 seed=6
run B10wp 1 gop=B pics=9 bits=10 w=416 h=240 seed=6 wp=1
run P10 2 gop=P pics=7 bits=10 nref=2 w=416 h=240 seed=5
run B8k2 2 gop=B pics=9 bits=8 w=416 h=240 seed=6
run B8k5 5 gop=B pics=9 bits=8 w=416 h=240 seed=6
run LDB 3 gop=LDB pics=6 nref=3 w=832 h=480 tile_cols=2 tile_rows=2 seed=7
