#!/bin/bash
export DE265HIP_TUNING=1      # the library reads its DE265HIP_* switches only in a process that sets this (csrc/env.h)
# GPU box: many small synthetic streams (oracle/_ref/f2_writer, all features, several configurations x seeds) decoded by the patched
# libde265 on the CPU and with the MI355X back end (synchronous and pipelined with worker threads); outputs must be byte-identical.
#   tools/exp/gpu_stream_sweep.sh [seeds_per_config] [out_file] [config_file: one f2_writer argument list per line; default: the list below]
cd "$(dirname "$0")/../.."
N=${1:-12}; OUT=${2:-gpurun_out/gpu_stream_sweep.txt}; mkdir -p "$(dirname "$OUT")"; TMP=$(mktemp -d)
LIB=$PWD/libde265_amd/libde265_hip.so; DEC=oracle/_ref/f1_dec; WR=oracle/_ref/f2_writer
bad=0; tot=0
: > "$OUT"
while read -r cfg; do
  [ -z "$cfg" ] && continue
  for seed in $(seq 1 $N); do
    tot=$((tot+1))
    $WR out=$TMP/s.bin seed=$seed $cfg || { echo "WRITER FAILED: $cfg seed=$seed" | tee -a "$OUT"; bad=$((bad+1)); continue; }
    timeout -k 5 60 $DEC $TMP/s.bin $TMP/cpu.yuv > $TMP/cpu.log 2>&1
    mode=$((seed % 2))
    if [ $mode = 0 ]; then env="F1_PIPELINE=0"; else env="F1_PIPELINE=3 F1_THREADS=4"; fi
    env F1_MODE=hip F1_HIP_LIB=$LIB $env timeout -k 5 60 $DEC $TMP/s.bin $TMP/hip.yuv > $TMP/hip.log 2>&1
    if ! cmp -s $TMP/cpu.yuv $TMP/hip.yuv || grep -v "Cannot run decoder multi-threaded" $TMP/hip.log | grep -qi "warning\|error\|mismatch"; then
      echo "MISMATCH: $cfg seed=$seed ($env): $(grep -v Cannot $TMP/hip.log | tail -2 | tr '\n' ' ')" | tee -a "$OUT"; bad=$((bad+1))
    fi
  done
  echo "done: $cfg" >> "$OUT"
done < <(if [ -n "$3" ]; then cat "$3"; else cat <<'CFGS'
gop=B pics=5 w=256 h=144 log2ctb=6 slices=2 wp=1
gop=P pics=4 w=176 h=144 log2ctb=4 log2maxtb=4 nref=4 lists_mod=1
gop=LDB pics=4 w=208 h=120 log2ctb=5 bits=10 sdh=1 tskip=1 cip=1 slices=4 lf_slices=0
gop=I pics=2 w=264 h=200 log2ctb=6 bits=10 pcm_bits=7 pcm_lf_off=1 tqbypass=1
gop=B pics=5 w=320 h=192 log2ctb=6 log2mincb=4 log2mintb=3 merge_cand=0 par_mrg=4 qg_depth=2 dens=80 max_level=60
gop=P pics=4 w=192 h=128 bits=12 wp=1 deblock=0 sao=1 qp=40
gop=LDB pics=4 w=192 h=128 bits=9 tmvp=0 strong=0 cuqpd=0 cb_off=-6 cr_off=7 qp=18 dens=90
gop=B pics=5 w=256 h=192 wpp=1 slices=3 scaling=2
gop=P pics=3 w=256 h=192 log2ctb=4 log2maxtb=4 tile_cols=4 tile_rows=3 tile_uniform=0 lf_tiles=0 slices=5 scaling=1
gop=LDB pics=4 w=256 h=192 dep=50 wpp=1 slices=2 bits=10
gop=P pics=3 w=256 h=192 log2ctb=4 log2maxtb=4 dep=40 tile_cols=3 tile_rows=2 slices=2
gop=B pics=5 w=136 h=104 log2ctb=4 log2maxtb=4 depth_inter=0 depth_intra=0 amp=0 pcm=0 bits=8 tqbypass=1 tskip=1
CFGS
fi)
echo "$((tot-bad)) of $tot streams identical" | tee -a "$OUT"
rm -rf "$TMP"
[ $bad = 0 ]
