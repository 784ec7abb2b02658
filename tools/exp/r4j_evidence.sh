#!/bin/bash
export DE265HIP_TUNING=1      # the library reads its DE265HIP_* switches only in a process that sets this (csrc/env.h)
# round 4: the cheap evidence items (f4 bench line + kernel stats, end-to-end stream bench, F1_PROFILE, two-rank rehearsal)
export TMPDIR=/tmp
O=gpurun_out/r4j; mkdir -p $O
python3 bench.py --chroma-format 3 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_444.json 2> $O/bench_444.err
python3 bench.py --chroma-format 2 --steps 10 --warmup 2 --no-cpu-baseline --no-copy-out > $O/bench_422.json 2> $O/bench_422.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats444 -- python3 bench.py --chroma-format 3 --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive > $O/bench_444_rocprof.json 2> $O/stats444.err
find $O/stats444 -name "*kernel_trace.csv" -delete
tools/exp/e2e_stream_bench.sh $O/e2e > $O/e2e.log 2>&1
# F1_PROFILE: where the host side of a picture goes
oracle/_ref/f2_writer out=/tmp/p.bin gop=B pics=64 w=3840 h=2160 bits=10 log2ctb=6 wpp=1 md5=0 seed=31 > /dev/null
F1_PROFILE=1 F1_TIMING=1 F1_CHECK_HASH=0 F1_MODE=hip F1_HIP_LIB=$PWD/libde265_amd/libde265_hip.so F1_PIPELINE=4 F1_THREADS=16 oracle/_ref/f1_dec /tmp/p.bin > $O/f1_profile.txt 2>&1
python3 bench.py --gpus 2 --single-device --backend gloo --steps 3 --warmup 1 --open-gop --no-cpu-baseline > $O/rehearsal_2ranks.json 2> $O/rehearsal.err
ls $O
