#!/bin/bash
export DE265HIP_TUNING=1      # the library reads its DE265HIP_* switches only in a process that sets this (csrc/env.h)
# Collects the round's profiling evidence on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag>        e.g. r01_d
# 1. rocprofv3 --kernel-trace --stats of the default bench (3 GOP streams) and of --streams 1
# 2. two PMC passes (FETCH_SIZE, WRITE_SIZE; kernel trace only, never with sys/hip/hsa tracing) of --streams 1
# Everything lands under gpurun_out/prof_<tag>/; copy the summaries into profiles/ afterwards.
set -e -o pipefail
tag=${1:-rXX}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --streams 1 --no-cpu-baseline --no-host-inclusive --no-e2e > $out/bench_1stream.json 2> $out/bench_1stream.err
python3 bench.py --streams 1 --lanes 2 --no-cpu-baseline --no-host-inclusive --no-e2e > $out/bench_1stream_2lanes.json 2> $out/bench_1stream_2lanes.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --no-e2e \
  > $out/bench_under_rocprof_3streams.json 2> $out/stats3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -- python3 bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --no-e2e \
  > $out/bench_under_rocprof_1stream.json 2> $out/stats1.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive --no-e2e \
  > $out/pmc_fetch.json 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive --no-e2e \
  > $out/pmc_write.json 2> $out/pmc_write.err
python3 tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json \
  "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive --no-e2e (two separate passes)" > /dev/null
# keep the merged-back payload small: the per-dispatch traces are not needed, only the stats
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
ls -R $out | head -40
