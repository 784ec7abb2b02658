#!/bin/bash
# round 4, second session: where does the product path stand with the last build?  (host stage per phase, pipeline trace
# summaries at 6 / 9 / 12 workers, kernel trace of the product path)
set -o pipefail
export TMPDIR=/tmp
DE265HIP_BUILD_TIMING=1 python tools/time_build.py > gpurun_out/r4b_time_build.txt 2>&1
for t in 6 9 12; do
  DE265HIP_PIPE_TRACE=1 DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 10 --host-threads $t --no-cpu-baseline --no-copy-out 2> /tmp/err_$t.txt > /tmp/out_$t.json
  echo "== host-threads $t" >> gpurun_out/r4b_pipe.txt
  python -c "
import json; d=json.loads(open('/tmp/out_$t.json').read()); print('value', d['value'], 'replay', d['device_replay']['value'], 'one thread', d['product_path']['value_1_host_thread'])" >> gpurun_out/r4b_pipe.txt
  grep "de265hip" /tmp/err_$t.txt | grep -v pipetrace | head -12 >> gpurun_out/r4b_pipe.txt
  python tools/exp/pipe_analyze.py /tmp/err_$t.txt 64 420 >> gpurun_out/r4b_pipe.txt
done
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r4b_prof -o b -- python3 bench.py --streams 3 --steps 4 --warmup 1 --host-threads 9 --no-cpu-baseline --no-copy-out > gpurun_out/r4b_prof.json 2> gpurun_out/r4b_prof.err
ls gpurun_out/r4b_prof
