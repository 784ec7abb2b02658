#!/bin/bash
DE265HIP_PIPE_TRACE=1 DE265HIP_PIPE_BATCH=${1:-2} DE265HIP_COPY_STREAMS=${2:-2} python bench.py --streams 3 --steps 6 --host-threads ${3:-6} --no-cpu-baseline --no-copy-out 2> /tmp/err.txt > /tmp/out.json
grep pipetrace /tmp/err.txt > gpurun_out/r4_pipetrace.txt
python -c "
import json; d=json.loads(open('/tmp/out.json').read()); print('value', d['value'])"
wc -l gpurun_out/r4_pipetrace.txt
