#!/bin/bash
for cfg in "9 1 4" "9 2 4" "12 1 4" "12 2 4"; do set -- $cfg
  DE265HIP_PIPE_TRACE=1 DE265HIP_PIPE_CHAINS=$2 DE265HIP_PIPE_BATCH=$3 DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 10 --host-threads $1 --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('host-threads $1 chains $2 batch $3: value', d['value'], 'replay', d['device_replay']['value'])"
  grep "de265hip pipeline: 2" /tmp/err.txt | head -1
  python tools/exp/pipe_analyze.py /tmp/err.txt 64 420 > /tmp/pa.txt; head -9 /tmp/pa.txt
done
