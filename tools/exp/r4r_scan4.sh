#!/bin/bash
for n in 3 4; do for cfg in "1 8" "2 4"; do set -- $cfg
  DE265HIP_SCAN_STREAMS=$n DE265HIP_PIPE_CHAINS=$1 DE265HIP_PIPE_BATCH=$2 python bench.py --steps 30 --no-cpu-baseline --no-copy-out 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scan streams $n chains $1 batch $2: value', d['value'], 'replay', d['device_replay']['value'])"
done; done
