#!/bin/bash
for rep in 1 2; do for cfg in "2 4 9" "2 8 9" "1 8 9" "3 4 9" "2 4 6" "2 8 6"; do set -- $cfg
  DE265HIP_PIPE_CHAINS=$1 DE265HIP_PIPE_BATCH=$2 python bench.py --streams 3 --steps 30 --host-threads $3 --no-cpu-baseline --no-copy-out 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chains $1 batch $2 host-threads $3: value', d['value'])"
done; done
