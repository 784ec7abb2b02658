#!/bin/bash
for cfg in "16 2" "32 2" "32 3" "48 3" "64 4"; do set -- $cfg
  DE265HIP_PIPE_WINDOW=$1 DE265HIP_PIPE_CHAINS=$2 DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 20 --host-threads 12 --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('window $1 chains $2: value', d['value'], 'replay', d['device_replay']['value'])"
  grep "de265hip pipeline: [0-9][0-9][0-9]" /tmp/err.txt | head -1 | sed -e 's/.*ms per picture: //'
done
