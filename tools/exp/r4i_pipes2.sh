#!/bin/bash
for cfg in "1024 2 9" "2048 2 9" "4096 2 9" "8192 2 9" "2048 3 9" "2048 3 12" "4096 3 12" "2048 1 9"; do set -- $cfg
  DE265HIP_SCAN_GRID=$1 DE265HIP_PIPE_CHAINS=$2 DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 20 --host-threads $3 --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scan grid $1 chains $2 host-threads $3: value', d['value'], 'replay', d['device_replay']['value'])"
  grep "de265hip pipeline: [0-9][0-9][0-9]" /tmp/err.txt | head -1 | sed -e 's/.*ms per picture: //'
done
