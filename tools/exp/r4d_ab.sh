#!/bin/bash
for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg
  DE265HIP_SCAN_PREFIX_TAIL=$1 DE265HIP_OWN_PREP=$2 DE265HIP_PIPE_CHAINS=1 DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 10 --host-threads 9 --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('prefix-tail $1 own-prep $2: value', d['value'], 'replay', d['device_replay']['value'])"
  grep "de265hip pipeline: 2" /tmp/err.txt | head -1 | sed -e 's/.*| //'
done
