#!/bin/bash
for o in 0 1; do for cfg in "9 2" "12 2" "12 3" "9 1"; do set -- $cfg
  if [ $o = 1 ]; then export DE265HIP_PIPE_ANY_ORDER=1; else unset DE265HIP_PIPE_ANY_ORDER; fi
  DE265HIP_PIPE_CHAINS=$2 DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 20 --host-threads $1 --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('any-order $o host-threads $1 chains $2: value', d['value'])"
  grep "de265hip pipeline: [0-9][0-9][0-9]" /tmp/err.txt | head -1 | sed -e 's/.*ms per picture: //'
done; done
