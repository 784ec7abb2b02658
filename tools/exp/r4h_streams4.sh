#!/bin/bash
for cfg in "3 9" "4 8" "4 12" "5 10" "6 12" "4 8 1" ; do set -- $cfg
  DE265HIP_PIPE_CHAINS=${3:-2} DE265HIP_PIPE_TIMING=1 python bench.py --streams $1 --steps 20 --host-threads $2 --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams $1 host-threads $2 chains ${3:-2}: value', d['value'], 'replay', d['device_replay']['value'])"
  grep "de265hip pipeline: [0-9][0-9][0-9]" /tmp/err.txt | head -1 | sed -e 's/.*ms per picture: //'
done
