#!/bin/bash
for rep in 1 2; do for so in tools/exp/lib_prio0.so tools/exp/lib_prio1.so libde265_amd/libde265_hip.so; do
  DE265HIP_SO=$PWD/$so DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 20 --host-threads 9 --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$so: value', d['value'], 'replay', d['device_replay']['value'])"
  grep "de265hip pipeline: [0-9][0-9][0-9]" /tmp/err.txt | head -1 | sed -e 's/.*ms per picture: //'
done; done
