#!/bin/bash
# which stage bounds the product path?  the same run without the reconstruction launches (builds, uploads, scans only)
for t in 6 12; do for nr in 0 1; do
  if [ $nr = 1 ]; then export DE265HIP_PIPE_NO_RUN=1; else unset DE265HIP_PIPE_NO_RUN; fi
  DE265HIP_PIPE_TRACE=1 DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 10 --host-threads $t --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('host-threads $t no-run $nr: value', d['value'], 'replay', d['device_replay']['value'])"
  grep "de265hip pipeline: 2" /tmp/err.txt | head -1
  python tools/exp/pipe_analyze.py /tmp/err.txt 64 420 | head -10
done; done
