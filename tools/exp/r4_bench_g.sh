#!/bin/bash
# round 4: product path against batch size of the scan, copy streams, host threads
set -o pipefail
for cfg in "4 2 6" "4 2 9" "2 2 6" "1 2 6" "4 3 9" "4 1 9" "8 2 9"; do
    set -- $cfg
    DE265HIP_PIPE_BATCH=$1 DE265HIP_COPY_STREAMS=$2 DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 10 --host-threads $3 --no-cpu-baseline --no-copy-out 2> /tmp/err.txt > /tmp/out.json
    if grep -q "Memory access fault" /tmp/err.txt; then echo "GPU FAULT"; tail -5 /tmp/err.txt; exit 1; fi
    python -c "
import json,sys
d=json.loads(open('/tmp/out.json').read()); print('batch $1 copy-streams $2 host-threads $3: value', d['value'], 'replay', d['device_replay']['value'])" || { tail -5 /tmp/err.txt; exit 1; }
    grep "de265hip pipeline\|de265hip decoder" /tmp/err.txt | sed -n '1p;5p'
done
