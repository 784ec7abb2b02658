#!/bin/bash
set -o pipefail
DE265HIP_PIPE_BATCH=${1:-2} DE265HIP_COPY_STREAMS=${2:-2} DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 10 --host-threads ${3:-6} --no-cpu-baseline --no-copy-out 2> /tmp/err.txt > /tmp/out.json
if grep -q "Memory access fault" /tmp/err.txt; then echo "GPU FAULT"; tail -5 /tmp/err.txt; exit 1; fi
python -c "
import json,sys
d=json.loads(open('/tmp/out.json').read()); print('value', d['value'], 'replay', d['device_replay']['value'])" || { tail -5 /tmp/err.txt; exit 1; }
grep "de265hip" /tmp/err.txt | head -14
