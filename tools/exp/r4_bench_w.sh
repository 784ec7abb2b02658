#!/bin/bash
# round 4: product path against copy streams per decoder, with the launcher's and the decoders' timing
set -o pipefail
for c in 1 2 3; do
    DE265HIP_COPY_STREAMS=$c DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 10 --no-cpu-baseline --no-copy-out 2> /tmp/err.txt > /tmp/out.json
    if grep -q "Memory access fault" /tmp/err.txt; then echo "GPU FAULT"; tail -5 /tmp/err.txt; exit 1; fi
    python -c "
import json,sys
d=json.loads(open('/tmp/out.json').read()); print('copy-streams $c value', d['value'], 'replay', d['device_replay']['value'], '1thr', d['product_path']['value_1_host_thread'])" || { tail -5 /tmp/err.txt; exit 1; }
    grep "de265hip pipeline\|de265hip decoder" /tmp/err.txt | head -8
done
