#!/bin/bash
# round 4: product path under runtime settings that change how copies and launches are issued
set -o pipefail
run() {
  env "$@" DE265HIP_PIPE_BATCH=2 DE265HIP_COPY_STREAMS=2 DE265HIP_PIPE_TIMING=1 python bench.py --streams 3 --steps 10 --host-threads 6 --no-cpu-baseline --no-copy-out 2> /tmp/err.txt > /tmp/out.json
  if grep -q "Memory access fault" /tmp/err.txt; then echo "GPU FAULT"; exit 1; fi
  python -c "
import json,sys
d=json.loads(open('/tmp/out.json').read()); print('$*: value', d['value'], 'replay', d['device_replay']['value'])" || tail -3 /tmp/err.txt
  grep "enqueue sections" /tmp/err.txt | head -1
}
run A=1
run HSA_ENABLE_SDMA=0
run HIP_FORCE_DEV_KERNARG=1
run HSA_ENABLE_SDMA=0 HIP_FORCE_DEV_KERNARG=1
run AMD_DIRECT_DISPATCH=0
