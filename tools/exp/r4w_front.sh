#!/bin/bash
for rep in 1 2; do for v in "" "DE265HIP_NO_FRONT=1"; do
  env $v python bench.py --steps 30 --no-cpu-baseline --no-copy-out 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${v:-default}: value', d['value'], 'replay', d['device_replay']['value'])"
done; done
