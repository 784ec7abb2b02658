#!/bin/bash
for rep in 1 2; do for v in 0 1; do
  DE265HIP_LF_TILE=$v python bench.py --steps 30 --no-cpu-baseline --no-copy-out 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('LF_TILE $v: value', d['value'], 'replay', d['device_replay']['value'])"
done; done
for v in 0 1; do DE265HIP_RESID_ONE_LAUNCH=$v python bench.py --steps 30 --no-cpu-baseline --no-copy-out 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('RESID_ONE_LAUNCH $v: value', d['value'], 'replay', d['device_replay']['value'])"; done
