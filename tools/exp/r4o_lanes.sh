#!/bin/bash
for cfg in "3 1" "3 2" "2 2" "4 1" "2 1"; do set -- $cfg
  python bench.py --streams $1 --lanes $2 --steps 20 --host-threads 9 --no-cpu-baseline --no-copy-out 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams $1 lanes $2: value', d['value'], 'replay', d['device_replay']['value'])"
done
