#!/bin/bash
# isolated kernel times of the bench GOP: prints kernels_isolated of a 1-stream device-replay bench
python bench.py --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --no-e2e 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('replay fps %.0f' % d['device_replay']['value'] if d.get('device_replay') else d['value']); print({k: round(v['us_per_picture'],1) for k,v in d['kernels_isolated'].items()})"
