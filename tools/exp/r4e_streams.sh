#!/bin/bash
for s in 2 3 4 5 6 8; do
  python bench.py --streams $s --steps 12 --no-cpu-baseline --no-copy-out 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams $s: value', d['value'], 'replay', d['device_replay']['value'], 'threads', d['product_path']['host_threads'])"
done
