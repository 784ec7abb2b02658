#!/bin/bash
for n in 0 1 2 4; do for el in 64 4000000; do
  DE265HIP_BENCH_NOISE=$n DE265HIP_BENCH_NOISE_ELEMS=$el python bench.py --streams 3 --steps 30 --no-host-inclusive --no-cpu-baseline 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('noise threads $n elems $el: replay', d['device_replay']['value'])"
  grep noise /tmp/err.txt
done; done
