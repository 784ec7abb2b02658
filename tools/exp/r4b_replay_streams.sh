#!/bin/bash
# device replay against the number of HIP streams a decoder creates (which hardware queue its kernel stream lands on)
for c in 1 2 3 4 5 6; do
  for q in "" 8; do
    v=$(GPU_MAX_HW_QUEUES=${q:-4} DE265HIP_COPY_STREAMS=$c python bench.py --streams 3 --steps 10 --no-host-inclusive --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['device_replay']['value'])")
    echo "copy-streams $c hwq ${q:-4}: replay $v"
  done
done
