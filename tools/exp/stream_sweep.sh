#!/bin/bash
# streams x k_run workers x hardware queues: what limits more than three GOP streams per GPU
# usage (through gpurun): tools/exp/stream_sweep.sh > gpurun_out/stream_sweep.txt
for q in 4 8; do
  for w in 128 256 512; do
    for s in 3 4 6 8; do
      v=$(GPU_MAX_HW_QUEUES=$q DE265HIP_RUN_WORKERS=$w timeout -k 10 120 python3 bench.py --streams $s --steps 8 --warmup 2 --no-cpu-baseline --no-host-inclusive 2>/dev/null \
          | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
      echo "queues=$q workers=$w streams=$s  frames/s, ms/step: $v"
    done
  done
done
