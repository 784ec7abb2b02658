#!/bin/bash
# host_inclusive rate (build -> run -> free through the C ABI) against the number of host threads building pictures
for n in ${THREADS:-4 8 16 32 64}; do
  echo -n "host threads $n: "
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --host-threads $n 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_inclusive']; print(h['value'], 'fps;', h['value_1_host_thread'], 'with one thread')" || exit 1
done
