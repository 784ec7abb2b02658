#!/bin/bash
# host_inclusive leg of bench.py under different environments: tools/exp/host_incl_ab.sh "A=1 B=2" "C=3" ...
for e in "$@"; do
  v=$(env $e timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); h=d['host_inclusive']; print(h['value'], h['value_1_host_thread'], h['host_threads'])")
  echo "$e : host_inclusive fps, 1-thread fps, threads: $v"
done
