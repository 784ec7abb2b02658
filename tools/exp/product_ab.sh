#!/bin/bash
# product-path rate of the bench for environment settings: tools/exp/product_ab.sh "A=1" "A=2" ...
for e in "$@"; do
  env $e DE265HIP_PIPE_TIMING=1 timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline $BENCH_ARGS 2>/tmp/pab.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', 'product fps %.0f ms/step %.2f replay %.0f 1thr %.0f' % (d['value'], d['ms_per_step'], d['device_replay']['value'], d['product_path']['value_1_host_thread']))" || exit 1
  grep "pipeline:" /tmp/pab.err | head -3
done
