#!/bin/bash
# product-path rate against the number of host threads: tools/exp/host_scaling.sh 3 6 9 12 15
for n in "$@"; do
  DE265HIP_PIPE_TIMING=1 timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --host-threads $n 2>/tmp/hs.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('threads $n', 'product fps %.0f (%.1f per thread) replay %.0f 1thr %.0f' % (d['value'], d['value']/$n, d['device_replay']['value'], d['product_path']['value_1_host_thread']))" || exit 1
  grep "pipeline:" /tmp/hs.err | head -3
done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -c processor /proc/cpuinfo; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
