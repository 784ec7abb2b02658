#!/bin/bash
DE265HIP_PIPE_TRACE=1 DE265HIP_PIPE_CHAINS=${1:-1} python bench.py --streams 3 --steps 10 --host-threads ${2:-9} --no-cpu-baseline --no-copy-out 2>/tmp/err.txt >/tmp/out.json
python - <<'PY'
import collections
ch=collections.defaultdict(list); pt=collections.defaultdict(list)
for l in open('/tmp/err.txt'):
    f=l.split()
    if l.startswith('chaintrace'): ch[f[1]].append([float(x) for x in f[2:]])
    if l.startswith('pipetrace'): pt[f[1]].append([float(x) for x in f[2:]])
for p,c in ch.items():
    c.sort()
    if len(c)<60: continue
    c=c[20:80]
    t0=c[0][0]
    print(p)
    prev=None
    for e,r,nb,inf in c[:40]:
        gap=(e-prev)*1e3 if prev else 0
        print('  enq %8.2f ms  dur %5.2f  gap-since-last-report %5.2f  ready-map %2d in-flight %2d'%((e-t0)*1e3,(r-e)*1e3,gap,nb,inf))
        prev=r
    break
PY
