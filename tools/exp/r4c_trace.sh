#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/r4c_prof
DE265HIP_PIPE_CHAINS=${1:-1} DE265HIP_PIPE_BATCH=${2:-4} rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4c_prof -o b -- python3 bench.py --streams 3 --steps 6 --warmup 1 --host-threads 9 --no-cpu-baseline --no-copy-out > gpurun_out/r4c_prof.json 2> gpurun_out/r4c_prof.err
python3 - <<'PY'
import csv,collections,glob,json
f=glob.glob('gpurun_out/r4c_prof/**/b_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
d=json.loads(open('gpurun_out/r4c_prof.json').read().strip().splitlines()[-1]); print('value',d['value'],'replay',d['device_replay']['value'])
agg=collections.defaultdict(lambda:[0,0])
for r in rows:
    n=r['Kernel_Name'].split('(')[0].replace('void d265::','').replace('d265::','').split('<')[0]
    k=(n, r['Grid_Size_Y'] if 'scan' in n else '-')
    agg[k][0]+=1; agg[k][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,v in sorted(agg.items()): print('%-22s y=%-4s n %5d avg %8.1f us total %8.1f ms'%(k[0],k[1],v[0],v[1]/v[0]/1e3,v[1]/1e6))
PY
rm -rf gpurun_out/r4c_prof
