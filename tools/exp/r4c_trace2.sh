#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/r4c_prof
DE265HIP_PIPE_CHAINS=${1:-1} DE265HIP_PIPE_BATCH=${2:-4} rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r4c_prof -o b -- python3 bench.py --streams 3 --steps 6 --warmup 1 --host-threads 9 --no-cpu-baseline --no-copy-out > gpurun_out/r4c_prof.json 2> gpurun_out/r4c_prof.err
python3 - <<'PY'
import csv,collections,glob,json
f=glob.glob('gpurun_out/r4c_prof/**/b_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0].replace('void d265::','').replace('d265::','').split('<')[0][:18],r['Stream_Id'],r['Grid_Size_Y']) for r in rows]
f2=glob.glob('gpurun_out/r4c_prof/**/b_memory_copy_trace.csv',recursive=True)[0]
for r in csv.DictReader(open(f2)):
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'H2D' if 'HOST_TO' in r['Direction'] else 'D2H',r['Stream_Id'],'-'))
ev.sort()
t0=ev[0][0]
orders=[e for e in ev if e[2]=='k_scan_order' and e[4]=='4']
mid=orders[int(len(orders)*0.5)][0]
print('scan/upload streams around t=%.1f ms'%((mid-t0)/1e6))
scan_streams=sorted(set(e[3] for e in ev if e[2]=='k_scan_tus'))
up=sorted(set(e[3] for e in ev if e[2]=='H2D'))
print('scan streams',scan_streams,'upload streams',up)
for s,e,n,st,gy in ev:
    if mid-4e6<=s<=mid+6e6 and (st in scan_streams or st in up):
        print('%9.3f %7.1f %-18s st%-3s y%s'%((s-t0)/1e6,(e-s)/1e3,n,st,gy))
PY
rm -rf gpurun_out/r4c_prof
