#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/r4h_prof
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r4h_prof -o b -- python3 bench.py --streams 3 --steps 12 --warmup 2 --host-threads 9 --no-cpu-baseline --no-copy-out $BUSY_EXTRA > gpurun_out/r4h_prof.json 2> gpurun_out/r4h_prof.err
python3 - <<'PY'
import csv,collections,glob,json
f=glob.glob('gpurun_out/r4h_prof/**/b_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
d=json.loads(open('gpurun_out/r4h_prof.json').read().strip().splitlines()[-1]); print('value',d['value'],'replay',d['device_replay']['value'])
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0].replace('void d265::','').replace('d265::','').split('<')[0][:18],r['Stream_Id']) for r in rows]
ev.sort()
# product region: between first and last k_scan_order with many k_run around -> take the longest stretch where scan kernels occur with gaps < 5 ms
sc=[e for e in ev if e[2]=='k_scan_order']
# find the largest cluster
best=(0,0,0); i=0
while i<len(sc):
    j=i
    while j+1<len(sc) and sc[j+1][0]-sc[j][1]<3e6: j+=1
    if j-i>best[0]: best=(j-i,sc[i][0],sc[j][1])
    i=j+1
n,t0,t1=best
print('product region %.1f ms, %d scan batches'%((t1-t0)/1e6,n))
reg=[e for e in ev if e[0]>=t0 and e[1]<=t1]
per=collections.defaultdict(list)
for e in reg: per[e[3]].append(e)
tot=t1-t0
for st,l in sorted(per.items(), key=lambda x:int(x[0])):
    busy=sum(e[1]-e[0] for e in l)
    names=collections.Counter(e[2] for e in l)
    print('stream %3s busy %5.1f%%  kernels %5d  %s'%(st,100*busy/tot,len(l),dict(names.most_common(3))))
# union busy + concurrency
pts=[]
for s,e,_,_ in reg: pts.append((s,1)); pts.append((e,-1))
pts.sort()
cur=0; last=t0; hist=collections.Counter()
for t,dv in pts:
    hist[cur]+=t-last; last=t; cur+=dv
print('concurrency histogram (%% of time): '+' '.join('%d:%.1f'%(k,100*v/tot) for k,v in sorted(hist.items())))
byk=collections.defaultdict(list)
for e in reg: byk[e[2]].append(e[1]-e[0])
print('kernel durations inside the region (us): '+'  '.join('%s %.0f' % (k, sum(v)/len(v)/1e3) for k,v in sorted(byk.items(), key=lambda kv:-sum(kv[1]))))
nrun=sum(1 for e in reg if e[2]=='k_run')
print('pictures (k_run launches) in region', nrun, '-> %.0f pictures/s'%(nrun/(tot/1e9)))
PY
rm -rf gpurun_out/r4h_prof
