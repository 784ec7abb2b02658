"""Reads a rocprofv3 kernel trace CSV and reports how many kernels / queues are active at once.
python tools/exp/trace_overlap.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = []
queues = collections.Counter()
for r in rows:
    if "d265" not in r["Kernel_Name"]:
        continue
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")
    ev.append((s, 1, q)); ev.append((e, -1, q)); queues[q] += 1
ev.sort()
cur = 0; hist = collections.Counter(); last = ev[0][0]; active = collections.Counter()
for t, d, q in ev:
    hist[(cur, sum(1 for v in active.values() if v > 0))] += t - last
    last = t; cur += d; active[q] += d
tot = sum(hist.values())
print("kernels per queue id:", dict(queues))
print("time share by (kernels running, queues active):")
for k in sorted(hist):
    if hist[k] / tot > 0.005:
        print("  ", k, "%.1f %%" % (100 * hist[k] / tot))
