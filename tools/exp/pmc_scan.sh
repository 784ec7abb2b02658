#!/bin/bash
# instruction mix of the scan kernels per launch (first launch: the all-intra picture, second: a B picture)
set -e
export TMPDIR=/tmp
out=gpurun_out/pmc_scan
rm -rf $out; mkdir -p $out
env "$@" rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $out/a -- python3 tools/profile_gop.py --pictures 2 --reps 1 > $out/a.log 2>&1
env "$@" rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $out/b -- python3 tools/profile_gop.py --pictures 2 --reps 1 > $out/b.log 2>&1
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out/c -- python3 tools/profile_gop.py --pictures 2 --reps 1 > $out/c.log 2>&1
python3 - <<'PY'
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("a", "b"):
    for f in glob.glob("gpurun_out/pmc_scan/%s/**/*counter_collection.csv" % d, recursive=True):
        order = collections.Counter(); seen = {}
        for r in csv.DictReader(open(f)):
            k = re.match(r"(?:void )?(?:d265::)?([A-Za-z_0-9]+)", r["Kernel_Name"]).group(1)
            if not k.startswith("k_scan"): continue
            did = r["Dispatch_Id"]
            if did not in seen: seen[did] = order[k]; order[k] += 1
            acc[(k, seen[did])][r["Counter_Name"]] += float(r["Counter_Value"])
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_scan/c/**/*kernel_trace.csv", recursive=True):
    for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
        k = re.match(r"(?:void )?(?:d265::)?([A-Za-z_0-9]+)", r["Kernel_Name"]).group(1)
        if k.startswith("k_scan"): dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, i) in sorted(acc):
    a = acc[(k, i)]; w = max(a["SQ_WAVES"], 1)
    d = dur[k][i] if i < len(dur[k]) else 0
    print("%-14s launch %d (%s) %7.1f us waves %6.0f  per wave: VALU %7.1f SALU %7.1f LDS %6.1f VMEM_RD %6.1f VMEM_WR %6.1f SMEM %5.1f  cycles %9.0f" % (
        k, i, "I" if i == 0 else "B", d, w, a["SQ_INSTS_VALU"] / w, a["SQ_INSTS_SALU"] / w, a["SQ_INSTS_LDS"] / w, a["SQ_INSTS_VMEM_RD"] / w, a["SQ_INSTS_VMEM_WR"] / w,
        a["SQ_INSTS_SMEM"] / w, a["SQ_WAVE_CYCLES"] / w))
PY
find $out -name "*.csv" -delete
