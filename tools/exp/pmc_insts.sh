#!/bin/bash
# instruction mix per kernel (rocprofv3 PMC, kernel trace only): tools/exp/pmc_insts.sh [env assignments...]
# prints per kernel: launches, waves, VALU / SALU / LDS / VMEM instructions per wave
set -e
export TMPDIR=/tmp
out=gpurun_out/pmc_insts
rm -rf $out; mkdir -p $out
env "$@" rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $out/a -- python3 tools/profile_gop.py --pictures 2 --reps 2 > $out/a.log 2>&1
env "$@" rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $out/b -- python3 tools/profile_gop.py --pictures 2 --reps 2 > $out/b.log 2>&1
python3 - <<'PY'
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for d in ("a", "b"):
    for f in glob.glob("gpurun_out/pmc_insts/%s/**/*counter_collection.csv" % d, recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = re.match(r"(?:void )?(?:d265::)?([A-Za-z_0-9]+)", r["Kernel_Name"]).group(1)
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if d == "a" and r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k in sorted(acc):
    if not k.startswith("k_"): continue
    a = acc[k]; w = max(a["SQ_WAVES"], 1)
    print("%-18s launches %3d waves/launch %8.0f  per wave: VALU %7.1f SALU %6.1f LDS %6.1f VMEM_RD %5.1f VMEM_WR %5.1f  wait_any/busy %.2f" % (
        k, n[k], w / max(n[k], 1), a["SQ_INSTS_VALU"] / w, a["SQ_INSTS_SALU"] / w, a["SQ_INSTS_LDS"] / w, a["SQ_INSTS_VMEM_RD"] / w, a["SQ_INSTS_VMEM_WR"] / w,
        a["SQ_WAIT_INST_ANY"] / max(a["SQ_BUSY_CYCLES"], 1)))
PY
find $out -name "*.csv" -delete
