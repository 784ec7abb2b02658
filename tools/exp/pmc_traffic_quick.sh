#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel launch of a 2-picture GOP (rocprofv3 PMC, kernel trace only; two passes)
set -e
export TMPDIR=/tmp
out=gpurun_out/pmc_quick
rm -rf $out; mkdir -p $out
env "$@" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f -- python3 tools/profile_gop.py --pictures 2 --reps 2 > $out/f.log 2>&1
env "$@" rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/w -- python3 tools/profile_gop.py --pictures 2 --reps 2 > $out/w.log 2>&1
python3 - <<'PY'
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for d in ("f", "w"):
    for f in glob.glob("gpurun_out/pmc_quick/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.match(r"(?:void )?(?:d265::)?([A-Za-z_0-9]+)", r["Kernel_Name"]).group(1)
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    if not k.startswith("k_"): continue
    print("%-18s launches %3d  FETCH_SIZE %8.1f KB/launch (x2 on gfx950: %8.1f)  WRITE_SIZE %8.1f KB/launch" % (
        k, n[k]["FETCH_SIZE"], acc[k]["FETCH_SIZE"] / max(n[k]["FETCH_SIZE"], 1), 2 * acc[k]["FETCH_SIZE"] / max(n[k]["FETCH_SIZE"], 1),
        acc[k]["WRITE_SIZE"] / max(n[k]["WRITE_SIZE"], 1)))
PY
find $out -name "*.csv" -delete
