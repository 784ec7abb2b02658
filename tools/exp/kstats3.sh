#!/bin/bash
# rocprofv3 kernel durations of a 3-picture GOP (I + B with one reference + B with two): tools/exp/kstats3.sh [env assignments...]
export TMPDIR=/tmp
out=gpurun_out/kstats3; rm -rf $out; mkdir -p $out
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out/a -- python3 tools/profile_gop.py --pictures 3 --reps 6 --only 2 > $out/a.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/kstats3/a/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0].replace("void d265::", "")
        if n.startswith("k_mc"): print("%-30s calls %4s avg %9.1f us  min %9.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
