#!/bin/bash
# where k_mc's wavefront time goes: more SQ counters (rocprofv3 PMC, kernel trace only), per wavefront
set -e
export TMPDIR=/tmp
out=gpurun_out/pmc_mc
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  env "$@" rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/profile_gop.py --pictures 2 --reps 2 > $out/p$i.log 2>&1 || echo "pass $i failed: $set"
done
python3 - <<'PY'
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_mc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.match(r"(?:void )?(?:d265::)?([A-Za-z_0-9]+)", r["Kernel_Name"]).group(1)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc):
    if not k.startswith("k_"): continue
    a = acc[k]; w = max(a["SQ_WAVES"], 1)
    print(k, "waves %.0f" % w, " ".join("%s=%.1f" % (c.replace("SQ_", ""), v / w) for c, v in sorted(a.items()) if c != "SQ_WAVES"))
PY
find $out -name "*.csv" -delete
