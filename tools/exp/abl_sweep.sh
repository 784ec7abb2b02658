#!/bin/bash
# timing-only ablation sweep of k_run on the 4K GOP (results of ablated runs are invalid pictures; only times count)
for d in ${ABL_BITS:-0 4 68 128 256}; do
  echo "== DE265HIP_DEBUG=$d"
  DE265HIP_SO=tools/exp/lib_abl.so DE265HIP_DEBUG=$d timeout -k 10 120 python tools/profile_gop.py --pictures 2 --reps 5 $ABL_ARGS 2>&1 | grep -E "^pic" | sed -e 's/.*| //' || exit 1
done
