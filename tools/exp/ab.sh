#!/bin/bash
# A/B timing of library variants on the 4K GOP inside one call: ab.sh name1 name2 ... (tools/exp/lib_<name>.so), two rounds
for round in 1 2; do for n in "$@"; do
  echo "== $n"
  DE265HIP_SO=tools/exp/lib_$n.so timeout -k 10 120 python tools/profile_gop.py --pictures 3 --reps 5 2>&1 | grep -E "^pic" | sed -e 's/.*| //' || exit 1
done; done
