#!/bin/bash
# A/B timing of environment settings on the 4K GOP inside one call: env_ab.sh "A=1" "A=2 B=3" ... (two rounds)
for round in 1 2; do for e in "$@"; do
  echo "== $e"
  env $e timeout -k 10 120 python tools/profile_gop.py --pictures 3 --reps 5 $AB_ARGS 2>&1 | grep -E "^pic" | sed -e 's/.*| //' || exit 1
done; done
