#!/bin/bash
t0=$(date +%s)
python bench.py 2> gpurun_out/bench_final.err > gpurun_out/bench_final.json
t1=$(date +%s)
echo "bench wall seconds: $((t1-t0))"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
print(d["value"], d["device_replay"]["value"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d.get("parity_vs_oracle"))
PY
