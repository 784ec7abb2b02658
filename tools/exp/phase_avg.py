"""Average the DE265HIP_BUILD_TIMING lines of a run by picture kind (I: tu_scan > 9 ms): python tools/exp/phase_avg.py file"""
import re, sys, collections
acc = {"I": collections.defaultdict(float), "B": collections.defaultdict(float)}; n = {"I": 0, "B": 0}
for line in open(sys.argv[1]):
    if not line.startswith("de265hip build:"): continue
    ph = {k: float(v) for k, v in re.findall(r"(\w+)=([0-9.]+)ms", line)}
    kind = "I" if ph.get("mc", 0) < 0.05 else "B"
    n[kind] += 1
    for k, v in ph.items(): acc[kind][k] += v
for kind in ("I", "B"):
    if n[kind]:
        print(kind, n[kind], "builds:", " ".join("%s=%.2f" % (k, v / n[kind]) for k, v in acc[kind].items()), "total=%.2f" % (sum(acc[kind].values()) / n[kind]))
