#!/usr/bin/env python3
"""tools/microbench_table.py [json] → markdown tables of tools/valu_microbench.hip's output."""
import collections
import json
import sys

d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/valu_microbench.json"))
rows = d["results"]
print("# VALU issue microbenchmark — %s (%s), %d CUs\n" % (d["device"], d["arch"], d["cus"]))
print("Cycles one SIMD spends per wave-instruction (median over waves of Δs_memtime ÷ instructions ÷ resident waves per "
      "SIMD); `wall` = the same from the hipEvent time × measured shader clock.  %d loops × %d instructions per wave.\n"
      % (d["loops"], d["body"]))
t = collections.OrderedDict()
for r in rows:
    if r["lanes"] == "all 64":
        t.setdefault(r["inst"], {})[r["waves_per_simd"]] = r
ws = (1, 2, 4, 6, 8)
print("| instruction | " + " | ".join("%d wave%s/SIMD" % (w, "" if w == 1 else "s") for w in ws) + " | wall @8 | clock MHz @8 |")
print("|---|" + "---|" * (len(ws) + 2))
for k, v in t.items():
    print("| `%s` | " % k + " | ".join("%.2f" % v[w]["cyc_per_inst_simd"] for w in ws) +
          " | %.2f | %d |" % (v[8]["wall_cyc_per_inst_simd"], v[8]["clock_mhz"]))
print("\n## Partial EXEC masks (divergence)\n")
print("| instruction | lanes active | 1 wave/SIMD | 4 waves/SIMD |")
print("|---|---|---|---|")
m = collections.OrderedDict()
for r in rows:
    if r["lanes"] != "all 64":
        m.setdefault((r["inst"], r["lanes"]), {})[r["waves_per_simd"]] = r["cyc_per_inst_simd"]
for (inst, lanes), v in m.items():
    print("| `%s` | %s | %.2f | %.2f |" % (inst, lanes, v.get(1, 0), v.get(4, 0)))
