#!/usr/bin/env python3
"""tools/valu_busy_calib.py <dir> — reads the rocprofv3 counter CSVs tools/valu_busy_calib.sh wrote for
tools/valu_microbench.hip and prints, per instruction loop at 8 waves per SIMD: the wall-clock cost per
wave-instruction per SIMD (the microbenchmark's own figure) next to SQ_ACTIVE_INST_VALU per instruction and the busy
fraction that counter implies — the calibration of the counter's unit."""
import csv, glob, json, os, sys, collections

d = sys.argv[1]
mb = json.load(open(os.path.join(d, "microbench.json")))
cus = mb["cus"]
simds = cus * 4
# the microbenchmark's dispatches, in launch order: every result line = 2 dispatches (warm-up + timed) of one kernel
order = [(r["inst"], r["waves_per_simd"], r["lanes"], r["wall_cyc_per_inst_simd"], r["clock_mhz"], r["ms"]) for r in mb["results"]]


def load(sub):
    per = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(d, sub, "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if not r["Kernel_Name"].startswith("k_"):
                continue
            k = int(r["Dispatch_Id"])
            per.setdefault(k, {"name": r["Kernel_Name"]})
            per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [per[k] for k in sorted(per)]


rows = load("pmc")
assert len(rows) == 2 * len(order), (len(rows), len(order))
rows2 = load("pmc2") if os.path.isdir(os.path.join(d, "pmc2")) else None
out = []
print("| loop | lanes | waves/SIMD | wall cycles per instr per SIMD | SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU | ACTIVE_INST_VALU x 4 / (SIMDs x kernel cycles) | SQ_THREAD_CYCLES_VALU / INSTS | SQ_INST_CYCLES_VALU / INSTS |")
print("|---|---|---|---|---|---|---|---|")
for i, (inst, w, lanes, wall, mhz, ms) in enumerate(order):
    c = rows[2 * i + 1]
    n = c.get("SQ_INSTS_VALU", 0.0)
    if not n or w not in (1, 8):
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    a = c["SQ_ACTIVE_INST_VALU"]
    ic = rows2[2 * i + 1].get("SQ_INST_CYCLES_VALU") / rows2[2 * i + 1]["SQ_INSTS_VALU"] if rows2 else None
    rec = {"inst": inst, "lanes": lanes, "waves_per_simd": w, "wall_cyc_per_inst_simd": wall, "active_per_inst": a / n,
           "busy_x4": a * 4.0 / (simds * cyc), "thread_cycles_per_inst": c["SQ_THREAD_CYCLES_VALU"] / n, "inst_cycles_per_inst": ic,
           "kernel_cycles": cyc, "insts": n}
    out.append(rec)
    print("| `%s` | %s | %d | %.2f | %.3f | %.3f | %.1f | %s |" % (inst, lanes, w, wall, a / n, rec["busy_x4"], rec["thread_cycles_per_inst"],
                                                                   "%.3f" % ic if ic is not None else "—"))
json.dump(out, open(os.path.join(d, "valu_busy_calib.json"), "w"), indent=1)
