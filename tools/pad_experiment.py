#!/usr/bin/env python3
"""tools/pad_experiment.py <dir> — reads what tools/pad_experiment.sh measured: per padding N the sample kernel's
average duration (rocprofv3 --kernel-trace) and its SQ_INSTS_VALU per launch, and the issue cost of a full-rate logic
instruction from the microbenchmark; prints the slope  d(kernel cycles per SIMD) / d(VALU instructions per SIMD)  — the issue cost the padding
really paid — next to the instruction's stand-alone cost."""
import csv, glob, json, os, sys

d = sys.argv[1]
mb = json.load(open(os.path.join(d, "microbench.json")))
nop = {r["waves_per_simd"]: (r["wall_cyc_per_inst_simd"], r["clock_mhz"]) for r in mb["results"] if r["inst"] == "v_and_b32" and r["lanes"] == "all 64"}
simds = mb["cus"] * 4
rows = []
for n in (4, 32, 64, 128):
    dur, calls = None, 0
    for f in glob.glob(os.path.join(d, "trace%d" % n, "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if "pt_samples_q" in r["Name"]:
                dur, calls = float(r["AverageNs"]), int(r["Calls"])
    insts = cyc = 0.0
    k = 0
    for f in glob.glob(os.path.join(d, "pmc%d" % n, "*", "*_counter_collection.csv")):
        per = {}
        for r in csv.DictReader(open(f)):
            if "pt_samples_q" in r["Kernel_Name"]:
                per.setdefault(r["Dispatch_Id"], {}).setdefault(r["Counter_Name"], 0.0)
                per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        k = len(per)
        insts = sum(p["SQ_INSTS_VALU"] for p in per.values()) / k
        cyc = sum(p["GRBM_GUI_ACTIVE"] for p in per.values()) / k / 8.0
    rows.append({"pad": n, "kernel_ms": dur * 1e-6, "calls": calls, "insts_valu": insts, "pmc_kernel_cycles": cyc, "pmc_launches": k})
base = rows[0]
print("v_and_b32 (same class as v_or_b32) stand-alone (wall cycles per wave-instruction per SIMD): " + ", ".join("%d waves/SIMD %.2f" % (w, nop[w][0]) for w in sorted(nop)))
print("| extra v_or_b32 per iteration | kernel ms (rocprofv3 avg) | SQ_INSTS_VALU per launch | added instructions per SIMD | added time, ns | added cycles per added instruction (at the timed run's clock = cycles / ms of the PMC run) |")
print("|---|---|---|---|---|---|")
for r in rows:
    dn = (r["insts_valu"] - base["insts_valu"]) / simds
    dt = (r["kernel_ms"] - base["kernel_ms"]) * 1e6
    # clock: the PMC run's cycles over the PMC run's duration is not available per launch; use 2.4 GHz nominal and the
    # ratio of kernel cycles (PMC run) as a cross-check
    r["added_per_simd"], r["added_ns"] = dn, dt
    r["cyc_per_added_inst_2p4"] = dt * 2.4 / dn if dn else None
    r["pmc_cyc_per_added_inst"] = (r["pmc_kernel_cycles"] - base["pmc_kernel_cycles"]) / dn if dn else None
    print("| %d | %.4f | %.4e | %.0f | %.0f | %s (PMC run, counted cycles: %s) |" % (
        r["pad"], r["kernel_ms"], r["insts_valu"], dn, dt, "%.2f at 2.4 GHz" % r["cyc_per_added_inst_2p4"] if dn else "—",
        "%.2f" % r["pmc_cyc_per_added_inst"] if dn else "—"))
json.dump({"v_and_b32": {str(w): nop[w][0] for w in nop}, "rows": rows}, open(os.path.join(d, "pad_experiment.json"), "w"), indent=1)
