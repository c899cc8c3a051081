#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory (gpurun_out/prof_<tag>/) into
profiles/<tag>_summary.md + profiles/<tag>_kernel_stats.csv (+ traffic.json entry).

usage: tools/summarize_profile.py <tag> [workload]
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
KERNELS = (sys.argv[3] if len(sys.argv) > 3 else "pt_prefix<false, false>;pt_samples_q<false, false>").split(";")
KERNEL = " + ".join(KERNELS)

lines = ["# rocprofv3 summary `%s` (workload %s)" % (tag, workload), ""]
def newest(pattern, window=1200.0):
    """gpurun merges every call's files into the same directory: keep those of the latest run only."""
    files = glob.glob(pattern)
    if not files:
        return []
    last = max(os.path.getmtime(f) for f in files)
    return sorted(f for f in files if last - os.path.getmtime(f) <= window)


stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"), 60.0)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as f:
        f.write(open(stats[0]).read())
    lines += ["## `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline`", "",
              "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| `%s` | %s | %.4f | %.4f | %.4f | %s |" % (r["Name"].split("(")[0][:70], r["Calls"],
                     float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Percentage"]))
    lines.append("")
bench = os.path.join(src, "bench_trace.json")
if os.path.isfile(bench) and os.path.getsize(bench):
    b = json.loads(open(bench).read().strip().splitlines()[-1])
    lines += ["bench.py under the tracer: value %.1f %s, kernel_ms (HIP events) %.4f, roofline.frac %.4f" %
              (b["value"], b["unit"], b["roofline"]["kernel_ms"], b["roofline"]["frac"]), ""]

per_kernel = {}
for f in newest(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        for k in KERNELS:
            if k.rstrip(">") in r["Kernel_Name"]:   # "pt_samples_q<false, true>" also matches "<false, true, 6>"
                per_kernel.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
# one trace call = one launch of each listed kernel: sum their per-launch means
counters = {}
for k, d in per_kernel.items():
    for name, v in d.items():
        counters.setdefault(name, []).append(sum(v) / len(v))
counters = {name: [sum(v)] for name, v in counters.items()}
if per_kernel:
    lines += ["## PMC per kernel (mean per dispatch)", "", "| kernel | counter | mean |", "|---|---|---|"]
    for k in KERNELS:
        for name in sorted(per_kernel.get(k, {})):
            v = per_kernel[k][name]
            lines.append("| `%s` | %s | %.6g |" % (k, name, sum(v) / len(v)))
    lines.append("")
if counters:
    lines += ["## PMC passes (separate `rocprofv3 --pmc ...` runs), summed over one launch of each of `%s`" % KERNEL,
              "", "| counter | per trace call | |", "|---|---|---|"]
    for k in sorted(counters):
        v = counters[k]
        lines.append("| %s | %.6g | %d |" % (k, sum(v) / len(v), len(v)))
    lines.append("")
    mean = {k: sum(v) / len(v) for k, v in counters.items()}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads
        # exactly 1/2 of the bytes of a wide coalesced stream (128-B requests tallied at 64 B).
        # This kernel's reads are scalar/gather traffic, not a wide stream, so the x2 is an
        # upper bound; both are reported.
        rd_raw = mean["FETCH_SIZE"] * 1024.0
        wr = mean["WRITE_SIZE"] * 1024.0
        lines += ["HBM traffic per launch: reads %.3f MB raw (%.3f MB with the gfx950 x2 correction), writes %.3f MB"
                  % (rd_raw / 1e6, 2 * rd_raw / 1e6, wr / 1e6), ""]
        tpath = os.path.join(dst, "traffic.json")
        t = json.load(open(tpath)) if os.path.isfile(tpath) else {}
        t[workload] = {"hbm_bytes_per_launch": int(2 * rd_raw + wr), "fetch_bytes_raw": int(rd_raw),
                       "fetch_bytes_corrected_x2": int(2 * rd_raw), "write_bytes": int(wr), "profile": tag}
        json.dump(t, open(tpath, "w"), indent=1, sort_keys=True)
    if "SQ_INSTS_VALU" in mean and "SQ_WAVES" in mean:
        lines += ["VALU instructions per wave: %.1f; SMEM per wave: %.1f; VMEM reads per wave: %.2f" %
                  (mean["SQ_INSTS_VALU"] / mean["SQ_WAVES"], mean.get("SQ_INSTS_SMEM", 0) / mean["SQ_WAVES"],
                   mean.get("SQ_INSTS_VMEM_RD", 0) / mean["SQ_WAVES"]), ""]
    if "SQ_THREAD_CYCLES_VALU" in mean and "SQ_INST_CYCLES_VALU" in mean and mean["SQ_INST_CYCLES_VALU"]:
        lines += ["SQ_THREAD_CYCLES_VALU / SQ_INST_CYCLES_VALU = %.2f active lanes per VALU issue cycle (64 = full)" %
                  (mean["SQ_THREAD_CYCLES_VALU"] / mean["SQ_INST_CYCLES_VALU"]), ""]
open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
