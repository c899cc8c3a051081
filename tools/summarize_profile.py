#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory (gpurun_out/prof_<tag>/) into
    profiles/<tag>_summary.md        kernel table, PMC per kernel, the VALU-issue roofline worked out by hand
    profiles/<tag>_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the bench command
    profiles/valu_mix.json[workload] what bench.py prices its live kernel time against
    profiles/traffic.json[workload]  HBM bytes per launch (FETCH_SIZE ×2 per the gfx950 correction, + WRITE_SIZE)

usage: tools/summarize_profile.py <tag> <workload> "<kernel substring>;<kernel substring>"
(the profile key — workload@WxHxspp — and the arithmetic policy are read from the bench line the profile was taken with)

The roofline (DESIGN.md §5 "Roofline"): the trace kernels are bound by VALU ISSUE, not by HBM.  A gfx950 SIMD
is 32 lanes wide: a full-rate wave64 instruction occupies it for 2 cycles, a half-rate one for 4, a
transcendental for 8 (tools/valu_microbench.hip, profiles/r02_valu_microbench.md).  rocprofv3 counts the issued
instructions per class (SQ_INSTS_VALU_*); the classes it has no counter for ("other": moves, selects, compares,
min/max, lane reads, division helpers) and the INT32 class (which mixes full- and half-rate opcodes) are priced
with the kernel's own static opcode mix (tools/isa_stats.py).  issue cycles ÷ (kernel cycles × 1024 SIMDs) is
the fraction of the chip's VALU issue slots the kernel fills — at most 1."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import isa_stats  # noqa: E402

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
KERNELS = (sys.argv[3] if len(sys.argv) > 3 else "pt_prefix<false, false>;pt_samples_q<false, false").split(";")
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
SIMDS = 1024          # 256 CUs × 4
PEAK_CLOCK_GHZ = 2.4  # MI355X_MICROARCH.md
HBM_PEAK = 8.0e12

ARITH = {"ieee": 0, "rocm-opencl-nocontract": 1, "rocm-opencl": 2}
bench_line = {}
_bl = os.path.join(src, "bench_trace.json")
if os.path.isfile(_bl) and os.path.getsize(_bl):
    try:
        bench_line = json.loads([l for l in open(_bl).read().strip().splitlines() if l.startswith("{")][-1])
    except Exception:
        bench_line = {}
arith = bench_line.get("arith", "rocm-opencl")
key = bench_line.get("config", {}).get("profile_key", workload)
NS = "pt_a%d::" % ARITH[arith]

lines = ["# rocprofv3 summary `%s` (workload %s, arithmetic policy %s)" % (tag, key, arith), ""]


def newest(pattern, window=1500.0):
    """gpurun merges every call's files into the same directory: keep those of the latest run only."""
    files = glob.glob(pattern)
    if not files:
        return []
    last = max(os.path.getmtime(f) for f in files)
    return sorted(f for f in files if last - os.path.getmtime(f) <= window)


def short(name):
    return name.replace("void ", "").split("(")[0]


kernel_ms = {}
stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"), 60.0)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as f:
        f.write(open(stats[0]).read())
    lines += ["## `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity-check`", "",
              "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| `%s` | %s | %.4f | %.4f | %.4f | %s |" % (short(r["Name"])[:70], r["Calls"],
                     float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Percentage"]))
        kernel_ms[short(r["Name"])] = float(r["AverageNs"]) / 1e6
    lines.append("")
bench = os.path.join(src, "bench_trace.json")
if os.path.isfile(bench) and os.path.getsize(bench):
    try:
        b = bench_line
        lines += ["bench.py under the tracer: value %.1f %s, kernel_ms (HIP events, whole trace call) %.4f" %
                  (b["value"], b["unit"], b["roofline"]["kernel_ms"]), ""]
    except Exception:
        pass

per_kernel = {}
for f in newest(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = short(r["Kernel_Name"])
        for k in KERNELS:
            counting = name.startswith((NS + "pt_prefix<true", NS + "pt_samples_q<true", NS + "pt_samples<true", NS + "pt_render<0, true",
                                        NS + "pt_render<1, true", NS + "pt_render<2, true"))   # COUNT builds: bench's untimed counting pass
            if name.startswith(NS + k.rstrip(">")) and not counting:
                per_kernel.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
per_kernel = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in per_kernel.items()}

if per_kernel:
    lines += ["## PMC per kernel (mean per dispatch; separate `rocprofv3 --pmc` passes of the same bench command)", ""]
    names = sorted(per_kernel)
    counters = sorted({c for d in per_kernel.values() for c in d})
    lines += ["| counter | " + " | ".join("`%s`" % n for n in names) + " |", "|---|" + "---|" * len(names)]
    for c in counters:
        lines.append("| %s | " % c + " | ".join("%.5g" % per_kernel[n][c] if c in per_kernel[n] else "" for n in names) + " |")
    lines.append("")

isa = {k["demangled"]: k for k in isa_stats.kernels(isa_stats.device_asm(arith=ARITH[arith])).values()}
mix_out = {}
for name, m in sorted(per_kernel.items()):
    if "SQ_INSTS_VALU" not in m or "GRBM_GUI_ACTIVE" not in m:
        continue
    k = isa.get(name)
    if not k:
        continue
    gc = isa_stats.group_costs(k["ops"])
    counted = {"ADD_F32": 2.0, "MUL_F32": 2.0, "FMA_F32": 2.0, "TRANS_F32": 8.0, "ADD_F64": 4.0, "MUL_F64": 4.0,
               "FMA_F64": 4.0, "TRANS_F64": 16.0, "INT64": 4.0, "CVT": 4.0}
    total = m["SQ_INSTS_VALU"]
    rows, cyc_est, cyc_low, cyc_high, known = [], 0.0, 0.0, 0.0, 0.0
    for g, cost in counted.items():
        n = m.get("SQ_INSTS_VALU_" + g, 0.0)
        if n:
            rows.append((g, n, cost, cost, cost))
            cyc_est += n * cost; cyc_low += n * cost; cyc_high += n * cost
            known += n
    n32 = m.get("SQ_INSTS_VALU_INT32", 0.0)
    c32 = gc.get("INT32", (0, 3.0))[1]
    rows.append(("INT32 (add/logic 2, shift/mul/3-operand 4; static mix)", n32, 2.0, c32, 4.0))
    cyc_est += n32 * c32; cyc_low += n32 * 2.0; cyc_high += n32 * 4.0
    other = max(total - known - n32, 0.0)
    co = gc.get("other", (0, 3.0))[1]
    rows.append(("other = SQ_INSTS_VALU − classes (mov 2.5, select 2.2, compare / min / max / lane / div helpers 4; static mix)",
                 other, 2.0, co, 4.0))
    cyc_est += other * co; cyc_low += other * 2.0; cyc_high += other * 4.0
    cycles = m["GRBM_GUI_ACTIVE"] / 8.0   # rocprofv3 sums the 8 XCDs
    ms = kernel_ms.get(name)
    slots = cycles * SIMDS
    lanes = m.get("SQ_THREAD_CYCLES_VALU", 0.0) / total if m.get("SQ_THREAD_CYCLES_VALU") else None
    lines += ["## VALU-issue roofline of `%s`" % name, "",
              "| class | wave-instructions per launch | cost low | cost used | cost high |", "|---|---|---|---|---|"]
    for g, n, lo, est, hi in rows:
        lines.append("| %s | %.5g (%.1f %%) | %.1f | %.2f | %.1f |" % (g, n, 100.0 * n / total, lo, est, hi))
    lines += ["",
              "* SQ_INSTS_VALU = %.5g wave-instructions; issue cycles = Σ count × cost = **%.4g** (bounds %.4g … %.4g)" % (total, cyc_est, cyc_low, cyc_high),
              "* kernel cycles = GRBM_GUI_ACTIVE / 8 = %.4g%s; issue slots = cycles × %d SIMDs = %.4g" %
              (cycles, (" (%.4f ms → %.2f GHz)" % (ms, cycles / ms / 1e6)) if ms else "", SIMDS, slots),
              "* **VALU issue fraction = %.3f** (bounds %.3f … %.3f); %.2f issue cycles and %.2f kernel cycles per VALU instruction per SIMD" %
              (cyc_est / slots, cyc_low / slots, cyc_high / slots, cyc_est / total, slots / total)]
    if lanes:
        lines.append("* lanes active per VALU instruction: SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = %.1f of 64 → useful-lane fraction of the issue slots %.3f" %
                     (lanes, cyc_est / slots * lanes / 64.0))
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"] * 4.0
        lines.append("* resident waves per SIMD (SQ_WAVE_CYCLES × 4 ÷ issue slots): %.2f; waiting on memory (SQ_WAIT_ANY) %.0f %%, "
                     "on issue (SQ_WAIT_INST_ANY) %.0f %%, executing (SQ_ACTIVE_INST_ANY) %.0f %% of wave time" %
                     (wc / slots, 100 * m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"],
                      100 * m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"]))
    hbm = None
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        rd_raw, wr = m["FETCH_SIZE"] * 1024.0, m["WRITE_SIZE"] * 1024.0
        hbm = 2 * rd_raw + wr
        t = (ms * 1e-3) if ms else cycles / (PEAK_CLOCK_GHZ * 1e9)
        lines.append("* HBM: FETCH_SIZE %.3f MB raw (×2 on gfx950 = %.3f MB) + WRITE_SIZE %.3f MB = %.3f MB per launch → %.1f GB/s = %.4f of 8 TB/s" %
                     (rd_raw / 1e6, 2 * rd_raw / 1e6, wr / 1e6, hbm / 1e6, hbm / t / 1e9, hbm / t / HBM_PEAK))
    lines.append("")
    mix_out[name] = {"valu_insts": total, "issue_cycles": cyc_est, "issue_cycles_low": cyc_low, "issue_cycles_high": cyc_high,
                     "kernel_cycles": cycles, "kernel_ms_profiled": ms, "lanes_per_inst": lanes, "hbm_bytes": hbm,
                     "classes": {g: n for g, n, _, _, _ in rows},
                     "vgpr": k["vgpr"], "scratch_bytes": k["scratch"], "spilled_vgprs": k["spilled_vgprs"]}

if mix_out:
    path = os.path.join(dst, "valu_mix.json")
    allmix = json.load(open(path)) if os.path.isfile(path) else {}
    import __graft_entry__ as g
    # the device sources the counts were measured on: bench.py compares (profile_matches_source)
    allmix[key] = {"profile": tag, "arith": arith, "kernels": mix_out, "source_digest": g.device_source_digest()}
    json.dump(allmix, open(path, "w"), indent=1, sort_keys=True)
    hb = [v["hbm_bytes"] for v in mix_out.values() if v["hbm_bytes"]]
    if hb:
        tpath = os.path.join(dst, "traffic.json")
        t = json.load(open(tpath)) if os.path.isfile(tpath) else {}
        t[key] = {"hbm_bytes_per_launch": int(sum(hb)), "profile": tag,
                       "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, summed over the kernels of one trace call"}
        json.dump(t, open(tpath, "w"), indent=1, sort_keys=True)

open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
