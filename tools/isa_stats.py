#!/usr/bin/env python3
"""tools/isa_stats.py [--arith K] [kernel-substring ...] — static ISA statistics of the shipped gfx950 kernels.

Compiles csrc/pt_kernels.hip for the device only (-S, the flags of __graft_entry__ for arithmetic policy K = 0 | 1 | 2,
default 2 = rocm-opencl, the policy bench.py times; ISA_ARITH in the environment does the same) and prints, per
kernel: VGPRs / SGPRs / scratch bytes / spilled VGPRs from the code-object metadata and the static opcode
histogram grouped into the issue-cost classes measured by tools/valu_microbench.hip
(profiles/r02_valu_microbench.md).  --json dumps everything for bench.py / profile summaries."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# Issue cost in cycles per wave64 instruction per SIMD, measured on MI355X with tools/valu_microbench.hip
# (profiles/r02_valu_microbench.md; 8 resident waves per SIMD, wall clock × measured shader clock, loop overhead
# included, rounded to the datapath's granularity): the SIMD is 32 lanes wide, so a full-rate wave64 instruction
# takes 2 cycles; a second group of opcodes runs at half rate, transcendentals at an eighth of the lanes.
FULL, HALF, TRANS = 2.0, 4.0, 8.0


def counter_group(op):
    """Which rocprofv3 SQ_INSTS_VALU_* class an opcode is assumed to be counted in ('other' = none of them)."""
    b = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if re.match(r"v_(rcp|rcp_iflag|rsq|sqrt|exp|log|sin|cos)_f32", b): return "TRANS_F32"
    if re.match(r"v_(add|sub|subrev)_f32", b): return "ADD_F32"
    if re.match(r"v_mul_f32", b): return "MUL_F32"
    if re.match(r"v_(fma|fmac|mac|mad|fmaak|fmamk)_f32", b): return "FMA_F32"
    if re.match(r"v_(add|mul|fma)_f64", b): return b[2:5].upper() + "_F64"
    if re.match(r"v_cvt_", b): return "CVT"
    if re.match(r"v_(lshl_add_u64|mad_u64_u32|mad_i64_i32|lshlrev_b64|lshrrev_b64|ashrrev_i64)", b): return "INT64"
    if re.match(r"v_(add|sub|subrev|addc|subb|subbrev)(_co)?_[ui]32|v_(and|or|xor|not)_b32|v_(lshlrev|lshrrev|ashrrev)_[bi]32|"
                r"v_mul_(lo|hi)_[ui]32|v_mul_[ui]32_[ui]24|v_mad_[ui]32_[ui]24|v_(lshl_add|add3|add_lshl|lshl_or|and_or|or3)_[ub]32|"
                r"v_bfe_|v_bfi_|v_(min|max)3?_[ui]32|v_bcnt|v_mbcnt|v_alignbit", b): return "INT32"
    return "other"


def issue_cost(op):
    """Measured issue cost of a VALU opcode (cycles per wave-instruction per SIMD)."""
    b = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if re.match(r"v_(rcp|rcp_iflag|rsq|sqrt|exp|log|sin|cos)_", b): return TRANS
    if re.match(r"v_(add|sub|subrev|mul|fma|fmac|mac|mad|fmaak|fmamk)_f32", b): return FULL
    if re.match(r"v_(add|sub|subrev)(_co)?_u32|v_(and|or|xor|not)_b32", b): return FULL
    if b.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr")): return 2.5      # 2.48 measured with distinct sources
    if b.startswith("v_cndmask_b32"): return 2.2                               # v_cmp + v_cndmask pair 6.5, v_cmp alone 4.3
    return HALF   # f64, packed f32, integer multiply / shifts / 3-operand integer, compares, conversions, min / max,
                  # floor, v_div_scale / fmas / fixup, lane reads and writes: 4.2–4.4 measured


def cost_class(op):
    """→ (class name, cycles) for a VALU opcode; None for non-VALU."""
    if not op.startswith("v_"):
        return None
    c = issue_cost(op)
    return ("trans" if c == TRANS else "half" if c == HALF else "full", c)


def group_costs(ops):
    """Static average issue cost per counter group of one kernel's opcode histogram: {group: (instructions, avg cost)}."""
    n, cyc = collections.Counter(), collections.Counter()
    for op, k in ops.items():
        if op.startswith("v_"):
            g = counter_group(op)
            n[g] += k
            cyc[g] += k * issue_cost(op)
    return {g: (n[g], cyc[g] / n[g]) for g in n}


def default_arith():
    for i, a in enumerate(sys.argv):
        if a == "--arith" and i + 1 < len(sys.argv):
            return int(sys.argv[i + 1])
    return int(os.environ.get("ISA_ARITH", "2"))


def device_asm(force=False, arith=None):
    import __graft_entry__ as g
    arith = default_arith() if arith is None else int(arith)
    extra = os.environ.get("ISA_EXTRA_FLAGS", "").split()   # e.g. ISA_EXTRA_FLAGS="-DPT_Q_BLOCK_WAVES=1" for a variant
    tag = ("_" + "_".join(e.lstrip("-D").replace("=", "") for e in extra)) if extra else ""
    out = os.path.join(ROOT, "build", "pt_kernels_a%d_gfx950%s.s" % (arith, tag))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = g.hip_sources()
    if force or not os.path.isfile(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs):
        flags = [f for f in g.HIP_FLAGS if f not in ("-shared", "-fPIC")] + g.POLICY_FLAGS[arith]
        subprocess.check_call([g.HIPCC] + flags + extra + ["--cuda-device-only", "-S", srcs[1], "-o", out],
                              stderr=subprocess.DEVNULL)
    return out


def demangle(names):
    p = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
    return [re.sub(r"\(.*", "", l.replace("void ", "")) for l in p.stdout.strip().split("\n")]


def kernels(path):
    text = open(path).read()
    meta = {}
    md = text[text.index("amdhsa.kernels:"):] if "amdhsa.kernels:" in text else ""
    for blk in re.split(r"^  - \.", md, flags=re.M)[1:]:
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm:
            continue
        get = lambda k: int(re.search(r"%s:\s+(\d+)" % k, blk).group(1)) if re.search(r"%s:\s+(\d+)" % k, blk) else 0
        meta[nm.group(1)] = dict(vgpr=get("vgpr_count"), sgpr=get("sgpr_count"), scratch=get("private_segment_fixed_size"),
                                 spilled_vgprs=get("vgpr_spill_count"), lds_static=get("group_segment_fixed_size"))
    out = collections.OrderedDict()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if name not in meta:
            continue
        ops = collections.Counter()
        for line in body.split("\n"):
            t = line.strip().split()
            if t and re.match(r"^[vs]_|^ds_|^global_|^scratch_|^buffer_|^flat_", t[0]):
                ops[t[0]] += 1
        out[name] = dict(meta[name], ops=ops)
    names = list(out)
    for n, d in zip(names, demangle(names)):
        out[n]["demangled"] = d
    return out


def summarize(k):
    ops = k["ops"]
    cls = collections.Counter()
    cyc = collections.Counter()
    for op, n in ops.items():
        c = cost_class(op)
        if c:
            cls[c[0]] += n
            cyc[c[0]] += n * c[1]
    valu = sum(cls.values())
    return dict(valu=valu, classes=dict(cls), cycles=dict(cyc),
                salu=sum(n for o, n in ops.items() if o.startswith("s_") and not o.startswith(("s_load", "s_waitcnt", "s_nop", "s_buffer"))),
                smem=sum(n for o, n in ops.items() if o.startswith(("s_load", "s_buffer"))),
                vmem=sum(n for o, n in ops.items() if o.startswith(("global_", "buffer_", "flat_"))),
                scratch_ops=sum(n for o, n in ops.items() if o.startswith("scratch_")),
                lds=sum(n for o, n in ops.items() if o.startswith("ds_")))


if __name__ == "__main__":
    args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--arith"]
    ks = kernels(device_asm("--force" in sys.argv))
    if "--json" in sys.argv:
        print(json.dumps({k["demangled"]: dict(summarize(k), vgpr=k["vgpr"], sgpr=k["sgpr"], scratch=k["scratch"],
                                               spilled_vgprs=k["spilled_vgprs"]) for k in ks.values()}, indent=1))
        sys.exit(0)
    for k in ks.values():
        if args and not any(a in k["demangled"] for a in args):
            continue
        s = summarize(k)
        print("%-44s vgpr %3d sgpr %3d scratch %3d B (%d spilled)  VALU %5d  SALU %5d  SMEM %3d  VMEM %3d  scratch-ops %3d  LDS %3d" %
              (k["demangled"], k["vgpr"], k["sgpr"], k["scratch"], k["spilled_vgprs"], s["valu"], s["salu"], s["smem"],
               s["vmem"], s["scratch_ops"], s["lds"]))
        if args:
            print("   classes:", {c: "%d (%.0f%%)" % (n, 100.0 * n / s["valu"]) for c, n in s["classes"].items()})
            top = [(o, n) for o, n in k["ops"].most_common() if o.startswith("v_")]
            print("   " + "  ".join("%s:%d" % (o, n) for o, n in top))
