#!/usr/bin/env python3
"""tools/ref_builtins_table.py — build container only (needs /root/reference).

What do the OpenCL builtins the reference kernel calls become when ROCm's OWN OpenCL tool chain builds
kernels/raytracer.cl for gfx950 (oracle/_ref_gfx950/, oracle/Makefile), and is that the IEEE-plain
sequence the oracle's pin assumes (oracle/ref_shim.cpp)?  Each builtin is compiled in a one-line OpenCL
kernel of its own, with the flags of the reference build and with -ffp-contract=off, and the ISA is
inspected.  Prints a markdown table (profiles/r02_ref_gfx950_builtins.md)."""
import os
import re
import subprocess
import sys
import tempfile

CLANG = "/opt/rocm/lib/llvm/bin/clang"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# builtin, OpenCL source of a probe kernel, what oracle/ref_shim.cpp defines it as
PROBES = [
    ("dot(float3, float3)", "o[0] = dot(vload3(0, a), vload3(1, a));", "(ax*bx + ay*by) + az*bz: 3 mul + 2 add, each rounded"),
    ("cross(float3, float3)", "vstore3(cross(vload3(0, a), vload3(1, a)), 0, o);", "ay*bz - az*by, ...: 6 mul + 3 sub, each rounded"),
    ("normalize(float3)", "vstore3(normalize(vload3(0, a)), 0, o);", "v / sqrt(dot(v, v)): IEEE sqrt, three IEEE divisions"),
    ("a / b (float)", "o[0] = a[0] / a[1];", "IEEE-754 correctly rounded division"),
    ("sqrt(float)", "o[0] = sqrt(a[0]);", "IEEE-754 correctly rounded square root"),
    ("mix(float3, float3, float)", "vstore3(mix(vload3(0, a), vload3(1, a), a[6]), 0, o);", "a + (b - a)*t, each op rounded"),
    ("min(float3, float3)", "vstore3(min(vload3(0, a), vload3(1, a)), 0, o);", "b < a ? b : a per component"),
    ("sign(float)", "o[0] = sign(a[0]);", "+-1, +-0, 0 for NaN"),
    ("pow(float, 5)", "o[0] = pow(a[0], 5);", "((x*x)*(x*x))*x (feeds a comparison only, raytracer.cl:404,423)"),
    ("(uint)fabs(float * double)", "o[0] = (float)(uint)fabs(a[0] * 438.5453);", "double product, truncation (raytracer.cl:114)"),
]


def isa(body, extra):
    src = "__kernel void probe(__global const float* a, __global float* o) { %s }\n" % body
    with tempfile.TemporaryDirectory() as d:
        cl, out = os.path.join(d, "p.cl"), os.path.join(d, "p.hsaco")
        open(cl, "w").write(src)
        subprocess.check_call([CLANG, "-x", "cl", "-cl-std=CL1.2", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950",
                               "-Xclang", "-finclude-default-header", "-O3"] + extra + [cl, "-o", out])
        dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", "--disassemble-symbols=probe", out], capture_output=True, text=True).stdout
    ops = re.findall(r"^\s+(v_\w+)", dis, re.M)
    keep = [re.sub(r"_(e32|e64)$", "", o) for o in ops if not re.match(r"v_(mov|readfirstlane|lshl|add_co|addc|accvgpr)", o)]
    return keep


def summary(ops):
    from collections import Counter
    c = Counter(ops)
    return ", ".join("%s×%d" % (k, v) if v > 1 else k for k, v in sorted(c.items()))


def verdict(name, ops):
    s = set(ops)
    fused = any(o.startswith(("v_fma", "v_fmac", "v_pk_fma", "v_mad_f32", "v_mac_f32")) for o in s)
    if name.startswith("dot"):
        return "fused multiply-add chain — NOT the shim's 3 mul + 2 add" if fused else "separate mul / add, as the shim"
    if name.startswith("normalize"):
        return "rsq-based (v_rsq_f32), no division — NOT the shim's sqrt + 3 IEEE divisions" if "v_rsq_f32" in s else "division-based"
    if name.startswith("a / b"):
        return ("IEEE sequence (v_div_scale / fmas / fixup), as the shim" if "v_div_scale_f32" in s else
                "reciprocal-based, 2.5 ulp (v_rcp_f32, no v_div_scale) — NOT IEEE division")
    if name.startswith("sqrt"):
        n = len(ops)
        return "hardware v_sqrt_f32 with range scaling, 3 ulp — NOT correctly rounded" if n < 12 else "correctly rounded expansion, as the shim"
    if name.startswith("pow"):
        return "exp2(y * log2(x)) family (v_log_f32 / v_exp_f32) — NOT the shim's multiply chain (only feeds a comparison)" if ("v_exp_f32" in s or "v_log_f32" in s) else "multiply chain"
    if name.startswith("cross"):
        return "fused (fma / packed)" if fused else "separate mul / sub, as the shim"
    if name.startswith("mix"):
        return "fused a + (b-a)*t (fma)" if fused else "separate sub / mul / add, as the shim"
    return "same result class as the shim (exact operation)"


def main():
    print("# The reference's OpenCL builtins as ROCm's OpenCL tool chain builds them for gfx950\n")
    print("`clang -x cl -cl-std=CL1.2 -target amdgcn-amd-amdhsa -mcpu=gfx950 -O3` (+ `-ffp-contract=off` in the last column), "
          "ROCm device libraries (opencl.bc, ocml.bc) linked — no stand-ins.  One probe kernel per builtin; moves and "
          "address arithmetic omitted.  The oracle's pin (`oracle/ref_shim.cpp`) defines every builtin as the plain "
          "IEEE sequence of the third column.\n")
    print("| builtin | VALU instructions (default flags) | the shim's definition | ROCm's implementation is … | with -ffp-contract=off |")
    print("|---|---|---|---|---|")
    for name, body, shim in PROBES:
        a, b = isa(body, []), isa(body, ["-ffp-contract=off"])
        print("| `%s` | %s | %s | %s | %s |" % (name, summary(a), shim, verdict(name, a), "same" if a == b else summary(b)))


if __name__ == "__main__":
    if not os.path.isdir("/root/reference"):
        sys.exit("build container only")
    main()
