#!/usr/bin/env python3
"""tools/arith_probe.py — offline (no GPU): do the ROCm-OpenCL arithmetic policies of the HIP kernels (csrc/pt_arith.hpp,
PT_ARITH=1) compile to the same floating-point instruction sequences as ROCm's OpenCL tool chain produces for the OpenCL
builtins the reference calls?

One probe kernel per builtin, written twice: in OpenCL C (compiled `clang -x cl -target amdgcn-amd-amdhsa -mcpu=gfx950
-O3 -ffp-contract=off` with ROCm's opencl.bc / ocml.bc linked — the reference's build) and in HIP on top of
csrc/pt_arith.hpp (compiled with the policy's flags, __graft_entry__.POLICY_FLAGS[1]).  The float arithmetic opcodes of
the two ISA streams are compared as multisets, a packed instruction counting as two scalar ones (this build passes
-fno-slp-vectorize); integer / select / compare glue (how `all(v == 0)` or a sign test is spelled) is not compared.  The
bit-for-bit check against the code object itself runs on the GPU (tests/test_gpu_ref950.py); this is the cheap early
warning that a tool-chain update changed an expansion.  Prints a table; exit status 1 on a difference."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLANG = "/opt/rocm/lib/llvm/bin/clang"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

CL = r'''
__kernel void p_dot(__global const float* a, __global float* o) { o[0] = dot(vload3(0, a), vload3(1, a)); }
__kernel void p_cross(__global const float* a, __global float* o) { vstore3(cross(vload3(0, a), vload3(1, a)), 0, o); }
__kernel void p_normalize(__global const float* a, __global float* o) { vstore3(normalize(vload3(0, a)), 0, o); }
__kernel void p_div(__global const float* a, __global float* o) { o[0] = a[0] / a[1]; }
__kernel void p_rcp(__global const float* a, __global float* o) { o[0] = 1.0f / a[0]; }
__kernel void p_sqrt(__global const float* a, __global float* o) { o[0] = sqrt(a[0]); }
__kernel void p_mix(__global const float* a, __global float* o) { vstore3(mix(vload3(0, a), vload3(1, a), a[6]), 0, o); }
__kernel void p_min(__global const float* a, __global float* o) { vstore3(min(vload3(0, a), vload3(1, a)), 0, o); }
__kernel void p_sign(__global const float* a, __global float* o) { o[0] = sign(a[0]); }
__kernel void p_pow(__global const float* a, __global float* o) { o[0] = pow(a[0], 5); }
__kernel void p_hash(__global const float* a, __global float* o) { o[0] = (float)(uint)fabs(dot(vload3(0, a), (float3)(123.9898, 348.233, 433.3314)) * 438.5453); }
'''
HIP = r'''
#include "%s"
using namespace PT_NS;
#define LD(i) mk(a[3 * (i)], a[3 * (i) + 1], a[3 * (i) + 2])
#define ST(v) do { V3 t_ = (v); o[0] = t_.x; o[1] = t_.y; o[2] = t_.z; } while (0)
extern "C" {
__global__ void p_dot(const float *a, float *o) { o[0] = dot(LD(0), LD(1)); }
__global__ void p_cross(const float *a, float *o) { ST(cross(LD(0), LD(1))); }
__global__ void p_normalize(const float *a, float *o) { ST(normalize(LD(0))); }
__global__ void p_div(const float *a, float *o) { o[0] = a[0] / a[1]; }
__global__ void p_rcp(const float *a, float *o) { o[0] = 1.0f / a[0]; }
__global__ void p_sqrt(const float *a, float *o) { o[0] = sqrt1(a[0]); }
__global__ void p_mix(const float *a, float *o) { V3 x = LD(0), y = LD(1); ST(mk(mix1(x.x, y.x, a[6]), mix1(x.y, y.y, a[6]), mix1(x.z, y.z, a[6]))); }
__global__ void p_min(const float *a, float *o) { ST(vmin(LD(0), LD(1))); }
__global__ void p_sign(const float *a, float *o) { o[0] = sign1(a[0]); }
__global__ void p_pow(const float *a, float *o) { o[0] = pow5(a[0]); }
__global__ void p_hash(const float *a, float *o) { o[0] = (float)(uint32_t)fabs((double)dot(LD(0), mk(123.9898f, 348.233f, 433.3314f)) * 438.5453); }
}
'''
NAMES = "dot cross normalize div rcp sqrt mix min sign pow hash".split()
FLOAT_OP = re.compile(r"v_(pk_)?(mul|add|sub|subrev|fma|fmac|fmaak|fmamk|mac|mad|rcp|rsq|sqrt|exp|log|ldexp|frexp_mant|frexp_exp_i32|"
                      r"rndne|trunc|floor|min|max|cvt_f32_i32|cvt_i32_f32|cvt_f64_f32|cvt_f32_u32|cvt_u32_f64)_(f32|f64|i32_f32|i32_f64)?")


def float_ops(path, sym):
    dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", "--disassemble-symbols=" + sym, path], capture_output=True, text=True).stdout
    c = collections.Counter()
    for m in re.finditer(r"^\s+(v_\w+)", dis, re.M):
        op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", m.group(1))
        if not FLOAT_OP.match(op):
            continue
        n = 2 if op.startswith("v_pk_") else 1
        if n == 2:
            c["(packed)"] += 1
        op = op.replace("v_pk_", "v_").replace("v_fmac_", "v_fma_").replace("v_fmaak_", "v_fma_").replace("v_fmamk_", "v_fma_")
        op = op.replace("v_subrev_", "v_sub_")
        c[op] += n
    return c


def main():
    import __graft_entry__ as g
    rc = 0
    with tempfile.TemporaryDirectory() as d:
        cl, hip = os.path.join(d, "p.cl"), os.path.join(d, "p.hip")
        open(cl, "w").write(CL)
        open(hip, "w").write(HIP % os.path.join(g.CSRC, "pt_arith.hpp"))
        subprocess.check_call([CLANG, "-x", "cl", "-cl-std=CL1.2", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-Xclang",
                               "-finclude-default-header", "-O3", "-ffp-contract=off", cl, "-o", os.path.join(d, "cl.hsaco")])
        flags = [f for f in g.HIP_FLAGS if f not in ("-fPIC",)] + g.POLICY_FLAGS[1]
        subprocess.check_call([g.HIPCC] + flags + ["--cuda-device-only", "-S", hip, "-o", os.path.join(d, "hip.s")], stderr=subprocess.DEVNULL)
        subprocess.check_call([CLANG, "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", os.path.join(d, "hip.s"), "-o", os.path.join(d, "hip.o")])
        print("| builtin | float operations, ROCm OpenCL build | HIP policy 1 | |\n|---|---|---|---|")
        for n in NAMES:
            a, b = float_ops(os.path.join(d, "cl.hsaco"), "p_" + n), float_ops(os.path.join(d, "hip.o"), "p_" + n)
            # A packed add / mul computes two results even where one is needed and spells a - b as a + (-b): with packed
            # instructions in a stream, add + sub + mul are compared as one class, up to the packed instructions' spare halves
            pk = a.pop("(packed)", 0) + b.pop("(packed)", 0)
            if pk:
                ta = sum(a.pop(k, 0) for k in ("v_add_f32", "v_sub_f32", "v_mul_f32"))
                tb = sum(b.pop(k, 0) for k in ("v_add_f32", "v_sub_f32", "v_mul_f32"))
                same = a == b and abs(ta - tb) <= pk
                a["add+sub+mul"], b["add+sub+mul"] = ta, tb
            else:
                same = a == b
            rc |= 0 if same else 1
            fmt = lambda c: ", ".join("%s×%d" % kv for kv in sorted(c.items()))
            print("| %s | %s | %s | %s |" % (n, fmt(a), "same" if same else fmt(b), "ok" if same else "DIFFERENT"))
    return rc


if __name__ == "__main__":
    sys.exit(main())
