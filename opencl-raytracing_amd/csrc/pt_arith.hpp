// pt_arith.hpp — the ARITHMETIC POLICY of the device code (one per translation unit, -DPT_ARITH=k).
//
// The reference's "random numbers" are table entries indexed by a hash of the ray direction
// (kernels/raytracer.cl:113-125), so a 1-ulp difference in any float operation re-routes a path: parity per pixel
// needs every pixel-sample bit-identical, and "the reference's result" exists only relative to ONE definition of the
// OpenCL builtins and of `/`, sqrt and the contraction of a*b+c.  Three definitions are built, each into a
// namespace of its own (PT_NS), selected at run time with rt_set_option(RT_OPT_ARITH, k):
//
//   PT_ARITH 0  RT_ARITH_IEEE                IEEE-754 binary32, every operation rounded once, no fused multiply-add,
//               correctly rounded `/` and sqrt, one plain formula per builtin (oracle/ref_shim.cpp).  What the CPU
//               chain (the CPU restatement under oracle/, its golden vectors under tests/golden) is pinned to: a host can restate it exactly.
//   PT_ARITH 1  RT_ARITH_ROCM_OCL_NOCONTRACT the builtins AS ROCm's OpenCL LIBRARY DEFINES THEM for gfx950 (opencl.bc /
//               ocml.bc of the same ROCm release, looked up with llvm-dis — tools/arith_probe.py keeps the comparison):
//                   dot(a,b)       = fma(a.z, b.z, fma(a.y, b.y, a.x*b.x))                  _Z3dotDv3_fS_
//                   cross(a,b).x   = fma(a.y, b.z, b.y*(-a.z)), .y, .z likewise            _Z5crossDv3_fS_
//                   normalize(v)   = v * rsqrt(dot(v,v)) with the library's range scaling  _Z9normalizeDv3_f → __ocml_rsqrt_f32 (v_rsq_f32)
//                   mix(a,b,t)     = fma(b - a, t, a)                                      _Z3mixDv3_fS_f
//                   min            = llvm.minnum (__ocml_min_f32),  sign = copysign(x is NaN or 0 ? 0 : 1, x)
//                   pow(x, 5)      = __ocml_pow_f32(x, 5.0f)  — the very function, the same bitcode file
//                   a / b          = the 2.5-ulp expansion OpenCL allows (frexp, v_rcp_f32, ldexp): `fdiv !fpmath 2.5`,
//                   sqrt(x)        = the 3-ulp one (range-scaled v_sqrt_f32): `llvm.sqrt !fpmath 3.0`
//               (the last two come from compiling this translation unit with -fno-hip-fp32-correctly-rounded-divide-sqrt:
//               clang then attaches the same !fpmath metadata as under -x cl, and the same backend expands them),
//               the kernel's OWN expressions not contracted (-ffp-contract=off).  Bit for bit what
//               oracle/_ref_gfx950/ref950_nocontract.hsaco computes — that code object is the checker
//               (tests/test_gpu_ref950.py), no host can restate v_rsq_f32 / v_rcp_f32 / v_sqrt_f32.
//   PT_ARITH 2  RT_ARITH_ROCM_OCL            the same, plus the contractions clang performs on the reference's own
//               expressions under OpenCL's default -ffp-contract=on: an a*b that is a direct operand of + or − in ONE
//               source expression becomes llvm.fmuladd (→ v_fma_f32 on gfx950).  The sites, listed from the IR of the
//               reference file (clang -x cl -O3 -Xclang -disable-llvm-passes -emit-llvm | grep fmuladd):
//                   genInitRay :137  fma(t, ver, fma(s, hor, llc))        rayPointAtParam :142  fma(dir, t, origin)
//                   hitSphere :152-153, hitLens :199-205   c = fma(-r, r, dot(oc,oc));  dis = fma(b, b, -c)
//                   getTextureUV :102   fma(C, v, fma(A, 1-u-v, B*u))     rayReflect :364   fma(-(2·dot), n, dir)
//                   rayRefract :381,:385 / rayRefractDielectric :424,:428   1 - k²(1 - cai²) = fma(-(k·k), fma(-cai, cai, 1), 1);
//                                        k·dir − n·(k·cai + sqrt) = fma(k, dir, −(n · fma(k, cai, sqrt)))
//                   schlick :404        fma(1 − r0, pow, r0)
//               Bit for bit what oracle/_ref_gfx950/ref950.hsaco — the reference as ROCm's OpenCL builds it by default
//               — computes.
//
// The acceleration structures' CULLING arithmetic (pt_device.hpp hit_spheres_bvh, pt_mesh_bvh.hpp) is ours in every
// policy; their margins were derived for policy 0 and hold a fortiori for 1 and 2: a fused multiply-add rounds once
// where policy 0 rounds twice, so every forward error bound of the discriminant / Möller–Trumbore numerators stands;
// the 1-ulp reciprocal in hitTriangle's f = 1/a (0.5 ulp in policy 0) adds 1.5·2^-24 to the barycentrics' relative
// slack, which the slab tests' additive 1e-5·(dfar + o_max) + 1e-6 covers 100-fold; rsq-normalised directions are a
// few ulp further from unit length, which the sphere walk's per-ray margin measures (|dd − 1|) rather than assumes.
#pragma once
#include "pt_types.hpp"

#ifndef PT_ARITH
#define PT_ARITH 0
#endif
#if PT_ARITH == 0
#define PT_NS pt_a0
#elif PT_ARITH == 1
#define PT_NS pt_a1
#elif PT_ARITH == 2
#define PT_NS pt_a2
#else
#error "PT_ARITH must be 0 (IEEE), 1 (ROCm OpenCL builtins, no contraction) or 2 (ROCm OpenCL default build)"
#endif
#define PT_OCL (PT_ARITH != 0)        // ROCm's OpenCL builtin library
#define PT_CONTRACT (PT_ARITH == 2)   // clang's -ffp-contract=on at the reference's expression sites

#if PT_OCL
// the library's own functions: the HIP tool chain links the very same ocml.bc as the OpenCL one
extern "C" __device__ float __ocml_rsqrt_f32(float);
extern "C" __device__ float __ocml_pow_f32(float, float);
extern "C" __device__ float __ocml_min_f32(float, float);
#endif

namespace PT_NS {
using namespace pt;

struct V3 {
    float x, y, z;
};

PT_DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
PT_DEV V3 ld3(const rt_float3 &f) { return V3{f.x, f.y, f.z}; }
PT_DEV V3 xyz(float4 f) { return V3{f.x, f.y, f.z}; }
PT_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV V3 operator*(V3 a, float k) { return V3{a.x * k, a.y * k, a.z * k}; }
PT_DEV V3 operator/(V3 a, float k) { return V3{a.x / k, a.y / k, a.z / k}; }   // policy 1, 2: three 2.5-ulp divisions (TU flag)
PT_DEV V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }

// ---- the reference's own expression sites (contracted only under policy 2) ------------------------------------
// a*b + c
PT_DEV float mad(float a, float b, float c) { return PT_CONTRACT ? __builtin_fmaf(a, b, c) : a * b + c; }
// c − a*b   (clang: fmuladd(−a, b, c))
PT_DEV float nmad(float a, float b, float c) { return PT_CONTRACT ? __builtin_fmaf(-a, b, c) : c - a * b; }
// a*b − c   (clang: fmuladd(a, b, −c))
PT_DEV float msub(float a, float b, float c) { return PT_CONTRACT ? __builtin_fmaf(a, b, -c) : a * b - c; }
PT_DEV V3 mad(V3 a, float k, V3 c) { return V3{mad(a.x, k, c.x), mad(a.y, k, c.y), mad(a.z, k, c.z)}; }      // a*k + c
PT_DEV V3 nmad(V3 a, float k, V3 c) { return V3{nmad(a.x, k, c.x), nmad(a.y, k, c.y), nmad(a.z, k, c.z)}; }  // c − a*k

// ---- builtins ---------------------------------------------------------------------------------------------------
#if !PT_OCL
// dot(a,b) = (ax*bx + ay*by) + az*bz — the builtin definition shared with the oracle
PT_DEV float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PT_DEV V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PT_DEV float sqrt1(float x) { return sqrtf(x); }   // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
PT_DEV V3 vmin(V3 a, V3 b) { return V3{b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z}; }
PT_DEV float sign1(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : (x == 0.0f ? x : 0.0f)); }
PT_DEV float pow5(float x) {
    float x2 = x * x;
    return (x2 * x2) * x;
}
PT_DEV float mix1(float a, float b, float t) { return a + (b - a) * t; }
#else
PT_DEV float dot(V3 a, V3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
PT_DEV V3 cross(V3 a, V3 b) {
    return V3{__builtin_fmaf(a.y, b.z, b.y * -a.z), __builtin_fmaf(a.z, b.x, b.z * -a.x), __builtin_fmaf(a.x, b.y, b.x * -a.y)};
}
PT_DEV float sqrt1(float x) { return __builtin_sqrtf(x); }   // llvm.sqrt !fpmath 3.0 under this TU's flags: range-scaled v_sqrt_f32
PT_DEV V3 vmin(V3 a, V3 b) { return V3{__ocml_min_f32(a.x, b.x), __ocml_min_f32(a.y, b.y), __ocml_min_f32(a.z, b.z)}; }
PT_DEV float sign1(float x) { return __builtin_copysignf((__builtin_isnan(x) || x == 0.0f) ? 0.0f : 1.0f, x); }
PT_DEV float pow5(float x) { return __ocml_pow_f32(x, 5.0f); }
PT_DEV float mix1(float a, float b, float t) { return __builtin_fmaf(b - a, t, a); }
#endif

// Three IEEE divisions by ONE denominator (normalize :137,:365,:397; the sphere / lens normal :160,:248) — policy 0.
// hipcc expands a correctly rounded a / d into
//     ds = div_scale(d, d, a); as = div_scale(a, d, a); r = rcp(ds); e = fma(-ds, r, 1); r = fma(e, r, r);
//     q = as * r; t = fma(-ds, q, as); q = fma(t, r, q); t = fma(-ds, q, as); q = div_fmas(t, r, q); div_fixup(q, d, a)
// (11 instructions, 43 issue cycles per SIMD measured — profiles/r02_valu_microbench.md).  div_scale only rescales
// operands whose quotient or reciprocal would leave the normal range, div_fmas is a plain fma when nothing was
// scaled, and div_fixup only replaces the result for zero / infinite / NaN operands.  For operands safely inside
// the normal range (|a| in [2^-90, 2^60] — in particular a != 0 — and |d| in [2^-30, 2^30]: exponent difference
// in (-126, 96), numerator exponent field > 23, 1/d normal) the expansion is therefore EXACTLY the plain sequence
// below, and its reciprocal part (rcp and two fma) depends on the denominator only: computed once and shared by
// the three numerators — the same operations with the same operands, hence the same bits, in 18 instead of 33
// instructions.  If any active lane of the wave is outside that range the whole wave takes the compiler's
// divisions (same bits for the in-range lanes, so the choice of path never shows in the result).
// tests/test_gpu_units.py::test_div3_is_three_ieee_divisions compares the two paths on 2^24 operand sets.
#ifndef PT_DIV3
#define PT_DIV3 0  // A/B on MI355X: bit-identical, 31 % fewer v_rcp and 15 % fewer fma issued, kernel time unchanged (C2 2.424 vs 2.418 ms)
#endif
PT_DEV bool div3_in_range(V3 a, float d) {
    uint32_t ax = __float_as_uint(a.x) & 0x7FFFFFFFu, ay = __float_as_uint(a.y) & 0x7FFFFFFFu,
             az = __float_as_uint(a.z) & 0x7FFFFFFFu, ad = __float_as_uint(d) & 0x7FFFFFFFu;
    uint32_t lo = min(min(ax, ay), az), hi = max(max(ax, ay), az);
    // 2^-90 = 0x12800000, 2^60 = 0x5D800000, 2^-30 = 0x30800000, 2^30 = 0x4E800000 (NaN / inf patterns are above all of them)
    return lo >= 0x12800000u && hi <= 0x5D800000u && ad >= 0x30800000u && ad <= 0x4E800000u;
}
PT_DEV float div_shared(float a, float nd, float r) {
    float q = a * r;
    float t = __builtin_fmaf(nd, q, a);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(nd, q, a);
    return __builtin_fmaf(t, r, q);
}
PT_DEV V3 div3(V3 a, float d) {
    if (!PT_OCL && PT_DIV3 && __all(div3_in_range(a, d))) {
        float nd = -d;
        float r = __builtin_amdgcn_rcpf(d);
        float e = __builtin_fmaf(nd, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
        return V3{div_shared(a.x, nd, r), div_shared(a.y, nd, r), div_shared(a.z, nd, r)};
    }
    return a / d;
}

#if !PT_OCL
// normalize(v) = v / sqrt(dot(v,v)); sqrtf and '/' are correctly rounded in HIP
// (-fhip-fp32-correctly-rounded-divide-sqrt is the default and is passed explicitly)
PT_DEV V3 normalize(V3 a) { return div3(a, sqrtf(dot(a, a))); }
#else
// _Z9normalizeDv3_f of ROCm's opencl.bc, statement by statement: the zero vector is returned as it is; a squared
// length below FLT_MIN (or infinite) is recomputed from the vector scaled by 2^86 (2^-66; a vector with infinite
// components becomes its sign pattern); then every component is multiplied by __ocml_rsqrt_f32 of the squared length
PT_DEV V3 normalize(V3 v) {
    if (v.x == 0.0f && v.y == 0.0f && v.z == 0.0f) return v;
    float d = dot(v, v);
    V3 s = v;
    if (d < 0x1p-126f) {
        s = v * 0x1p86f;
        d = dot(s, s);
    } else if (d == INFINITY) {
        s = v * 0x1p-66f;
        d = dot(s, s);
        if (d == INFINITY) {
            s = V3{__builtin_copysignf(__builtin_isinf(s.x) ? 1.0f : 0.0f, s.x), __builtin_copysignf(__builtin_isinf(s.y) ? 1.0f : 0.0f, s.y),
                   __builtin_copysignf(__builtin_isinf(s.z) ? 1.0f : 0.0f, s.z)};
            d = dot(s, s);
        }
    }
    return s * __ocml_rsqrt_f32(d);
}
#endif

// :127 — false for NaN
PT_DEV bool in_range(float x) { return (x - RT_MAX_DISTANCE) * (x - RT_MIN_DISTANCE) <= 0.0f; }

struct Ray {
    V3 o, d;
};
PT_DEV V3 point_at(const Ray &r, float t) { return mad(r.d, t, r.o); }  // :141  origin + dir * t

}  // namespace PT_NS
