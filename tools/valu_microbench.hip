// valu_microbench.hip — settles the gfx950 VALU issue model the path tracer is priced against
// (VERDICT r01 "Next" #1a): cycles per wave-instruction per SIMD for each instruction class the
// trace kernels are made of, at 1 / 2 / 4 / 6 / 8 resident waves per SIMD, plus what a partial EXEC
// mask costs (divergence) and the dependent-issue latency.
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_microbench.hip -o gpurun_out/valu_microbench
//   gpurun_out/valu_microbench > gpurun_out/valu_microbench.json
//
// Method: every wave runs LOOPS iterations of a body of 64 independent instructions of one class
// (8 destination registers round-robin, inline asm so the compiler cannot fuse or drop them); a
// workgroup is 256 threads = one wave per SIMD; W workgroups per CU are made resident together by
// launching exactly CUs × W of them.  Per wave: Δs_memtime around the loop (shader clock).  Reported:
//   cyc_per_inst_simd = median over waves of Δ / (LOOPS × 64) / W     (SIMD's cost of one wave-instruction)
// and the same figure from the wall clock (hipEvent) × the measured shader clock.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)

struct Out {
    unsigned long long cycles;   // Δ s_memtime of the loop
    unsigned long long real;     // Δ s_memrealtime (100 MHz)
};

// lane masks: 0 all 64 · 1 lanes 0..31 · 2 lanes 32..63 · 3 lanes 0..15 · 4 even lanes · 5 lane 0 only · 6 lanes 0..47
__device__ __forceinline__ bool lane_on(int mask) {
    unsigned l = threadIdx.x & 63u;
    switch (mask) {
        case 0: return true;
        case 1: return l < 32u;
        case 2: return l >= 32u;
        case 3: return l < 16u;
        case 4: return (l & 1u) == 0u;
        case 5: return l == 0u;
        default: return l < 48u;
    }
}

#define KERNEL_PROLOGUE                                                              \
    float a0 = seed[threadIdx.x & 63], a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, \
          a6 = a0 + 6.0f, a7 = a0 + 7.0f;                                            \
    float b = seed[64 + (threadIdx.x & 63)], c = seed[128 + (threadIdx.x & 63)];     \
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, db = b; \
    unsigned u0 = __float_as_uint(a0), u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7; \
    unsigned ub = __float_as_uint(b) | 1u;                                          \
    (void)d0; (void)d1; (void)d2; (void)d3; (void)d4; (void)d5; (void)d6; (void)d7; (void)db; \
    (void)u0; (void)u1; (void)u2; (void)u3; (void)u4; (void)u5; (void)u6; (void)u7; (void)ub; (void)c; \
    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;                               \
    __syncthreads();                                                                 \
    if (lane_on(mask)) {                                                             \
        t0 = __builtin_amdgcn_s_memtime();                                           \
        r0 = __builtin_amdgcn_s_memrealtime();                                       \
        for (int it = 0; it < loops; it++) {

#define KERNEL_EPILOGUE                                                              \
        }                                                                            \
        t1 = __builtin_amdgcn_s_memtime();                                           \
        r1 = __builtin_amdgcn_s_memrealtime();                                       \
    }                                                                                \
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + \
              __uint_as_float(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);               \
    if (s == 123.456f) sink[0] = s;                                                  \
    unsigned first = mask == 2 ? 32u : 0u;                                           \
    if ((threadIdx.x & 63u) == first) {                                              \
        unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);           \
        out[w].cycles = t1 - t0;                                                     \
        out[w].real = r1 - r0;                                                       \
    }

#define DEF_KERNEL(NAME, INST)                                                                       \
    __global__ __launch_bounds__(256) void k_##NAME(const float *seed, Out *out, float *sink, int loops, int mask) { \
        KERNEL_PROLOGUE                                                                              \
        BODY64(INST)                                                                                 \
        KERNEL_EPILOGUE                                                                              \
    }

// ---- instruction classes (destination a<i>/d<i>/u<i> round-robin → independent)
#define I_ADD_F32(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a##i) : "v"(b));
#define I_MUL_F32(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a##i) : "v"(b));
#define I_FMA_F32(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a##i) : "v"(b), "v"(c));
#define I_MOV_B32(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a##i) : "v"(b));
#define I_MAX_F32(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a##i) : "v"(b));
#define I_RCP_F32(i) asm volatile("v_rcp_f32 %0, %1" : "=v"(a##i) : "v"(b));
#define I_RSQ_F32(i) asm volatile("v_rsq_f32 %0, %1" : "=v"(a##i) : "v"(b));
#define I_SQRT_F32(i) asm volatile("v_sqrt_f32 %0, %1" : "=v"(a##i) : "v"(b));
#define I_ADD_F64(i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d##i) : "v"(db));
#define I_MUL_F64(i) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d##i) : "v"(db));
#define I_FMA_F64(i) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d##i) : "v"(db));
#define I_CVT_F64_F32(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d##i) : "v"(b));
#define I_CVT_U32_F64(i) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u##i) : "v"(db));
#define I_CVT_F32_U32(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a##i) : "v"(ub));
#define I_MUL_LO_U32(i) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(u##i) : "v"(ub));
#define I_MUL_HI_U32(i) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(u##i) : "v"(ub));
#define I_ADD_U32(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u##i) : "v"(ub));
#define I_AND_B32(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u##i) : "v"(ub));
#define I_LSHL_B32(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u##i));
#define I_CMP_F32(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a##i), "v"(b) : "vcc");
#define I_CMP_CNDMASK(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a##i) : "v"(b), "v"(c) : "vcc");
#define I_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b) : "vcc");
#define I_DIV_SCALE(i) asm volatile("v_div_scale_f32 %0, vcc, %1, %2, %1" : "=v"(a##i) : "v"(b), "v"(c) : "vcc");
#define I_DIV_FMAS(i) asm volatile("v_div_fmas_f32 %0, %1, %2, %0" : "+v"(a##i) : "v"(b), "v"(c) : "vcc");
#define I_DIV_FIXUP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define I_PK_MUL_F32(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(d##i) : "v"(db));
#define I_PK_ADD_F32(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(d##i) : "v"(db));
#define I_PK_FMA_F32(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(d##i) : "v"(db));
#define I_READFIRSTLANE(i) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(a##i) : "s20");
#define I_S_NOP(i) asm volatile("s_nop 0");
#define I_V_NOP(i) asm volatile("v_nop");
#define I_S_ADD(i) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");
#define I_CNDMASK_VCC(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b));   // vcc set before the loop, not clobbered
#define I_CMP_E64(i) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a##i), "v"(b) : "s20", "s21");
#define I_READLANE(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a##i) : "s20");
#define I_WRITELANE(i) asm volatile("v_writelane_b32 %0, s20, 3" : "+v"(a##i) : : "s20");
#define I_LSHL_ADD_U64(i) asm volatile("v_lshl_add_u64 %0, %1, 2, %0" : "+v"(d##i) : "v"(db));
#define I_MAD_U64_U32(i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %1, %0" : "+v"(d##i) : "v"(ub) : "s20", "s21");
#define I_LSHL_ADD_U32(i) asm volatile("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(u##i) : "v"(ub));
#define I_ADD3_U32(i) asm volatile("v_add3_u32 %0, %1, %1, %0" : "+v"(u##i) : "v"(ub));
#define I_BFE_U32(i) asm volatile("v_bfe_u32 %0, %1, 3, 5" : "=v"(u##i) : "v"(ub));
#define I_MIN_F32(i) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a##i) : "v"(b));
#define I_MIN_U32(i) asm volatile("v_min_u32 %0, %1, %0" : "+v"(u##i) : "v"(ub));
#define I_MAX3_U32(i) asm volatile("v_max3_u32 %0, %1, %1, %0" : "+v"(u##i) : "v"(ub));
#define I_MUL_U32_U24(i) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(u##i) : "v"(ub));
#define I_SUB_F32(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a##i) : "v"(b));
#define I_XOR_B32(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u##i) : "v"(ub));
#define I_MOV_DISTINCT(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a##i) : "v"(u##i));
#define I_FMAC_F32(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define I_CMP_CLASS(i) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(a##i), "v"(ub) : "vcc");
#define I_FLOOR_F32(i) asm volatile("v_floor_f32 %0, %1" : "=v"(a##i) : "v"(b));
// dependent chain: every instruction reads the previous one's result
#define I_DEP_ADD_F32(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a0) : "v"(b));
#define I_DEP_MUL_F64(i) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d0) : "v"(db));
#define I_DEP_RCP_F32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a0));
// the IEEE divide and square root as hipcc expands them (whole expressions, compiler-scheduled)
#define I_IEEE_DIV(i) a##i = b / a##i;
#define I_IEEE_SQRT(i) a##i = sqrtf(a##i);
// one VALU + one SALU alternating: does the scalar unit issue beside the vector stream?
#define I_VALU_SALU(i) asm volatile("v_add_f32 %0, %1, %0\n\ts_add_u32 s20, s20, 1" : "+v"(a##i) : "v"(b) : "s20", "scc");
// LDS: broadcast b128 read (all lanes one address) and per-lane b32 read
#define I_DS_B128_BCAST(i) asm volatile("ds_read_b128 %0, %1" : "=v"(q##i) : "v"(lds_b) : "memory");
#define I_DS_B32(i) asm volatile("ds_read_b32 %0, %1" : "=v"(a##i) : "v"(lds_l) : "memory");

DEF_KERNEL(add_f32, I_ADD_F32)
DEF_KERNEL(mul_f32, I_MUL_F32)
DEF_KERNEL(fma_f32, I_FMA_F32)
DEF_KERNEL(mov_b32, I_MOV_B32)
DEF_KERNEL(max_f32, I_MAX_F32)
DEF_KERNEL(rcp_f32, I_RCP_F32)
DEF_KERNEL(rsq_f32, I_RSQ_F32)
DEF_KERNEL(sqrt_f32, I_SQRT_F32)
DEF_KERNEL(add_f64, I_ADD_F64)
DEF_KERNEL(mul_f64, I_MUL_F64)
DEF_KERNEL(fma_f64, I_FMA_F64)
DEF_KERNEL(cvt_f64_f32, I_CVT_F64_F32)
DEF_KERNEL(cvt_u32_f64, I_CVT_U32_F64)
DEF_KERNEL(cvt_f32_u32, I_CVT_F32_U32)
DEF_KERNEL(mul_lo_u32, I_MUL_LO_U32)
DEF_KERNEL(mul_hi_u32, I_MUL_HI_U32)
DEF_KERNEL(add_u32, I_ADD_U32)
DEF_KERNEL(and_b32, I_AND_B32)
DEF_KERNEL(lshl_b32, I_LSHL_B32)
DEF_KERNEL(cmp_f32, I_CMP_F32)
DEF_KERNEL(cmp_cndmask_pair, I_CMP_CNDMASK)
DEF_KERNEL(cndmask, I_CNDMASK)
DEF_KERNEL(div_scale, I_DIV_SCALE)
DEF_KERNEL(div_fmas, I_DIV_FMAS)
DEF_KERNEL(div_fixup, I_DIV_FIXUP)
DEF_KERNEL(pk_mul_f32, I_PK_MUL_F32)
DEF_KERNEL(pk_add_f32, I_PK_ADD_F32)
DEF_KERNEL(pk_fma_f32, I_PK_FMA_F32)
DEF_KERNEL(readfirstlane, I_READFIRSTLANE)
DEF_KERNEL(s_nop, I_S_NOP)
DEF_KERNEL(s_add, I_S_ADD)
DEF_KERNEL(dep_add_f32, I_DEP_ADD_F32)
DEF_KERNEL(dep_mul_f64, I_DEP_MUL_F64)
DEF_KERNEL(dep_rcp_f32, I_DEP_RCP_F32)
DEF_KERNEL(ieee_div_expr, I_IEEE_DIV)
DEF_KERNEL(ieee_sqrt_expr, I_IEEE_SQRT)
DEF_KERNEL(valu_salu_pair, I_VALU_SALU)
DEF_KERNEL(cndmask_vcc, I_CNDMASK_VCC)
DEF_KERNEL(cmp_e64, I_CMP_E64)
DEF_KERNEL(readlane, I_READLANE)
DEF_KERNEL(writelane, I_WRITELANE)
DEF_KERNEL(lshl_add_u64, I_LSHL_ADD_U64)
DEF_KERNEL(mad_u64_u32, I_MAD_U64_U32)
DEF_KERNEL(lshl_add_u32, I_LSHL_ADD_U32)
DEF_KERNEL(add3_u32, I_ADD3_U32)
DEF_KERNEL(bfe_u32, I_BFE_U32)
DEF_KERNEL(min_f32, I_MIN_F32)
DEF_KERNEL(min_u32, I_MIN_U32)
DEF_KERNEL(max3_u32, I_MAX3_U32)
DEF_KERNEL(mul_u32_u24, I_MUL_U32_U24)
DEF_KERNEL(sub_f32, I_SUB_F32)
DEF_KERNEL(xor_b32, I_XOR_B32)
DEF_KERNEL(mov_distinct, I_MOV_DISTINCT)
DEF_KERNEL(fmac_f32, I_FMAC_F32)
DEF_KERNEL(cmp_class, I_CMP_CLASS)
DEF_KERNEL(floor_f32, I_FLOOR_F32)
DEF_KERNEL(v_nop, I_V_NOP)

typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_ds_b128_bcast(const float *seed, Out *out, float *sink, int loops, int mask) {
    __shared__ float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = seed[i & 127];
    f4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0, q4 = q0, q5 = q0, q6 = q0, q7 = q0;
    unsigned lds_b = (unsigned)(size_t)lds + 16u * (unsigned)(seed[0] > 1e30f);
    KERNEL_PROLOGUE
    BODY64(I_DS_B128_BCAST)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KERNEL_EPILOGUE
    f4 qs = q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
    if (qs.x + qs.y + qs.z + qs.w == 123.456f) sink[1] = qs.x;
}
__global__ __launch_bounds__(256) void k_ds_b32(const float *seed, Out *out, float *sink, int loops, int mask) {
    __shared__ float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = seed[i & 127];
    unsigned lds_l = (unsigned)(size_t)lds + 4u * (threadIdx.x & 63u);
    KERNEL_PROLOGUE
    BODY64(I_DS_B32)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KERNEL_EPILOGUE
}

typedef void (*kern_t)(const float *, Out *, float *, int, int);
struct Case {
    const char *name;
    kern_t fn;
    int per_body;  // machine instructions per macro expansion
};

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    float *d_seed, *d_sink;
    std::vector<float> seed(1024);
    for (int i = 0; i < 1024; i++) seed[i] = 1.0f + 0.37f * (float)((i * 2654435761u) % 1000u) / 1000.0f;
    CHECK(hipMalloc(&d_seed, 1024 * sizeof(float)));
    CHECK(hipMalloc(&d_sink, 16 * sizeof(float)));
    CHECK(hipMemcpy(d_seed, seed.data(), 1024 * sizeof(float), hipMemcpyHostToDevice));
    const int max_waves = cus * 4 * 8;
    Out *d_out;
    CHECK(hipMalloc(&d_out, max_waves * sizeof(Out)));
    std::vector<Out> h_out(max_waves);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    const Case cases[] = {
        {"v_add_f32", k_add_f32, 1}, {"v_mul_f32", k_mul_f32, 1}, {"v_fma_f32", k_fma_f32, 1}, {"v_mov_b32", k_mov_b32, 1},
        {"v_max_f32", k_max_f32, 1}, {"v_rcp_f32", k_rcp_f32, 1}, {"v_rsq_f32", k_rsq_f32, 1}, {"v_sqrt_f32", k_sqrt_f32, 1},
        {"v_add_f64", k_add_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_fma_f64", k_fma_f64, 1},
        {"v_cvt_f64_f32", k_cvt_f64_f32, 1}, {"v_cvt_u32_f64", k_cvt_u32_f64, 1}, {"v_cvt_f32_u32", k_cvt_f32_u32, 1},
        {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1}, {"v_add_u32", k_add_u32, 1},
        {"v_and_b32", k_and_b32, 1}, {"v_lshlrev_b32", k_lshl_b32, 1}, {"v_cmp_lt_f32", k_cmp_f32, 1},
        {"v_cmp+v_cndmask (pair)", k_cmp_cndmask_pair, 2},
        {"v_div_scale_f32", k_div_scale, 1}, {"v_div_fmas_f32", k_div_fmas, 1}, {"v_div_fixup_f32", k_div_fixup, 1},
        {"v_pk_mul_f32", k_pk_mul_f32, 1}, {"v_pk_add_f32", k_pk_add_f32, 1}, {"v_pk_fma_f32", k_pk_fma_f32, 1},
        {"v_readfirstlane_b32", k_readfirstlane, 1}, {"s_nop 0", k_s_nop, 1}, {"s_add_u32", k_s_add, 1},
        {"dependent v_add_f32", k_dep_add_f32, 1}, {"dependent v_mul_f64", k_dep_mul_f64, 1},
        {"dependent v_rcp_f32", k_dep_rcp_f32, 1},
        {"IEEE a/b (hipcc expansion, per expression)", k_ieee_div_expr, 1},
        {"IEEE sqrtf (hipcc expansion, per expression)", k_ieee_sqrt_expr, 1},
        {"v_add_f32 + s_add_u32 (pair)", k_valu_salu_pair, 2},
        {"ds_read_b128 broadcast", k_ds_b128_bcast, 1}, {"ds_read_b32 per lane", k_ds_b32, 1},
        {"v_cndmask_b32 (vcc)", k_cndmask_vcc, 1}, {"v_cmp_lt_f32 e64 (sgpr pair)", k_cmp_e64, 1},
        {"v_readlane_b32", k_readlane, 1}, {"v_writelane_b32", k_writelane, 1}, {"v_lshl_add_u64", k_lshl_add_u64, 1},
        {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_lshl_add_u32", k_lshl_add_u32, 1}, {"v_add3_u32", k_add3_u32, 1},
        {"v_bfe_u32", k_bfe_u32, 1}, {"v_min_f32", k_min_f32, 1}, {"v_min_u32", k_min_u32, 1},
        {"v_max3_u32", k_max3_u32, 1}, {"v_mul_u32_u24", k_mul_u32_u24, 1}, {"v_sub_f32", k_sub_f32, 1},
        {"v_xor_b32", k_xor_b32, 1}, {"v_mov_b32 (distinct sources)", k_mov_distinct, 1}, {"v_fmac_f32", k_fmac_f32, 1},
        {"v_cmp_class_f32", k_cmp_class, 1}, {"v_floor_f32", k_floor_f32, 1}, {"v_nop", k_v_nop, 1},
    };
    const int waves_per_simd[] = {1, 2, 4, 6, 8};
    const int loops = 400;

    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"loops\": %d, \"body\": 64,\n \"results\": [\n", prop.name,
           prop.gcnArchName, cus, loops);
    bool first_line = true;
    auto run = [&](const Case &cs, int W, int mask, const char *mask_name) {
        const int blocks = cus * W, waves = blocks * 4;
        for (int rep = 0; rep < 2; rep++) {  // rep 0 warms up
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(cs.fn, dim3(blocks), dim3(256), 0, 0, d_seed, d_out, d_sink, loops, mask);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
        }
        CHECK(hipGetLastError());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h_out.data(), d_out, waves * sizeof(Out), hipMemcpyDeviceToHost));
        std::vector<double> cyc(waves), clk(waves);
        for (int i = 0; i < waves; i++) {
            cyc[i] = (double)h_out[i].cycles;
            clk[i] = h_out[i].real ? (double)h_out[i].cycles / (double)h_out[i].real * 100.0 : 0.0;  // MHz
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(clk.begin(), clk.end());
        double med = cyc[waves / 2], n_inst = (double)loops * 64.0 * cs.per_body;
        double mhz = clk[waves / 2];
        printf("%s  {\"inst\": \"%s\", \"waves_per_simd\": %d, \"lanes\": \"%s\", \"cyc_per_inst_wave\": %.3f, "
               "\"cyc_per_inst_simd\": %.3f, \"wall_cyc_per_inst_simd\": %.3f, \"clock_mhz\": %.0f, \"ms\": %.4f}",
               first_line ? "" : ",\n", cs.name, W, mask_name, med / n_inst, med / n_inst / W,
               (double)ms * 1e-3 * mhz * 1e6 / (n_inst * W), mhz, ms);
        first_line = false;
    };
    for (const Case &cs : cases)
        for (int W : waves_per_simd) run(cs, W, 0, "all 64");
    // what a partial EXEC mask costs (SIMD-32 executes a wave64 instruction in two passes of 32 lanes)
    const char *mask_names[] = {"all 64", "0..31", "32..63", "0..15", "even lanes", "lane 0", "0..47"};
    const Case div_cases[] = {cases[0], cases[5], cases[9], cases[14]};
    for (const Case &cs : div_cases)
        for (int mask = 1; mask <= 6; mask++)
            for (int W : {1, 4}) run(cs, W, mask, mask_names[mask]);
    printf("\n ]}\n");
    return 0;
}
