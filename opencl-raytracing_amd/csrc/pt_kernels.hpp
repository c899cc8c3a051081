// pt_kernels.hpp — what one arithmetic policy's translation unit (pt_kernels.hip, -DPT_ARITH=k) exports to the host
// side of librt_amd.so: the launchers of its kernels.  All launches go to ctx->stream; every function returns an RT_*
// code and records its message on the context.
#pragma once
#include "rt_context.hpp"

namespace pt {

struct KernelSet {
    int arith;             // RT_ARITH_* this set was compiled for
    const char *name;
    // direct path: MODE_ACCUM / MODE_TRACE / MODE_RETRACE (rt_render, rt_render_again, prefix sharing off)
    int (*launch_render)(rt_context *ctx, int mode, const float cam[12], uint32_t first, uint32_t count, uint32_t glog2);
    // fused path: pt_prefix + pt_samples_q / pt_samples_w / pt_samples
    int (*launch_fused)(rt_context *ctx, const float cam[12], uint32_t first, uint32_t count, uint32_t glog2);
    // rt_trace_samples: d_in = n x, n y, n sample (uint32), d_out = 3 n floats
    int (*launch_probe)(rt_context *ctx, const FrameParams &fp, const DeviceScene &sc, const uint32_t *d_in, uint32_t n, float *d_out);
    // unit probes (rt_debug_hit / rt_debug_material / rt_debug_div3)
    int (*launch_debug_hit)(rt_context *ctx, const DeviceScene &sc, int kind, const float *d_rays, const uint32_t *d_prim,
                            const uint32_t *d_face, uint32_t n, float *d_out);
    int (*launch_debug_material)(rt_context *ctx, const DeviceScene &sc, int routine, const float *d_in, uint32_t n, float *d_out);
    int (*launch_debug_div3)(rt_context *ctx, const float *d_in, uint32_t n, float *d_out);
    // per-face unit normals normalize(cross(e1, e2)) (raytracer.cl:285) with THIS policy's builtins, written into the
    // (A, e1, e2, n) records: n_records records of 3 float4 each
    int (*launch_face_normals)(rt_context *ctx, float4 *d_records, uint32_t n_records);
    // one builtin per record, for tests against oracle/_ref_gfx950's probe kernels: op 0 dot, 1 cross, 2 normalize,
    // 3 a/b, 4 sqrt, 5 mix, 6 min, 7 sign, 8 pow(x,5), 9 the table hash; in: n × 8 floats, out: n × 4 floats
    int (*launch_debug_builtin)(rt_context *ctx, int op, const float *d_in, uint32_t n, float *d_out);
};

// defined by pt_kernels.hip compiled with -DPT_ARITH=0 / 1 / 2
const KernelSet *kernel_set_a0();
const KernelSet *kernel_set_a1();
const KernelSet *kernel_set_a2();

}  // namespace pt
