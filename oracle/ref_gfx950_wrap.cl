// ref_gfx950_wrap.cl — TEST INFRASTRUCTURE (build container + GPU box), written for this repository.
//
// The reference's kernel file, compiled for gfx950 by ROCm's own OpenCL tool chain with ROCm's REAL
// OpenCL builtin library (opencl.bc / ocml.bc / ockl.bc — no stand-ins), is what the reference would
// execute on an MI355X under ROCm's OpenCL runtime.  Its kernels `trace` / `retrace` take OpenCL image
// objects, which a HIP process cannot create; this wrapper therefore includes the reference file WHERE
// IT LIES (-I /root/reference, nothing is copied) and adds one kernel that runs the reference's own
// genInitRay + getCol for pixel (get_global_id(0), get_global_id(1)) and stores the LINEAR radiance of
// one sample in a plain buffer instead of an image.  The image2d_array_t argument is only dereferenced
// by t_textured materials (raytracer.cl:105-107, :474-477): scenes without them pass a null handle.
// The reference's createScene kernel (:541-558) is used as it is.
//
// Build recipe: oracle/Makefile, target ref_gfx950 → oracle/_ref_gfx950/ref950.hsaco (git-ignored).
#include "kernels/raytracer.cl"

__kernel void ref950_sample_frame(__global float4* out, int width, int height,
                                  __global const float* camera_buffer, __global const float* random_buffer,
                                  __global const Scene* scene, __read_only image2d_array_t texture, uint sample) {
    int x = get_global_id(0);
    int y = get_global_id(1);
    if (x >= width || y >= height) return;

    float s = (float)x / (float)width;     // trace: (float)x / (float)get_image_width(image), raytracer.cl:500-501
    float t = (float)y / (float)height;

    vec3 camera_pos = getVec(camera_buffer, 0);
    Ray r_main = genInitRay(camera_buffer, &camera_pos, s, t);
    col c = getCol(&r_main, random_buffer, scene, texture, sample);
    out[(size_t)y * (size_t)width + (size_t)x] = (float4)(c, 1.0f);
}

// One OpenCL builtin per record, as ROCm's OpenCL library defines it (tests compare the HIP kernels' arithmetic
// policies 1 / 2, rt_debug_builtin, with this bit for bit): in n x 8 floats {a.xyz, b.xyz, t, -}, out n x float4.
__kernel void ref950_probe_builtin(int op, __global const float* in, uint n, __global float4* out) {
    uint i = get_global_id(0);
    if (i >= n) return;
    __global const float* a = in + 8 * (size_t)i;
    float3 x = (float3)(a[0], a[1], a[2]), y = (float3)(a[3], a[4], a[5]);
    float t = a[6];
    float4 o = (float4)(0.0f);
    switch (op) {
        case 0: o.x = dot(x, y); break;
        case 1: o.xyz = cross(x, y); break;
        case 2: o.xyz = normalize(x); break;
        case 3: { o.x = a[0] / a[1]; o.y = 1.0f / a[0]; float3 c = x / t; o.z = c.y; o.w = c.z; } break;
        case 4: o.x = sqrt(a[0]); break;
        case 5: o.xyz = mix(x, y, t); break;
        case 6: o.xyz = min(x, y); break;
        case 7: o.x = sign(a[0]); break;
        case 8: o.x = pow(a[0], 5); break;
        case 9: o.x = as_float((uint)fabs(dot(x, (float3)(123.9898, 348.233, 433.3314)) * 438.5453)); break;   // raytracer.cl:114
    }
    out[i] = o;
}
