// ref_gfx950_host.cpp — TEST INFRASTRUCTURE: runs the gfx950 build of the reference's kernel file
// (oracle/_ref_gfx950/ref950.hsaco: kernels/raytracer.cl compiled by ROCm's OpenCL tool chain with ROCm's
// real OpenCL builtin library, see ref_gfx950_wrap.cl and oracle/Makefile) on the GPU through the HIP
// module API.  It answers one question: how far is the arithmetic contract the oracle is
// pinned to — "reference source + IEEE-plain builtins, no FMA" — from what the reference computes when AMD's
// own OpenCL compiler builds it for this very chip (fma-contracted dot, rsq-based normalize, rcp-based
// divide)?  Nothing in the product, and no pass/fail parity test, depends on this file.
//
// The scene arrays are uploaded in the byte layouts of include/rt_amd.h, which ARE the reference's device
// struct layouts (raytracer.cl:25-91; tests/test_host.py checks sizes and offsets), the reference's own
// createScene kernel (:541-558) stashes the pointers, and ref950_sample_frame (our wrapper) calls the
// reference's genInitRay + getCol once per pixel for one sample.
//
//   hipcc -O2 -fPIC -shared oracle/ref_gfx950_host.cpp -o oracle/_ref_gfx950/libref950.so
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../include/rt_amd.h"

namespace {

std::string g_err;
// one entry per code object (the default build and the -ffp-contract=off build can be open side by side)
struct Module {
    std::string path;
    hipModule_t mod = nullptr;
    hipFunction_t create = nullptr, frame = nullptr, builtin = nullptr;
};
std::vector<Module> g_mods;

int fail(const char *what, hipError_t e) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
}
#define TRY(x)                                  \
    do {                                        \
        hipError_t e_ = (x);                    \
        if (e_ != hipSuccess) return fail(#x, e_); \
    } while (0)

__global__ void acc_add(float4 *acc, const float4 *frame, size_t n) {
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 a = acc[i], f = frame[i];
    acc[i] = make_float4(a.x + f.x, a.y + f.y, a.z + f.z, a.w + 1.0f);
}

struct Dev {
    void *p = nullptr;
    hipError_t put(const void *src, size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);  // empty arrays become dummies (src/scene.cpp:41-44)
        if (e == hipSuccess && bytes) e = hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
        return e;
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
    ~Dev() { if (p) (void)hipFree(p); }
};

}  // namespace

extern "C" {

const char *ref950_last_error(void) { return g_err.c_str(); }

// → handle (>= 0) of the code object, or -1
int ref950_open(const char *hsaco_path, int device) {
    if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return -1; }
    for (size_t i = 0; i < g_mods.size(); i++)
        if (g_mods[i].path == hsaco_path) return (int)i;
    Module m;
    m.path = hsaco_path;
    if (hipModuleLoad(&m.mod, hsaco_path) != hipSuccess) { g_err = std::string("hipModuleLoad failed: ") + hsaco_path; return -1; }
    if (hipModuleGetFunction(&m.create, m.mod, "createScene") != hipSuccess ||
        hipModuleGetFunction(&m.frame, m.mod, "ref950_sample_frame") != hipSuccess ||
        hipModuleGetFunction(&m.builtin, m.mod, "ref950_probe_builtin") != hipSuccess) { g_err = "kernel lookup failed"; return -1; }
    g_mods.push_back(m);
    return (int)g_mods.size() - 1;
}

// One OpenCL builtin per record, evaluated by ROCm's OpenCL library inside the code object (ref_gfx950_wrap.cl
// ref950_probe_builtin): in n × 8 floats, out n × 4 floats.
int ref950_builtin(int handle, int op, const float *in8, size_t n, float *out4) {
    if (handle < 0 || (size_t)handle >= g_mods.size()) { g_err = "ref950_open first"; return 1; }
    if (!in8 || !out4 || n == 0 || n > (1u << 26)) { g_err = "bad arguments"; return 1; }
    Dev din, dout;
    TRY(din.put(in8, n * 8 * sizeof(float)));
    TRY(dout.alloc(n * 16));
    uint32_t nn = (uint32_t)n;
    void *args[] = {&op, &din.p, &nn, &dout.p};
    TRY(hipModuleLaunchKernel(g_mods[handle].builtin, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, nullptr, args, nullptr));
    TRY(hipDeviceSynchronize());
    TRY(hipMemcpy(out4, dout.p, n * 16, hipMemcpyDeviceToHost));
    return 0;
}

// Linear radiance of samples first .. first+count-1 of every pixel.
//   out_sum   w*h*4 floats: per pixel the sum over the samples in .xyz and the count in .w
//   out_last  optional, w*h*4 floats: the frame of the last sample alone (count = 1 → that sample's frame)
// grid_w × grid_h (0 = the whole frame): only the pixels x < grid_w, y < grid_h are launched — the frame's corner at
// the origin, with the whole frame's width / height in the primary rays and the table index (get_global_id is the
// pixel coordinate, raytracer.cl:115; a HIP launch cannot give an OpenCL kernel a global offset).  Other pixels stay 0.
int ref950_render(int handle, const rt_scene_desc *d, const float cam[12], const float *table, int w, int h, uint32_t first,
                  uint32_t count, int grid_w, int grid_h, float *out_sum, float *out_last) {
    if (handle < 0 || (size_t)handle >= g_mods.size()) { g_err = "ref950_open first"; return 1; }
    hipFunction_t g_create = g_mods[handle].create, g_frame = g_mods[handle].frame;
    if (grid_w <= 0 || grid_w > w) grid_w = w;
    if (grid_h <= 0 || grid_h > h) grid_h = h;
    if (!d || !cam || !table || w < 1 || h < 1 || !out_sum) { g_err = "bad arguments"; return 1; }
    Dev mats, sph, pla, len, vtx, uv, idx, mesh, mod, scene, dcam, dtab, frame, acc;
    TRY(mats.put(d->materials, (size_t)d->material_count * sizeof(rt_material)));
    TRY(sph.put(d->spheres, (size_t)d->sphere_count * sizeof(rt_sphere)));
    TRY(pla.put(d->planes, (size_t)d->plane_count * sizeof(rt_plane)));
    TRY(len.put(d->lenses, (size_t)d->lens_count * sizeof(rt_lens)));
    TRY(vtx.put(d->vertices, (size_t)d->vertex_count * sizeof(rt_float3)));
    std::vector<rt_float2> uvz;
    const rt_float2 *uvp = d->uvs;
    if (d->uv_count == 0 && d->vertex_count) {  // a short uv array is read out of bounds by the reference; zero-filled here
        uvz.assign(d->vertex_count, rt_float2{0.0f, 0.0f});
        uvp = uvz.data();
    }
    TRY(uv.put(uvp, (size_t)d->vertex_count * sizeof(rt_float2)));
    TRY(idx.put(d->indices, (size_t)d->index_count * sizeof(uint32_t)));
    TRY(mesh.put(d->meshes, (size_t)d->mesh_count * sizeof(rt_mesh)));
    TRY(mod.put(d->models, (size_t)d->model_count * sizeof(rt_model)));
    TRY(scene.alloc(256));  // Scene is 88 bytes (raytracer.cl:74-91)
    TRY(dcam.put(cam, 12 * sizeof(float)));
    TRY(dtab.put(table, (size_t)RT_RANDOM_TABLE_FLOATS * sizeof(float)));
    const size_t n = (size_t)w * h;
    TRY(frame.alloc(n * 16));
    TRY(acc.alloc(n * 16));
    TRY(hipMemset(frame.p, 0, n * 16));
    TRY(hipMemset(acc.p, 0, n * 16));

    // createScene(scene, materials, spheres, planes, lenses, vertices, uvs, indices, meshes, models, ObjectCounter)
    struct { uint32_t sphere_count, plane_count, lens_count, model_count; } oc = {d->sphere_count, d->plane_count,
                                                                                d->lens_count, d->model_count};
    void *cargs[] = {&scene.p, &mats.p, &sph.p, &pla.p, &len.p, &vtx.p, &uv.p, &idx.p, &mesh.p, &mod.p, &oc};
    TRY(hipModuleLaunchKernel(g_create, 1, 1, 1, 1, 1, 1, 0, nullptr, cargs, nullptr));

    void *null_image = nullptr;  // image2d_array_t: only t_textured dereferences it
    for (uint32_t k = 0; k < count; k++) {
        uint32_t sample = first + k;
        void *fargs[] = {&frame.p, &w, &h, &dcam.p, &dtab.p, &scene.p, &null_image, &sample};
        // the reference enqueues global size (W, H) with the runtime's choice of local size (src/raytracer.cpp:137)
        TRY(hipModuleLaunchKernel(g_frame, (unsigned)((grid_w + 63) / 64), (unsigned)((grid_h + 3) / 4), 1, 64, 4, 1, 0, nullptr,
                                  fargs, nullptr));
        hipLaunchKernelGGL(acc_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (float4 *)acc.p,
                           (const float4 *)frame.p, n);
    }
    TRY(hipDeviceSynchronize());
    TRY(hipMemcpy(out_sum, acc.p, n * 16, hipMemcpyDeviceToHost));
    if (out_last) TRY(hipMemcpy(out_last, frame.p, n * 16, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
