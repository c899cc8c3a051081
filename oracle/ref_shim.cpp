// ref_shim.cpp — TEST INFRASTRUCTURE (oracle/_ref build).  Not part of the product.
//
// Host-side definitions of the 19 OpenCL builtins that the UNMODIFIED reference
// kernel file needs once it has been compiled for x86-64 where it lies
// (/root/reference/kernels/raytracer.cl, see oracle/Makefile), plus a flat
// extern "C" driver API (ref_*) around the reference's own functions
// `trace`, `retrace`, `createScene`, `getCol`, `genInitRay`, `hitSphere`,
// `hitPlane`, `hitLens`, `hitTriangle`, `hitScene`, ... (raytracer.cl:129-558).
//
// No OpenCL device exists in the build container, so the definitions below are
// STAND-INS written for this build: every one is the plain IEEE-754
// single-precision formula of the OpenCL 1.2 specification, evaluated in a fixed
// order, with no fused multiply-add.  The C restatement (oracle/pt_oracle.c) and
// the HIP kernels use exactly the same formulas — that is the arithmetic contract
// of this build — but it is ONE legal choice, not THE reference's:
// a real OpenCL library is free to fuse, to use rsq / rcp, and ROCm's does
// (profiles/r02_ref_gfx950_builtins.md).  Because of these stand-ins, oracle/_ref
// pins the oracle to "reference source + IEEE-plain builtins" only; by the build
// rules that counts as PARITY UNPINNED (oracle/pt_oracle.c header).
// The library stays in the build container (.gpurunignore).
//
// Built with ROCm's clang++ for x86-64 (same compiler as the .cl object, so the
// ext_vector_type calling convention matches), -O2 -ffp-contract=off.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/rt_amd.h"

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef int int2 __attribute__((ext_vector_type(2)));

// ---- image handles: what image2d_t / image2d_array_t point to here ----------
struct HostImage {
    int w, h, layers;
    float *rgba;  // layers * h * w * 4
};

static thread_local size_t g_gid[2];

// ---- the builtin definitions (mangled names = what the .cl object imports) --
extern "C++" {

size_t cl_get_global_id(unsigned d) asm("_Z13get_global_idj");
size_t cl_get_global_id(unsigned d) { return d < 2 ? g_gid[d] : 0; }

// dot(a,b) = (ax*bx + ay*by) + az*bz
float cl_dot(float3 a, float3 b) asm("_Z3dotDv3_fS_");
float cl_dot(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

float3 cl_cross(float3 a, float3 b) asm("_Z5crossDv3_fS_");
float3 cl_cross(float3 a, float3 b) {
    float3 r;
    r.x = a.y * b.z - a.z * b.y;
    r.y = a.z * b.x - a.x * b.z;
    r.z = a.x * b.y - a.y * b.x;
    return r;
}

float cl_sqrt(float x) asm("_Z4sqrtf");
float cl_sqrt(float x) { return __builtin_sqrtf(x); }  // correctly rounded (sqrtss)

float3 cl_sqrt3(float3 v) asm("_Z4sqrtDv3_f");
float3 cl_sqrt3(float3 v) {
    float3 r;
    r.x = __builtin_sqrtf(v.x);
    r.y = __builtin_sqrtf(v.y);
    r.z = __builtin_sqrtf(v.z);
    return r;
}

// normalize(v) = v / sqrt(dot(v,v)): three IEEE divides, no rsqrt
float3 cl_normalize(float3 v) asm("_Z9normalizeDv3_f");
float3 cl_normalize(float3 v) {
    float len = __builtin_sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z);
    float3 r;
    r.x = v.x / len;
    r.y = v.y / len;
    r.z = v.z / len;
    return r;
}

// min(x,y) = y < x ? y : x (OpenCL 1.2 §6.12.4), component-wise
float3 cl_min3(float3 a, float3 b) asm("_Z3minDv3_fS_");
float3 cl_min3(float3 a, float3 b) {
    float3 r;
    r.x = b.x < a.x ? b.x : a.x;
    r.y = b.y < a.y ? b.y : a.y;
    r.z = b.z < a.z ? b.z : a.z;
    return r;
}

// mix(x,y,a) = x + (y - x) * a
float3 cl_mix3(float3 a, float3 b, float t) asm("_Z3mixDv3_fS_f");
float3 cl_mix3(float3 a, float3 b, float t) {
    float3 r;
    r.x = a.x + (b.x - a.x) * t;
    r.y = a.y + (b.y - a.y) * t;
    r.z = a.z + (b.z - a.z) * t;
    return r;
}

// sign(x): 1 if x>0, -1 if x<0, ±0 for ±0, 0 for NaN
float cl_sign(float x) asm("_Z4signf");
float cl_sign(float x) {
    if (x > 0.0f) return 1.0f;
    if (x < 0.0f) return -1.0f;
    if (x == 0.0f) return x;
    return 0.0f;
}

// pow(x,y): the kernel only ever calls pow(1-cos, 5) (raytracer.cl:404); defined
// as the multiply chain ((x*x)*(x*x))*x for y == 5, libm otherwise.
float cl_pow(float x, float y) asm("_Z3powff");
float cl_pow(float x, float y) {
    if (y == 5.0f) {
        float x2 = x * x;
        return (x2 * x2) * x;
    }
    return std::pow(x, y);
}

double cl_fabsd(double x) asm("_Z4fabsd");
double cl_fabsd(double x) { return __builtin_fabs(x); }

void *cl_translate_sampler(int v) asm("__translate_sampler_initializer");
void *cl_translate_sampler(int v) { return (void *)(intptr_t)v; }

int cl_img_w_ro(HostImage *im) asm("_Z15get_image_width14ocl_image2d_ro");
int cl_img_w_ro(HostImage *im) { return im->w; }
int cl_img_w_wo(HostImage *im) asm("_Z15get_image_width14ocl_image2d_wo");
int cl_img_w_wo(HostImage *im) { return im->w; }
int cl_img_h_ro(HostImage *im) asm("_Z16get_image_height14ocl_image2d_ro");
int cl_img_h_ro(HostImage *im) { return im->h; }
int cl_img_h_wo(HostImage *im) asm("_Z16get_image_height14ocl_image2d_wo");
int cl_img_h_wo(HostImage *im) { return im->h; }

// read_imagef(image2d_t, nearest/unnormalized sampler, int2)
float4 cl_read_imagef_2d(HostImage *im, void *sampler, int2 c)
    asm("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_i");
float4 cl_read_imagef_2d(HostImage *im, void *, int2 c) {
    const float *p = im->rgba + 4 * ((size_t)c.y * im->w + c.x);
    float4 r = {p[0], p[1], p[2], p[3]};
    return r;
}

void cl_write_imagef_2d(HostImage *im, int2 c, float4 v) asm("_Z12write_imagef14ocl_image2d_woDv2_iDv4_f");
void cl_write_imagef_2d(HostImage *im, int2 c, float4 v) {
    float *p = im->rgba + 4 * ((size_t)c.y * im->w + c.x);
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
    p[3] = v.w;
}

// read_imagef(image2d_array_t, linear/normalized sampler, float4(u,v,layer,0)):
// OpenCL 1.2 §8.2 bilinear: u = s*w; i0 = floor(u-0.5); a = (u-0.5)-i0;
// T = (1-a)(1-b)T00 + a(1-b)T10 + (1-a)b T01 + ab T11, summed left to right.
// CLK_ADDRESS_NONE leaves out-of-range texels undefined; this definition
// clamps the four texel coordinates to the edge.  layer = clamp(rint(layer)).
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

float4 cl_read_imagef_2da(HostImage *im, void *sampler, float4 c)
    asm("_Z11read_imagef20ocl_image2d_array_ro11ocl_samplerDv4_f");
float4 cl_read_imagef_2da(HostImage *im, void *, float4 c) {
    int layer = clampi((int)__builtin_rintf(c.z), 0, im->layers - 1);
    float u = c.x * (float)im->w - 0.5f;
    float v = c.y * (float)im->h - 0.5f;
    float fu = __builtin_floorf(u), fv = __builtin_floorf(v);
    float a = u - fu, b = v - fv;
    // NaN / huge coordinates: (int) of those is undefined, pin them to texel 0
    int i0 = (fu >= -1.0f && fu <= 1.0e9f) ? (int)fu : 0;
    int j0 = (fv >= -1.0f && fv <= 1.0e9f) ? (int)fv : 0;
    int i1 = clampi(i0 + 1, 0, im->w - 1), j1 = clampi(j0 + 1, 0, im->h - 1);
    i0 = clampi(i0, 0, im->w - 1);
    j0 = clampi(j0, 0, im->h - 1);
    const float *base = im->rgba + (size_t)layer * im->w * im->h * 4;
    const float *t00 = base + 4 * ((size_t)j0 * im->w + i0);
    const float *t10 = base + 4 * ((size_t)j0 * im->w + i1);
    const float *t01 = base + 4 * ((size_t)j1 * im->w + i0);
    const float *t11 = base + 4 * ((size_t)j1 * im->w + i1);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float4 r;
    r.x = ((w00 * t00[0] + w10 * t10[0]) + w01 * t01[0]) + w11 * t11[0];
    r.y = ((w00 * t00[1] + w10 * t10[1]) + w01 * t01[1]) + w11 * t11[1];
    r.z = ((w00 * t00[2] + w10 * t10[2]) + w01 * t01[2]) + w11 * t11[2];
    r.w = ((w00 * t00[3] + w10 * t10[3]) + w01 * t01[3]) + w11 * t11[3];
    return r;
}

}  // extern "C++"

// ---- the reference's own functions, as the compiled .cl object exports them --
struct ClRay {
    float3 origin;
    float3 dir;
    float param;
};  // raytracer.cl:17-21 (48 bytes)

struct ClHPI {
    float t;
    float3 p;
    float3 normal;
    float2 uv;
    unsigned texture_ID;
    unsigned mat_ID;
};  // raytracer.cl:31-38

struct ClScene {
    const void *materials, *spheres, *planes, *lenses, *vertex_buffer, *texture_uv_buffer, *index_buffer,
        *mesh_buffer, *models;
    unsigned sphere_count, plane_count, lens_count, model_count;
};  // raytracer.cl:74-91 (88 bytes)
static_assert(sizeof(ClScene) == 88, "Scene layout");
static_assert(sizeof(ClRay) == 48, "Ray layout");

struct ClObjectCounter {
    unsigned sphere_count, plane_count, lens_count, model_count;
};  // raytracer.cl:534-539

extern "C" {
void __clang_ocl_kern_imp_trace(HostImage *image, const float *camera, const float *random, const ClScene *scene,
                                HostImage *texture);
void __clang_ocl_kern_imp_retrace(HostImage *image_in, HostImage *image_out, const float *camera, const float *random,
                                  const ClScene *scene, HostImage *texture, unsigned sample);
void __clang_ocl_kern_imp_createScene(ClScene *scene, const void *materials, const void *spheres, const void *planes,
                                      const void *lenses, const void *vertex, const void *uv, const void *index,
                                      const void *mesh, const void *models, ClObjectCounter counter);
float3 getCol(ClRay *r, const float *random, const ClScene *scene, HostImage *texture, unsigned sample);
ClRay genInitRay(const float *camera, const float3 *origin, float s, float t);
bool hitSphere(const ClRay *r, const void *s, ClHPI *hpi);
bool hitPlane(const ClRay *r, const void *p, ClHPI *hpi);
bool hitLens(const ClRay *r, const void *l, ClHPI *hpi);
bool hitTriangle(const ClRay *r, const ClScene *scene, const void *mesh, unsigned a, unsigned b, unsigned c,
                 ClHPI *hpi);
bool hitScene(const ClRay *r, const ClScene *scene, ClHPI *hpi);
void rayReflect(ClRay *r, float3 *c, const ClHPI *hpi, const ClScene *scene);
void rayRefract(ClRay *r, float3 *c, ClHPI *hpi, const ClScene *scene);
void rayScatter(ClRay *r, float3 *c, const ClHPI *hpi, const float *random, unsigned s_seed, const ClScene *scene);
void rayRefractDielectric(ClRay *r, float3 *c, ClHPI *hpi, const float *random, unsigned s_seed, const ClScene *scene);
}

// ---- flat driver API ---------------------------------------------------------
namespace {

struct RefScene {
    ClScene cl;
    HostImage tex;
    float dummy_texel[4];
};

void build_scene(RefScene &rs, const rt_scene_desc *d, const float *tex, int tw, int th, int layers) {
    static const char dummy[64] = {0};
    ClObjectCounter oc = {d->sphere_count, d->plane_count, d->lens_count, d->model_count};
    // the reference's own createScene kernel fills the Scene struct (raytracer.cl:541-558)
    __clang_ocl_kern_imp_createScene(
        &rs.cl, d->materials ? (const void *)d->materials : dummy, d->spheres ? (const void *)d->spheres : dummy,
        d->planes ? (const void *)d->planes : dummy, d->lenses ? (const void *)d->lenses : dummy,
        d->vertices ? (const void *)d->vertices : dummy, d->uvs ? (const void *)d->uvs : dummy,
        d->indices ? (const void *)d->indices : dummy, d->meshes ? (const void *)d->meshes : dummy,
        d->models ? (const void *)d->models : dummy, oc);
    if (tex && layers > 0) {
        rs.tex = HostImage{tw, th, layers, const_cast<float *>(tex)};
    } else {
        rs.dummy_texel[0] = rs.dummy_texel[1] = rs.dummy_texel[2] = rs.dummy_texel[3] = 0.0f;
        rs.tex = HostImage{1, 1, 1, rs.dummy_texel};
    }
}

template <class F>
void parallel_rows(int y0, int h, int threads, F f) {
    if (threads <= 1) {
        for (int y = y0; y < y0 + h; y++) f(y);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([=]() {
            for (int y = y0 + t; y < y0 + h; y += threads) f(y);
        });
    for (auto &th : pool) th.join();
}

}  // namespace

extern "C" {

// one `trace` launch (global size (W,H), raytracer.cl:496); only the work-items of
// the region [x0,x0+cw)×[y0,y0+ch) are executed
int ref_trace(float *image_rgba, int w, int h, int x0, int y0, int cw, int ch, const float *camera,
              const float *table, const rt_scene_desc *scene, const float *tex, int tw, int th, int layers,
              int threads) {
    RefScene rs;
    build_scene(rs, scene, tex, tw, th, layers);
    HostImage img{w, h, 1, image_rgba};
    parallel_rows(y0, ch, threads, [&](int y) {
        for (int x = x0; x < x0 + cw; x++) {
            g_gid[0] = (size_t)x;
            g_gid[1] = (size_t)y;
            HostImage local = img;
            __clang_ocl_kern_imp_trace(&local, camera, table, &rs.cl, &rs.tex);
        }
    });
    return 0;
}

// one `retrace` launch, image_in == image_out as in src/raytracer.cpp:114-115
int ref_retrace(float *image_rgba, int w, int h, int x0, int y0, int cw, int ch, const float *camera,
                const float *table, const rt_scene_desc *scene, const float *tex, int tw, int th, int layers,
                unsigned sample, int threads) {
    RefScene rs;
    build_scene(rs, scene, tex, tw, th, layers);
    HostImage img{w, h, 1, image_rgba};
    parallel_rows(y0, ch, threads, [&](int y) {
        for (int x = x0; x < x0 + cw; x++) {
            g_gid[0] = (size_t)x;
            g_gid[1] = (size_t)y;
            HostImage local = img;
            __clang_ocl_kern_imp_retrace(&local, &local, camera, table, &rs.cl, &rs.tex, sample);
        }
    });
    return 0;
}

// linear radiance of individual pixel-samples: genInitRay + getCol exactly as
// `trace`/`retrace` call them (raytracer.cl:500-507 / 517-528)
int ref_samples(int w, int h, const float *camera, const float *table, const rt_scene_desc *scene, const float *tex,
                int tw, int th, int layers, const uint32_t *xs, const uint32_t *ys, const uint32_t *samples, size_t n,
                float *out_rgb) {
    RefScene rs;
    build_scene(rs, scene, tex, tw, th, layers);
    for (size_t i = 0; i < n; i++) {
        g_gid[0] = xs[i];
        g_gid[1] = ys[i];
        float s = (float)(int)xs[i] / (float)w;
        float t = (float)(int)ys[i] / (float)h;
        float3 cam_pos = {camera[0], camera[1], camera[2]};
        ClRay r = genInitRay(camera, &cam_pos, s, t);
        float3 c = getCol(&r, table, &rs.cl, &rs.tex, samples[i]);
        out_rgb[3 * i + 0] = c.x;
        out_rgb[3 * i + 1] = c.y;
        out_rgb[3 * i + 2] = c.z;
    }
    return 0;
}

// unit probes.  rays: n × 6 floats (origin, dir).  out: n × 12 floats
// {hit, t, p.xyz, normal.xyz, uv.xy, texture_ID(bits), mat_ID(bits)}; fields the
// reference leaves unwritten are reported as 0.
static void put_hpi(float *o, bool hit, const ClHPI &h) {
    o[0] = hit ? 1.0f : 0.0f;
    o[1] = h.t;
    o[2] = h.p.x; o[3] = h.p.y; o[4] = h.p.z;
    o[5] = h.normal.x; o[6] = h.normal.y; o[7] = h.normal.z;
    o[8] = h.uv.x; o[9] = h.uv.y;
    memcpy(o + 10, &h.texture_ID, 4);
    memcpy(o + 11, &h.mat_ID, 4);
}
static ClRay make_ray(const float *r) {
    ClRay ray;
    ray.origin = float3{r[0], r[1], r[2]};
    ray.dir = float3{r[3], r[4], r[5]};
    ray.param = 0.0f;
    return ray;
}

// kind: 0 sphere, 1 plane, 2 lens — primitive i of the scene arrays, or 3 = hitScene
int ref_hit(int kind, const rt_scene_desc *scene, const float *rays, const uint32_t *prim, size_t n, float *out) {
    RefScene rs;
    build_scene(rs, scene, nullptr, 0, 0, 0);
    for (size_t i = 0; i < n; i++) {
        ClRay ray = make_ray(rays + 6 * i);
        ClHPI h;
        memset(&h, 0, sizeof h);
        bool hit = false;
        switch (kind) {
            case 0: hit = hitSphere(&ray, scene->spheres + prim[i], &h); break;
            case 1: hit = hitPlane(&ray, scene->planes + prim[i], &h); break;
            case 2: hit = hitLens(&ray, scene->lenses + prim[i], &h); break;
            case 3: hit = hitScene(&ray, &rs.cl, &h); break;
            default: return -1;
        }
        if (!hit) memset(&h, 0, sizeof h);
        put_hpi(out + 12 * i, hit, h);
    }
    return 0;
}

// hitTriangle on face `face[i]` of mesh `mesh[i]`
int ref_hit_triangle(const rt_scene_desc *scene, const float *rays, const uint32_t *mesh, const uint32_t *face,
                     size_t n, float *out) {
    RefScene rs;
    build_scene(rs, scene, nullptr, 0, 0, 0);
    for (size_t i = 0; i < n; i++) {
        ClRay ray = make_ray(rays + 6 * i);
        ClHPI h;
        memset(&h, 0, sizeof h);
        bool hit = hitTriangle(&ray, &rs.cl, scene->meshes + mesh[i], 3 * face[i], 3 * face[i] + 1, 3 * face[i] + 2, &h);
        if (!hit) memset(&h, 0, sizeof h);
        put_hpi(out + 12 * i, hit, h);
    }
    return 0;
}

// material routines (raytracer.cl:362-435) on their own.  in: n × 16 floats
// {ray dir.xyz, hit p.xyz, hit normal.xyz, colour so far.xyz, mat_ID bits, s_seed bits, gid0 bits, gid1 bits};
// out: n × 9 floats {new origin.xyz, new dir.xyz, colour.xyz}.  routine: 0 rayReflect, 1 rayRefract,
// 2 rayScatter, 3 rayRefractDielectric.  (mixCol is getCol's, not the routines': not applied.)
int ref_material(int routine, const rt_scene_desc *scene, const float *table, const float *in, size_t n, float *out) {
    RefScene rs;
    build_scene(rs, scene, nullptr, 0, 0, 0);
    for (size_t i = 0; i < n; i++) {
        const float *v = in + 16 * i;
        ClRay r;
        r.origin = float3{0.0f, 0.0f, 0.0f};
        r.dir = float3{v[0], v[1], v[2]};
        r.param = 0.0f;
        ClHPI h;
        memset(&h, 0, sizeof h);
        h.p = float3{v[3], v[4], v[5]};
        h.normal = float3{v[6], v[7], v[8]};
        float3 c = {v[9], v[10], v[11]};
        unsigned seed, gx, gy;
        memcpy(&h.mat_ID, v + 12, 4);
        memcpy(&seed, v + 13, 4);
        memcpy(&gx, v + 14, 4);
        memcpy(&gy, v + 15, 4);
        g_gid[0] = gx;
        g_gid[1] = gy;
        switch (routine) {
            case 0: rayReflect(&r, &c, &h, &rs.cl); break;
            case 1: rayRefract(&r, &c, &h, &rs.cl); break;
            case 2: rayScatter(&r, &c, &h, table, seed, &rs.cl); break;
            case 3: rayRefractDielectric(&r, &c, &h, table, seed, &rs.cl); break;
            default: return -1;
        }
        float *o = out + 9 * i;
        o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
        o[3] = r.dir.x; o[4] = r.dir.y; o[5] = r.dir.z;
        o[6] = c.x; o[7] = c.y; o[8] = c.z;
    }
    return 0;
}

}  // extern "C"
