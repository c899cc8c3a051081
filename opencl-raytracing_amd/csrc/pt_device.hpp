// pt_device.hpp — device-side path tracing for gfx950 (MI355X), wave64.
//
// Hand-written HIP equivalent of everything kernels `trace` / `retrace` of the
// reference reach (kernels/raytracer.cl:93-494; line numbers below are that
// file's).  Results are bit-identical per pixel-sample to the reference
// compiled without FMA contraction: every float operation below is a single
// IEEE-754 binary32 operation in the reference's evaluation order.  This file
// must be compiled with -ffp-contract=off and without any fast-math flag.
//
// Mapping to the hardware:
//   * one work-item per pixel-sample; the 64 lanes of a wave are 64 samples of
//     one pixel (or 64/g pixels × g samples), so primitive indices are
//     wave-uniform: primitive records are fetched with SCALAR loads (s_load,
//     scalar cache → L2) and cost no VGPRs and no LDS bandwidth;
//   * the nearest-hit search keeps only (t, id) per lane and rebuilds the hit
//     record of the winner afterwards (same arithmetic → same bits);
//   * table / texture gathers are the only divergent memory accesses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_amd.h"

namespace pt {

struct V3 {
    float x, y, z;
};

#define PT_DEV __device__ __forceinline__

PT_DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
PT_DEV V3 ld3(const rt_float3 &f) { return V3{f.x, f.y, f.z}; }
PT_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV V3 operator*(V3 a, float k) { return V3{a.x * k, a.y * k, a.z * k}; }
PT_DEV V3 operator/(V3 a, float k) { return V3{a.x / k, a.y / k, a.z / k}; }
PT_DEV V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }
// dot(a,b) = (ax*bx + ay*by) + az*bz — the builtin definition shared with the oracle
PT_DEV float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PT_DEV V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// normalize(v) = v / sqrt(dot(v,v)); sqrtf and '/' are correctly rounded in HIP
// (-fhip-fp32-correctly-rounded-divide-sqrt is the default and is passed explicitly)
PT_DEV V3 normalize(V3 a) { return a / sqrtf(dot(a, a)); }
PT_DEV V3 vmin(V3 a, V3 b) { return V3{b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z}; }
PT_DEV float sign1(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : (x == 0.0f ? x : 0.0f)); }
PT_DEV float pow5(float x) {
    float x2 = x * x;
    return (x2 * x2) * x;
}
// :127 — false for NaN
PT_DEV bool in_range(float x) { return (x - RT_MAX_DISTANCE) * (x - RT_MIN_DISTANCE) <= 0.0f; }

struct Ray {
    V3 o, d;
};
PT_DEV V3 point_at(const Ray &r, float t) { return r.o + r.d * t; }  // :141

// Everything the kernels read.  Passed by value (kernarg segment → SGPRs); this
// replaces the reference's device-resident Scene struct and its createScene
// pointer-stashing kernel (:74-91, :541-558).
struct DeviceScene {
    const rt_material *materials;
    const rt_sphere *spheres;
    const rt_plane *planes;
    const rt_lens *lenses;
    const rt_float3 *vertices;
    const rt_float2 *uvs;
    const uint32_t *indices;
    const rt_mesh *meshes;
    const rt_model *models;
    const float *table;   // 400 000 floats
    const float4 *tex;    // layers × h × w texels
    int tex_w, tex_h, tex_layers;
    uint32_t sphere_count, plane_count, lens_count, model_count;
};

// per-lane work counters (only in COUNT builds)
struct LaneCounters {
    uint32_t c[14];
};
enum {
    CN_SAMPLES, CN_BOUNCES, CN_T_SPHERE, CN_T_PLANE, CN_T_LENS, CN_T_MODEL, CN_T_MESH, CN_T_TRI, CN_H_TRI,
    CN_H_BOUNCE, CN_N_SCATTER, CN_N_DIELECTRIC, CN_N_TEXFETCH, CN_IMAGE_READS
};

// ---- table index (:113-125) --------------------------------------------------
// fp64 hash of the direction; the index sum is below 2^32 for w,h <= RT_MAX_DIM
// and sample <= RT_MAX_SAMPLE + RT_DEPTH, so 32-bit arithmetic equals the
// reference's 64-bit size_t arithmetic.
PT_DEV uint32_t dir_hash(V3 d) {
    float dp = dot(d, mk(123.9898f, 348.233f, 433.3314f));
    return (uint32_t)fabs((double)dp * 438.5453);
}
PT_DEV V3 random_vec(const float *__restrict__ table, V3 dir, uint32_t s_seed, uint32_t gx, uint32_t gy) {
    uint32_t idx = (dir_hash(dir) + (s_seed * 2683u + gx * 3931u + gy * 2504u) * 3u) % RT_RANDOM_BUFFER_SIZE;
    const float *t = table + idx;  // three consecutive FLOATS at float offset idx (:109-111,117)
    return mk(t[0], t[1], t[2]);
}
PT_DEV float random_u(const float *__restrict__ table, V3 dir, uint32_t s_seed, uint32_t gx, uint32_t gy) {
    uint32_t idx = (dir_hash(dir) + (s_seed * 2683u + gx * 3931u + gy)) % RT_RANDOM_BUFFER_SIZE;
    return table[3 * RT_RANDOM_BUFFER_SIZE + idx];
}

// ---- nearest-hit search --------------------------------------------------------
// id of the winning primitive: kind in the top 2 bits
enum : uint32_t { K_SPHERE = 0u << 30, K_PLANE = 1u << 30, K_LENS = 2u << 30, K_MESH = 3u << 30, K_MASK = 3u << 30 };

struct Best {
    float t;       // nearest t so far (starts at MAX_DISTANCE)
    uint32_t id;   // kind | index (sphere/plane/lens index, or mesh index for K_MESH)
    uint32_t face; // K_MESH: face index inside the mesh
    uint32_t mat;  // K_MESH: material of the owning model
    float u, v;    // K_MESH: barycentrics of the hit
};

// :149-174 — returns the accepted root or a negative number
PT_DEV float sphere_t(const Ray &r, V3 c, float rad) {
    V3 oc = c - r.o;
    float b = dot(oc, r.d);
    float cc = dot(oc, oc) - rad * rad;
    float dis = b * b - cc;
    float t = -1.0f;
    if (dis > 0) {
        float d = sqrtf(dis);
        float t0 = b - d;
        if (in_range(t0)) t = t0;
        else {
            float t1 = b + d;
            if (in_range(t1)) t = t1;
        }
    }
    return t;  // accepted roots are >= MIN_DISTANCE > 0
}

// :176-194
PT_DEV float plane_t(const Ray &r, V3 p0, V3 n, float *a_out) {
    float a = dot(r.d, n);
    float b = dot(p0 - r.o, n);
    float t = b / a;
    *a_out = a;
    return in_range(t) ? t : -1.0f;
}

// :196-255 — intersection of two spheres; which = 0 → surface 1 (p1,r1), 1 → surface 2
PT_DEV float lens_t(const Ray &r, const rt_lens &l, int *which) {
    V3 oc = ld3(l.p1) - r.o;
    float b1 = dot(oc, r.d);
    float c = dot(oc, oc) - l.r1 * l.r1;
    float dis1 = b1 * b1 - c;
    oc = ld3(l.p2) - r.o;
    float b2 = dot(oc, r.d);
    c = dot(oc, oc) - l.r2 * l.r2;
    float dis2 = b2 * b2 - c;
    if (dis1 > 0 && dis2 > 0) {
        float d1 = sqrtf(dis1), d2 = sqrtf(dis2);
        float t1A = b1 - d1, t1B = b1 + d1, t2A = b2 - d2, t2B = b2 + d2;
        float t;
        int w;
        if ((t1B < t2A) || (t2B < t1A)) return -1.0f;
        else if (RT_MIN_DISTANCE <= t1A || RT_MIN_DISTANCE <= t2A) {
            if (t2A <= t1A) { w = 0; t = t1A; } else { w = 1; t = t2A; }
        } else if (RT_MIN_DISTANCE <= t1B && RT_MIN_DISTANCE <= t2B) {
            if (t1B <= t2B) { w = 0; t = t1B; } else { w = 1; t = t2B; }
        } else return -1.0f;
        if (t <= RT_MAX_DISTANCE) {
            *which = w;
            return t;
        }
    }
    return -1.0f;
}

// :257-289 — Möller–Trumbore without culling.  Returns t (>0) on a hit.
PT_DEV float triangle_t(const Ray &r, V3 A, V3 e1, V3 e2, float *u_out, float *v_out) {
    V3 h = cross(r.d, e2);
    float a = dot(e1, h);
    if (a > -RT_TRIANGLE_EPSILON && a < RT_TRIANGLE_EPSILON) return -1.0f;
    float f = 1.0f / a;
    V3 s = r.o - A;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return -1.0f;
    V3 q = cross(s, e1);
    float v = f * dot(r.d, q);
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    float t = f * dot(e2, q);
    if (!in_range(t)) return -1.0f;
    *u_out = u;
    *v_out = v;
    return t;
}

struct Hit {
    V3 p, n;
    float u, v;     // texture coordinates (mesh hits)
    uint32_t tex;
    uint32_t mat;
};

// :322-360 hitScene, :305-320 hitModel, :291-303 hitMeshOut.
// Nearest over spheres → planes → lenses → models with strict '<' (the earlier
// primitive keeps a tie); inside a mesh the FIRST front-facing hit in face
// order wins, not the nearest.
template <bool COUNT>
PT_DEV bool hit_scene(const DeviceScene &sc, const Ray &r, Hit &hit, LaneCounters *cn) {
    Best best;
    best.t = RT_MAX_DISTANCE;
    best.id = 0xFFFFFFFFu;
    best.face = 0;
    best.mat = 0;
    best.u = best.v = 0.0f;

    for (uint32_t i = 0; i < sc.sphere_count; i++) {
        const rt_sphere &s = sc.spheres[i];  // wave-uniform index → scalar loads
        float t = sphere_t(r, ld3(s.pos), s.r);
        if (t > 0.0f && t < best.t) {
            best.t = t;
            best.id = K_SPHERE | i;
        }
    }
    for (uint32_t i = 0; i < sc.plane_count; i++) {
        const rt_plane &p = sc.planes[i];
        float a;
        float t = plane_t(r, ld3(p.pos), ld3(p.normal), &a);
        if (t > 0.0f && t < best.t) {
            best.t = t;
            best.id = K_PLANE | i;
        }
    }
    for (uint32_t i = 0; i < sc.lens_count; i++) {
        int which;
        float t = lens_t(r, sc.lenses[i], &which);
        if (t > 0.0f && t < best.t) {
            best.t = t;
            best.id = K_LENS | i;
        }
    }
    for (uint32_t mo = 0; mo < sc.model_count; mo++) {
        const rt_model &model = sc.models[mo];
        float model_best = RT_MAX_DISTANCE;  // hitModel's own hit_min (:307)
        bool model_hit = false;
        Best cand = best;
        for (uint32_t k = 0; k < model.mesh_count; k++) {
            uint32_t mi = model.mesh_anchor + k;
            const rt_mesh &mesh = sc.meshes[mi];
            if (COUNT) cn->c[CN_T_MESH]++;
            // hitMeshOut: scan faces until this lane has its first front-facing hit
            bool found = false;
            float ft = 0.0f, fu = 0.0f, fv = 0.0f;
            uint32_t fface = 0;
            for (uint32_t f = 0; f < mesh.face_count; f++) {
                if (found) continue;  // this lane is done with the mesh; others keep scanning
                if (COUNT) cn->c[CN_T_TRI]++;
                const uint32_t *ib = sc.indices + mesh.index_anchor + 3u * f;
                V3 A = ld3(sc.vertices[mesh.vertex_anchor + ib[0]]);
                V3 B = ld3(sc.vertices[mesh.vertex_anchor + ib[1]]);
                V3 C = ld3(sc.vertices[mesh.vertex_anchor + ib[2]]);
                V3 e1 = B - A, e2 = C - A;
                float u, v;
                float t = triangle_t(r, A, e1, e2, &u, &v);
                if (t > 0.0f) {
                    if (COUNT) cn->c[CN_H_TRI]++;
                    V3 n = normalize(cross(e1, e2));
                    if (dot(n, r.d) < 0.0f) {
                        found = true;
                        ft = t; fu = u; fv = v; fface = f;
                    }
                }
            }
            if (found && ft < model_best) {
                model_best = ft;
                model_hit = true;
                cand.t = ft;
                cand.id = K_MESH | mi;
                cand.face = fface;
                cand.mat = model.mat_ID;
                cand.u = fu;
                cand.v = fv;
            }
        }
        if (model_hit && cand.t < best.t) best = cand;
    }

    if (COUNT) {
        cn->c[CN_BOUNCES]++;
        cn->c[CN_T_SPHERE] += sc.sphere_count;
        cn->c[CN_T_PLANE] += sc.plane_count;
        cn->c[CN_T_LENS] += sc.lens_count;
        cn->c[CN_T_MODEL] += sc.model_count;
    }
    if (best.id == 0xFFFFFFFFu) return false;

    // rebuild the winner's record with the reference's arithmetic
    uint32_t kind = best.id & K_MASK, idx = best.id & ~K_MASK;
    hit.p = point_at(r, best.t);
    hit.u = hit.v = 0.0f;
    hit.tex = 0;
    if (kind == K_SPHERE) {
        const rt_sphere &s = sc.spheres[idx];
        hit.n = (hit.p - ld3(s.pos)) / s.r;  // :160
        hit.mat = s.mat_ID;
    } else if (kind == K_PLANE) {
        const rt_plane &p = sc.planes[idx];
        V3 n = ld3(p.normal);
        hit.n = neg(n) * sign1(dot(r.d, n));  // :187
        hit.mat = p.mat_ID;
    } else if (kind == K_LENS) {
        const rt_lens &l = sc.lenses[idx];
        int which = 0;
        (void)lens_t(r, l, &which);
        hit.n = which == 0 ? (hit.p - ld3(l.p1)) / l.r1 : (hit.p - ld3(l.p2)) / l.r2;  // :248
        hit.mat = l.mat_ID;
    } else {
        const rt_mesh &mesh = sc.meshes[idx];
        const uint32_t *ib = sc.indices + mesh.index_anchor + 3u * best.face;
        uint32_t ia = mesh.vertex_anchor + ib[0], ibx = mesh.vertex_anchor + ib[1], ic = mesh.vertex_anchor + ib[2];
        V3 A = ld3(sc.vertices[ia]), B = ld3(sc.vertices[ibx]), C = ld3(sc.vertices[ic]);
        hit.n = normalize(cross(B - A, C - A));  // :285
        rt_float2 ua = sc.uvs[ia], ub = sc.uvs[ibx], uc = sc.uvs[ic];
        float wgt = 1.0f - best.u - best.v;  // :102
        hit.u = (ua.x * wgt + ub.x * best.u) + uc.x * best.v;
        hit.v = (ua.y * wgt + ub.y * best.u) + uc.y * best.v;
        hit.tex = mesh.texture_ID;
        hit.mat = best.mat;
    }
    return true;
}

// ---- materials -------------------------------------------------------------------
// :362-367
PT_DEV void reflect(Ray &r, V3 &c, const Hit &h, V3 n, int type, float extra) {
    r.o = h.p;
    float k = 2.0f * dot(r.d, n);
    r.d = normalize(r.d - n * k);
    if (type == RT_REFLECTIVE) c = c * extra;
}

// shared front of :369-381 / :407-418
PT_DEV void facing(const Ray &r, const Hit &h, float extra, V3 &n, float &ratio, float &cai) {
    cai = dot(r.d, h.n);
    if (cai > 0) {
        n = neg(h.n);
        ratio = extra;
        cai = -cai;
    } else {
        n = h.n;
        ratio = 1.0f / extra;
    }
}

// :382-386 / :424-429 — the refracted direction is not renormalised
PT_DEV bool try_refract(Ray &r, const Hit &h, V3 n, float ratio, float cai) {
    float disc = 1.0f - ratio * ratio * (1.0f - cai * cai);
    if (disc > 0.0f) {
        r.o = h.p;
        r.d = r.d * ratio - n * (ratio * cai + sqrtf(disc));
        return true;
    }
    return false;
}

// :401-405
PT_DEV float schlick(float cosine, float ratio) {
    float r0 = (1.0f - ratio) / (1.0f + ratio);
    r0 *= r0;
    return r0 + (1.0f - r0) * pow5(1.0f - cosine);
}

// :105-107 with the bilinear definition of DESIGN.md (OpenCL 1.2 §8.2, edge clamp)
PT_DEV int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
PT_DEV V3 texture_rgb(const DeviceScene &sc, float s, float t, uint32_t tex_id) {
    if (sc.tex_layers <= 0) return mk(0.0f, 0.0f, 0.0f);
    int W = sc.tex_w, H = sc.tex_h;
    int layer = tex_id < (uint32_t)sc.tex_layers ? (int)tex_id : 0;
    float u = s * (float)W - 0.5f, v = t * (float)H - 0.5f;
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    int i0 = (fu >= -1.0f && fu <= 1.0e9f) ? (int)fu : 0;
    int j0 = (fv >= -1.0f && fv <= 1.0e9f) ? (int)fv : 0;
    int i1 = clampi(i0 + 1, 0, W - 1), j1 = clampi(j0 + 1, 0, H - 1);
    i0 = clampi(i0, 0, W - 1);
    j0 = clampi(j0, 0, H - 1);
    const float4 *base = sc.tex + (size_t)layer * W * H;
    float4 t00 = base[(size_t)j0 * W + i0], t10 = base[(size_t)j0 * W + i1];
    float4 t01 = base[(size_t)j1 * W + i0], t11 = base[(size_t)j1 * W + i1];
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return mk(((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x,
              ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y,
              ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z);
}

// :444-486 getCol — colour mixing is min(), the sky is black, a path that
// survives DEPTH bounces returns what it has.
template <bool COUNT>
PT_DEV V3 radiance(const DeviceScene &sc, Ray r, uint32_t sample, uint32_t gx, uint32_t gy, LaneCounters *cn) {
    V3 out = mk(1.0f, 1.0f, 1.0f);
    for (uint32_t i = 0; i < RT_DEPTH; i++) {
        Hit h;
        if (!hit_scene<COUNT>(sc, r, h, cn)) return mk(0.0f, 0.0f, 0.0f);
        if (COUNT) cn->c[CN_H_BOUNCE]++;
        const rt_material &m = sc.materials[h.mat];  // per-lane index → vector loads
        int type = m.type;
        float extra = m.extra_data;
        V3 col = ld3(m.color);
        if (type == RT_LIGHT) return vmin(out, col);
        if (type == RT_DIFFUSE || type == RT_TEXTURED) {  // :393-399 rayScatter
            if (COUNT) cn->c[CN_N_SCATTER]++;
            V3 rv = random_vec(sc.table, r.d, i + sample, gx, gy);
            r.d = normalize(h.n + rv);
            r.o = h.p;
            out = out * extra;
            if (type == RT_TEXTURED) {
                if (COUNT) cn->c[CN_N_TEXFETCH]++;
                col = texture_rgb(sc, h.u, h.v, h.tex);
            }
        } else if (type == RT_REFLECTIVE) {
            reflect(r, out, h, h.n, type, extra);
        } else if (type == RT_REFRACTIVE) {  // :369-391
            V3 n;
            float ratio, cai;
            facing(r, h, extra, n, ratio, cai);
            if (!try_refract(r, h, n, ratio, cai)) reflect(r, out, h, n, type, extra);
        } else if (type == RT_DIELECTRIC) {  // :407-435
            if (COUNT) cn->c[CN_N_DIELECTRIC]++;
            V3 n;
            float ratio, cai;
            facing(r, h, extra, n, ratio, cai);
            float prob = schlick(-cai, ratio);
            float rnd = random_u(sc.table, r.d, i + sample, gx, gy);
            if (!(prob < rnd && try_refract(r, h, n, ratio, cai))) reflect(r, out, h, n, type, extra);
        } else {
            continue;  // unknown type: the reference's switch has no default (rejected by rt_set_scene)
        }
        out = vmin(out, col);
    }
    return out;
}

// :129-139, :500-505 — no pixel jitter
PT_DEV Ray primary_ray(const float *cam, uint32_t x, uint32_t y, int w, int h) {
    float s = (float)(int)x / (float)w;
    float t = (float)(int)y / (float)h;
    Ray r;
    r.o = mk(cam[0], cam[1], cam[2]);
    V3 llc = mk(cam[3], cam[4], cam[5]), hor = mk(cam[6], cam[7], cam[8]), ver = mk(cam[9], cam[10], cam[11]);
    r.d = normalize((llc + hor * s) + ver * t);
    return r;
}

}  // namespace pt
