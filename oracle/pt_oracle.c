/*
 * pt_oracle.c — TEST INFRASTRUCTURE: CPU restatement of the reference's hot path.
 *
 * This file is the parity oracle of the MI355X path tracer.  It is NOT part of
 * the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it, and only as the checker / the CPU number printed beside the
 * GPU one.  The product (opencl-raytracing_amd/csrc) neither includes, links
 * nor calls anything in oracle/.
 *
 * What it restates: kernels `trace` and `retrace` of the reference and every
 * function they reach (kernels/raytracer.cl:93-532; citations below are lines
 * of that file).  It is written from the behavioural description in SURVEY.md
 * §8a as scalar C — not a transliteration of the OpenCL source.
 *
 * PARITY UNPINNED (in the sense of the build rules): the reference ships no tests,
 * golden vectors or fixtures, and no OpenCL runtime for x86 exists in the build
 * container.  What this oracle is checked against — bit for bit per pixel-sample
 * (tests/test_oracle_vs_ref.py) and through tests/golden/ — is the UNMODIFIED
 * reference kernel file compiled for x86-64 (oracle/_ref) linked with the 19 OpenCL
 * builtins of oracle/ref_shim.cpp, which were WRITTEN FOR THIS BUILD as the plain
 * IEEE-754 formula of each builtin.  So the pin is "reference source + IEEE-plain
 * builtins, no FMA", not an execution of the reference under a real OpenCL runtime.
 * A real one exists for the GPU: ROCm's own OpenCL tool chain builds the same file
 * for gfx950 with its real builtin library (oracle/_ref_gfx950); its dot() is an
 * fma chain, its normalize() rsq-based, its '/' rcp-based
 * (profiles/r02_ref_gfx950_builtins.md), so it cannot agree bit for bit; the
 * measured distance on the MI355X — 97.7 % of C2's pixel-samples identical, frame
 * means 2.5e-5 apart, inside the Monte-Carlo noise — is in
 * profiles/r02_ref_distance_*.json and DESIGN.md §3.
 *
 * Arithmetic contract (shared with oracle/ref_shim.cpp and the HIP kernels):
 * IEEE-754 binary32, round to nearest even, no FMA contraction, no
 * reassociation; builtins as defined in ref_shim.cpp.  Build: gcc -O2
 * -ffp-contract=off, baseline x86-64 (SSE2 scalar math).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rt_amd.h"

typedef struct { float x, y, z; } v3;
typedef struct { v3 o, d; } ray_t;
typedef struct {
    float t;
    v3 p, n;
    float u, v;
    uint32_t tex, mat;
} hit_t;

typedef struct {
    const rt_scene_desc *sc;
    const float *table;
    const float *tex;
    int tw, th, layers;
    const float *cam;
    int w, h;
} world_t;

/* ---- builtins (same formulas as ref_shim.cpp) ------------------------------ */
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v_of(const rt_float3 *f) { return V(f->x, f->y, f->z); }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 scale(v3 a, float k) { return V(a.x * k, a.y * k, a.z * k); }
static inline v3 divs(v3 a, float k) { return V(a.x / k, a.y / k, a.z / k); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline v3 normalize(v3 a) { return divs(a, sqrtf(dot(a, a))); }
static inline v3 vmin(v3 a, v3 b) { return V(b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z); }
static inline float signf_(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : (x == 0.0f ? x : 0.0f)); }
static inline float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }

/* :127 — also rejects NaN (comparison false) */
static inline int in_range(float x) { return (x - RT_MAX_DISTANCE) * (x - RT_MIN_DISTANCE) <= 0.0f; }
/* :141 */
static inline v3 point_at(const ray_t *r, float t) { return add(r->o, scale(r->d, t)); }

/* ---- table index (:113-125): 64-bit sum, fp64 hash of the direction -------- */
static inline uint32_t dir_hash(v3 d) {
    float dp = dot(d, V(123.9898f, 348.233f, 433.3314f));
    return (uint32_t)fabs((double)dp * 438.5453);
}
static inline v3 random_vec(const world_t *wd, v3 dir, uint32_t s_seed, uint32_t gx, uint32_t gy) {
    uint64_t idx = ((uint64_t)dir_hash(dir) +
                    ((uint64_t)(uint32_t)(s_seed * 2683u) + (uint64_t)gx * 3931u + (uint64_t)gy * 2504u) * 3u) %
                   RT_RANDOM_BUFFER_SIZE;
    const float *t = wd->table + (uint32_t)idx; /* three consecutive FLOATS at offset idx (:109-111,117) */
    return V(t[0], t[1], t[2]);
}
static inline float random_u(const world_t *wd, v3 dir, uint32_t s_seed, uint32_t gx, uint32_t gy) {
    uint64_t idx = ((uint64_t)dir_hash(dir) + ((uint64_t)(uint32_t)(s_seed * 2683u) + (uint64_t)gx * 3931u + (uint64_t)gy)) %
                   RT_RANDOM_BUFFER_SIZE;
    return wd->table[3 * RT_RANDOM_BUFFER_SIZE + (uint32_t)idx];
}

/* ---- intersections ---------------------------------------------------------- */
/* :149-174 */
static int hit_sphere(const ray_t *r, const rt_sphere *s, hit_t *h) {
    v3 c = v_of(&s->pos);
    v3 oc = sub(c, r->o);
    float b = dot(oc, r->d);
    float cc = dot(oc, oc) - s->r * s->r;
    float dis = b * b - cc;
    if (dis > 0) {
        float d = sqrtf(dis);
        float t = b - d;
        if (!in_range(t)) {
            t = b + d;
            if (!in_range(t)) return 0;
        }
        h->t = t;
        h->p = point_at(r, t);
        h->n = divs(sub(h->p, c), s->r);
        h->mat = s->mat_ID;
        return 1;
    }
    return 0;
}

/* :176-194 */
static int hit_plane(const ray_t *r, const rt_plane *pl, hit_t *h) {
    v3 n = v_of(&pl->normal);
    float a = dot(r->d, n);
    float b = dot(sub(v_of(&pl->pos), r->o), n);
    float t = b / a;
    if (in_range(t)) {
        h->t = t;
        h->p = point_at(r, t);
        h->n = scale(neg(n), signf_(a));
        h->mat = pl->mat_ID;
        return 1;
    }
    return 0;
}

/* :196-255 — intersection of two spheres */
static int hit_lens(const ray_t *r, const rt_lens *l, hit_t *h) {
    v3 p1 = v_of(&l->p1), p2 = v_of(&l->p2);
    v3 oc = sub(p1, r->o);
    float b1 = dot(oc, r->d);
    float c = dot(oc, oc) - l->r1 * l->r1;
    float dis1 = b1 * b1 - c;
    oc = sub(p2, r->o);
    float b2 = dot(oc, r->d);
    c = dot(oc, oc) - l->r2 * l->r2;
    float dis2 = b2 * b2 - c;
    if (dis1 > 0 && dis2 > 0) {
        float d1 = sqrtf(dis1), d2 = sqrtf(dis2);
        float t1A = b1 - d1, t1B = b1 + d1, t2A = b2 - d2, t2B = b2 + d2;
        v3 centre;
        float rad, t;
        if ((t1B < t2A) || (t2B < t1A)) return 0;
        else if (RT_MIN_DISTANCE <= t1A || RT_MIN_DISTANCE <= t2A) { /* entering from outside */
            if (t2A <= t1A) { centre = p1; rad = l->r1; t = t1A; }
            else            { centre = p2; rad = l->r2; t = t2A; }
        } else if (RT_MIN_DISTANCE <= t1B && RT_MIN_DISTANCE <= t2B) { /* leaving from inside */
            if (t1B <= t2B) { centre = p1; rad = l->r1; t = t1B; }
            else            { centre = p2; rad = l->r2; t = t2B; }
        } else return 0;
        if (t <= RT_MAX_DISTANCE) {
            h->t = t;
            h->p = point_at(r, t);
            h->n = divs(sub(h->p, centre), rad);
            h->mat = l->mat_ID;
            return 1;
        }
    }
    return 0;
}

/* :93-103, :257-289 — Möller–Trumbore, no culling */
static int hit_triangle(const world_t *wd, const ray_t *r, const rt_mesh *m, uint32_t face, hit_t *h) {
    const rt_scene_desc *sc = wd->sc;
    const uint32_t *ib = sc->indices + m->index_anchor + 3u * face;
    size_t ia = (size_t)m->vertex_anchor + ib[0], ibx = (size_t)m->vertex_anchor + ib[1],
           ic = (size_t)m->vertex_anchor + ib[2];
    v3 A = v_of(sc->vertices + ia), B = v_of(sc->vertices + ibx), C = v_of(sc->vertices + ic);
    v3 e1 = sub(B, A), e2 = sub(C, A);
    v3 hv = cross(r->d, e2);
    float a = dot(e1, hv);
    if (a > -RT_TRIANGLE_EPSILON && a < RT_TRIANGLE_EPSILON) return 0;
    float f = 1.0f / a;
    v3 s = sub(r->o, A);
    float u = f * dot(s, hv);
    if (u < 0.0f || u > 1.0f) return 0;
    v3 q = cross(s, e1);
    float v = f * dot(r->d, q);
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = f * dot(e2, q);
    if (in_range(t)) {
        const rt_float2 *ua = sc->uvs + ia, *ub = sc->uvs + ibx, *uc = sc->uvs + ic;
        float wgt = 1.0f - u - v;
        h->u = (ua->x * wgt + ub->x * u) + uc->x * v;
        h->v = (ua->y * wgt + ub->y * u) + uc->y * v;
        h->t = t;
        h->p = point_at(r, t);
        h->n = normalize(cross(e1, e2));
        return 1;
    }
    return 0;
}

/* :291-303 — FIRST front-facing hit in face order, not the nearest */
static int hit_mesh(const world_t *wd, const ray_t *r, const rt_mesh *m, hit_t *h, rt_counters *cn) {
    for (uint32_t i = 0; i < m->face_count; i++) {
        cn->t_tri++;
        if (hit_triangle(wd, r, m, i, h)) {
            cn->h_tri++;
            if (dot(h->n, r->d) < 0.0f) {
                h->tex = m->texture_ID;
                return 1;
            }
        }
    }
    return 0;
}

/* :305-320 */
static int hit_model(const world_t *wd, const ray_t *r, const rt_model *mo, hit_t *h, rt_counters *cn) {
    int any = 0;
    float best = RT_MAX_DISTANCE;
    hit_t cand;
    for (uint32_t i = 0; i < mo->mesh_count; i++) {
        cn->t_mesh++;
        if (hit_mesh(wd, r, wd->sc->meshes + mo->mesh_anchor + i, &cand, cn) && cand.t < best) {
            any = 1;
            *h = cand;
            h->mat = mo->mat_ID;
            best = cand.t;
        }
    }
    return any;
}

/* :322-360 — nearest over spheres, planes, lenses, models; strict < keeps the
 * earlier primitive on ties.  uv/tex of a non-mesh hit are indeterminate in the
 * reference (never written); here they are 0. */
static int hit_scene(const world_t *wd, const ray_t *r, hit_t *h, rt_counters *cn) {
    const rt_scene_desc *sc = wd->sc;
    int any = 0;
    float best = RT_MAX_DISTANCE;
    hit_t cand;
    memset(&cand, 0, sizeof cand);
    cn->bounces++;
    cn->t_sphere += sc->sphere_count;
    cn->t_plane += sc->plane_count;
    cn->t_lens += sc->lens_count;
    cn->t_model += sc->model_count;
    for (uint32_t i = 0; i < sc->sphere_count; i++)
        if (hit_sphere(r, sc->spheres + i, &cand) && cand.t < best) { any = 1; *h = cand; best = cand.t; }
    for (uint32_t i = 0; i < sc->plane_count; i++)
        if (hit_plane(r, sc->planes + i, &cand) && cand.t < best) { any = 1; *h = cand; best = cand.t; }
    for (uint32_t i = 0; i < sc->lens_count; i++)
        if (hit_lens(r, sc->lenses + i, &cand) && cand.t < best) { any = 1; *h = cand; best = cand.t; }
    for (uint32_t i = 0; i < sc->model_count; i++)
        if (hit_model(wd, r, sc->models + i, &cand, cn) && cand.t < best) { any = 1; *h = cand; best = cand.t; }
    return any;
}

/* ---- materials -------------------------------------------------------------- */
/* :362-367 */
static void reflect(ray_t *r, v3 *c, const hit_t *h, const rt_material *m) {
    r->o = h->p;
    float k = 2.0f * dot(r->d, h->n);
    r->d = normalize(sub(r->d, scale(h->n, k)));
    if (m->type == RT_REFLECTIVE) *c = scale(*c, m->extra_data);
}

/* shared front of :369-381 and :407-418 */
static void facing(const ray_t *r, const hit_t *h, const rt_material *m, v3 *n, float *ratio, float *cai) {
    *cai = dot(r->d, h->n);
    if (*cai > 0) {
        *n = neg(h->n);
        *ratio = m->extra_data;
        *cai = -*cai;
    } else {
        *n = h->n;
        *ratio = 1.0f / m->extra_data;
    }
}

/* :382-386 / :424-429; the refracted direction is NOT renormalised */
static int try_refract(ray_t *r, const hit_t *h, v3 n, float ratio, float cai) {
    float disc = 1.0f - ratio * ratio * (1.0f - cai * cai);
    if (disc > 0.0f) {
        r->o = h->p;
        r->d = sub(scale(r->d, ratio), scale(n, ratio * cai + sqrtf(disc)));
        return 1;
    }
    return 0;
}

/* :369-391 */
static void refract(ray_t *r, v3 *c, hit_t *h, const rt_material *m) {
    v3 n;
    float ratio, cai;
    facing(r, h, m, &n, &ratio, &cai);
    if (!try_refract(r, h, n, ratio, cai)) {
        h->n = n;
        reflect(r, c, h, m);
    }
}

/* :393-399 */
static void scatter(const world_t *wd, ray_t *r, v3 *c, const hit_t *h, const rt_material *m, uint32_t seed,
                    uint32_t gx, uint32_t gy) {
    v3 rv = random_vec(wd, r->d, seed, gx, gy);
    r->d = normalize(add(h->n, rv));
    r->o = h->p;
    *c = scale(*c, m->extra_data);
}

/* :401-405 */
static float schlick(float cosine, float ratio) {
    float r0 = (1.0f - ratio) / (1.0f + ratio);
    r0 *= r0;
    return r0 + (1.0f - r0) * pow5(1.0f - cosine);
}

/* :407-435 */
static void dielectric(const world_t *wd, ray_t *r, v3 *c, hit_t *h, const rt_material *m, uint32_t seed, uint32_t gx,
                       uint32_t gy) {
    v3 n;
    float ratio, cai;
    facing(r, h, m, &n, &ratio, &cai);
    float prob = schlick(-cai, ratio);
    float rnd = random_u(wd, r->d, seed, gx, gy);
    if (prob < rnd && try_refract(r, h, n, ratio, cai)) return;
    h->n = n;
    reflect(r, c, h, m);
}

/* :105-107 + the bilinear definition of ref_shim.cpp */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static v3 texture_rgb(const world_t *wd, float s, float t, uint32_t tex_id) {
    if (!wd->tex || wd->layers <= 0) return V(0.0f, 0.0f, 0.0f);
    int W = wd->tw, H = wd->th;
    int layer = clampi((int)rintf((float)tex_id), 0, wd->layers - 1);
    float u = s * (float)W - 0.5f, v = t * (float)H - 0.5f;
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    int i0 = (fu >= -1.0f && fu <= 1.0e9f) ? (int)fu : 0;
    int j0 = (fv >= -1.0f && fv <= 1.0e9f) ? (int)fv : 0;
    int i1 = clampi(i0 + 1, 0, W - 1), j1 = clampi(j0 + 1, 0, H - 1);
    i0 = clampi(i0, 0, W - 1);
    j0 = clampi(j0, 0, H - 1);
    const float *base = wd->tex + (size_t)layer * W * H * 4;
    const float *t00 = base + 4 * ((size_t)j0 * W + i0), *t10 = base + 4 * ((size_t)j0 * W + i1);
    const float *t01 = base + 4 * ((size_t)j1 * W + i0), *t11 = base + 4 * ((size_t)j1 * W + i1);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return V(((w00 * t00[0] + w10 * t10[0]) + w01 * t01[0]) + w11 * t11[0],
             ((w00 * t00[1] + w10 * t10[1]) + w01 * t01[1]) + w11 * t11[1],
             ((w00 * t00[2] + w10 * t10[2]) + w01 * t01[2]) + w11 * t11[2]);
}

/* :444-486 — up to DEPTH bounces; colour mixing is min(), sky is black, a path
 * that survives 30 bounces returns what it has */
static v3 radiance(const world_t *wd, ray_t r, uint32_t sample, uint32_t gx, uint32_t gy, rt_counters *cn) {
    v3 out = V(1.0f, 1.0f, 1.0f);
    for (uint32_t i = 0; i < RT_DEPTH; i++) {
        hit_t h;
        if (!hit_scene(wd, &r, &h, cn)) return V(0.0f, 0.0f, 0.0f);
        cn->h_bounce++;
        const rt_material *m = wd->sc->materials + h.mat;
        v3 col = v_of(&m->color);
        switch (m->type) {
            case RT_DIFFUSE:
                cn->n_scatter++;
                scatter(wd, &r, &out, &h, m, i + sample, gx, gy);
                break;
            case RT_LIGHT:
                return vmin(out, col);
            case RT_REFLECTIVE:
                reflect(&r, &out, &h, m);
                break;
            case RT_REFRACTIVE:
                refract(&r, &out, &h, m);
                break;
            case RT_DIELECTRIC:
                cn->n_dielectric++;
                dielectric(wd, &r, &out, &h, m, i + sample, gx, gy);
                break;
            case RT_TEXTURED:
                cn->n_scatter++;
                cn->n_texfetch++;
                scatter(wd, &r, &out, &h, m, i + sample, gx, gy);
                col = texture_rgb(wd, h.u, h.v, h.tex);
                break;
            default:
                continue; /* unknown type: the reference's switch has no default */
        }
        out = vmin(out, col);
    }
    return out;
}

/* :129-139, :500-505 — no pixel jitter: every sample of a pixel starts identically */
static ray_t primary_ray(const world_t *wd, uint32_t x, uint32_t y) {
    const float *c = wd->cam;
    float s = (float)(int)x / (float)wd->w;
    float t = (float)(int)y / (float)wd->h;
    ray_t r;
    r.o = V(c[0], c[1], c[2]);
    v3 llc = V(c[3], c[4], c[5]), hor = V(c[6], c[7], c[8]), ver = V(c[9], c[10], c[11]);
    r.d = normalize(add(add(llc, scale(hor, s)), scale(ver, t)));
    return r;
}

static v3 sample_radiance(const world_t *wd, uint32_t x, uint32_t y, uint32_t sample, rt_counters *cn) {
    cn->samples++;
    return radiance(wd, primary_ray(wd, x, y), sample, x, y, cn);
}

static void add_counters(rt_counters *dst, const rt_counters *src) {
    uint64_t *d = (uint64_t *)dst;
    const uint64_t *s = (const uint64_t *)src;
    for (size_t i = 0; i < sizeof(rt_counters) / sizeof(uint64_t); i++) d[i] += s[i];
}

/* ---- frame drivers ---------------------------------------------------------- */
typedef struct {
    world_t wd;
    float *image;        /* full-frame RGBA32F, gamma space */
    int x0, y0, cw, ch;  /* region */
    uint32_t first, count;
    int mode;            /* 0: trace (sample `first`, overwrite); 1: retrace sample `first`;
                            2: progressive trace+retrace for samples 0..count-1 per pixel */
    int tid, nthreads;
    rt_counters cn;
} job_t;

/* :507-509 and :524-531 */
static void shade_pixel(job_t *j, uint32_t x, uint32_t y) {
    float *px = j->image + 4 * ((size_t)y * j->wd.w + x);
    uint32_t s0 = j->mode == 2 ? 0 : j->first, s1 = j->mode == 2 ? j->count : j->first + 1;
    for (uint32_t s = s0; s < s1; s++) {
        v3 c = sample_radiance(&j->wd, x, y, s, &j->cn);
        int blend = j->mode == 1 || (j->mode == 2 && s > 0);
        if (blend) {
            j->cn.image_reads++;
            v3 prev = V(px[0], px[1], px[2]);
            v3 lin = V(prev.x * prev.x, prev.y * prev.y, prev.z * prev.z);
            float k = (float)s / (float)(s + 1);
            c = V(c.x + (lin.x - c.x) * k, c.y + (lin.y - c.y) * k, c.z + (lin.z - c.z) * k);
        }
        px[0] = sqrtf(c.x);
        px[1] = sqrtf(c.y);
        px[2] = sqrtf(c.z);
        px[3] = 1.0f;
    }
}

static void *job_main(void *arg) {
    job_t *j = (job_t *)arg;
    for (int y = j->y0 + j->tid; y < j->y0 + j->ch; y += j->nthreads)
        for (int x = j->x0; x < j->x0 + j->cw; x++) shade_pixel(j, (uint32_t)x, (uint32_t)y);
    return NULL;
}

static int run_jobs(job_t *proto, int threads, rt_counters *out) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    job_t *jobs = (job_t *)calloc((size_t)threads, sizeof(job_t));
    pthread_t *tids = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    if (!jobs || !tids) return -1;
    for (int t = 0; t < threads; t++) {
        jobs[t] = *proto;
        jobs[t].tid = t;
        jobs[t].nthreads = threads;
        memset(&jobs[t].cn, 0, sizeof(rt_counters));
        if (threads == 1) job_main(&jobs[t]);
        else pthread_create(&tids[t], NULL, job_main, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) {
        if (threads > 1) pthread_join(tids[t], NULL);
        if (out) add_counters(out, &jobs[t].cn);
    }
    free(jobs);
    free(tids);
    return 0;
}

static world_t make_world(int w, int h, const float *cam, const float *table, const rt_scene_desc *sc,
                          const float *tex, int tw, int th, int layers) {
    world_t wd;
    wd.sc = sc; wd.table = table; wd.tex = tex; wd.tw = tw; wd.th = th; wd.layers = layers;
    wd.cam = cam; wd.w = w; wd.h = h;
    return wd;
}

/* region render.  mode 0 = one `trace` launch (sample 0), 1 = one `retrace`
 * launch of sample `first`, 2 = trace + (count-1) retraces.  Region outside
 * [x0,x0+cw)×[y0,y0+ch) is untouched.  counters (may be NULL) are ADDED to. */
int oracle_render(float *image_rgba, int w, int h, int x0, int y0, int cw, int ch, const float *cam,
                  const float *table, const rt_scene_desc *sc, const float *tex, int tw, int th, int layers,
                  int mode, uint32_t first, uint32_t count, int threads, rt_counters *counters) {
    if (!image_rgba || !cam || !table || !sc || w < 1 || h < 1 || x0 < 0 || y0 < 0 || x0 + cw > w || y0 + ch > h)
        return -1;
    job_t j;
    memset(&j, 0, sizeof j);
    j.wd = make_world(w, h, cam, table, sc, tex, tw, th, layers);
    j.image = image_rgba;
    j.x0 = x0; j.y0 = y0; j.cw = cw; j.ch = ch;
    j.first = first; j.count = count; j.mode = mode;
    return run_jobs(&j, threads, counters);
}

/* linear radiance of individual pixel-samples */
int oracle_samples(int w, int h, const float *cam, const float *table, const rt_scene_desc *sc, const float *tex,
                   int tw, int th, int layers, const uint32_t *xs, const uint32_t *ys, const uint32_t *samples,
                   size_t n, float *out_rgb, rt_counters *counters) {
    world_t wd = make_world(w, h, cam, table, sc, tex, tw, th, layers);
    rt_counters cn;
    memset(&cn, 0, sizeof cn);
    for (size_t i = 0; i < n; i++) {
        v3 c = sample_radiance(&wd, xs[i], ys[i], samples[i], &cn);
        out_rgb[3 * i] = c.x; out_rgb[3 * i + 1] = c.y; out_rgb[3 * i + 2] = c.z;
    }
    if (counters) add_counters(counters, &cn);
    return 0;
}

/* per-pixel SUM of linear radiance over samples first..first+count-1 in double
 * precision (RGB, 3 doubles per pixel of the region, row-major) — the exact
 * value the fused GPU accumulation approximates */
int oracle_linear_sum(double *sum_rgb, int w, int h, int x0, int y0, int cw, int ch, const float *cam,
                      const float *table, const rt_scene_desc *sc, const float *tex, int tw, int th, int layers,
                      uint32_t first, uint32_t count) {
    world_t wd = make_world(w, h, cam, table, sc, tex, tw, th, layers);
    rt_counters cn;
    memset(&cn, 0, sizeof cn);
    for (int y = 0; y < ch; y++)
        for (int x = 0; x < cw; x++) {
            double r = 0, g = 0, b = 0;
            for (uint32_t s = first; s < first + count; s++) {
                v3 c = sample_radiance(&wd, (uint32_t)(x0 + x), (uint32_t)(y0 + y), s, &cn);
                r += c.x; g += c.y; b += c.z;
            }
            double *o = sum_rgb + 3 * ((size_t)y * cw + x);
            o[0] = r; o[1] = g; o[2] = b;
        }
    return 0;
}

/* unit probes — same record layout as ref_hit() in ref_shim.cpp */
static void put_hit(float *o, int hit, const hit_t *h) {
    memset(o, 0, 12 * sizeof(float));
    if (!hit) return;
    o[0] = 1.0f; o[1] = h->t;
    o[2] = h->p.x; o[3] = h->p.y; o[4] = h->p.z;
    o[5] = h->n.x; o[6] = h->n.y; o[7] = h->n.z;
    o[8] = h->u; o[9] = h->v;
    memcpy(o + 10, &h->tex, 4);
    memcpy(o + 11, &h->mat, 4);
}

int oracle_hit(int kind, const rt_scene_desc *sc, const float *rays, const uint32_t *prim, size_t n, float *out) {
    world_t wd = make_world(1, 1, NULL, NULL, sc, NULL, 0, 0, 0);
    rt_counters cn;
    memset(&cn, 0, sizeof cn);
    for (size_t i = 0; i < n; i++) {
        ray_t r;
        r.o = V(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
        r.d = V(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
        hit_t h;
        memset(&h, 0, sizeof h);
        int hit;
        switch (kind) {
            case 0: hit = hit_sphere(&r, sc->spheres + prim[i], &h); break;
            case 1: hit = hit_plane(&r, sc->planes + prim[i], &h); break;
            case 2: hit = hit_lens(&r, sc->lenses + prim[i], &h); break;
            case 3: hit = hit_scene(&wd, &r, &h, &cn); break;
            default: return -1;
        }
        put_hit(out + 12 * i, hit, &h);
    }
    return 0;
}

int oracle_hit_triangle(const rt_scene_desc *sc, const float *rays, const uint32_t *mesh, const uint32_t *face,
                        size_t n, float *out) {
    world_t wd = make_world(1, 1, NULL, NULL, sc, NULL, 0, 0, 0);
    for (size_t i = 0; i < n; i++) {
        ray_t r;
        r.o = V(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
        r.d = V(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
        hit_t h;
        memset(&h, 0, sizeof h);
        int hit = hit_triangle(&wd, &r, sc->meshes + mesh[i], face[i], &h);
        put_hit(out + 12 * i, hit, &h);
    }
    return 0;
}

/* material routines on their own — same record layout as ref_material() in ref_shim.cpp:
 * in n × 16 floats {dir, p, normal, colour, mat_ID bits, s_seed bits, gid0 bits, gid1 bits} → out n × 9 floats
 * {new origin, new dir, colour}; routine 0 rayReflect :362, 1 rayRefract :369, 2 rayScatter :393,
 * 3 rayRefractDielectric :407 */
int oracle_material(int routine, const rt_scene_desc *sc, const float *table, const float *in, size_t n, float *out) {
    world_t wd = make_world(1, 1, NULL, table, sc, NULL, 0, 0, 0);
    for (size_t i = 0; i < n; i++) {
        const float *v = in + 16 * i;
        ray_t r;
        r.o = V(0.0f, 0.0f, 0.0f);
        r.d = V(v[0], v[1], v[2]);
        hit_t h;
        memset(&h, 0, sizeof h);
        h.p = V(v[3], v[4], v[5]);
        h.n = V(v[6], v[7], v[8]);
        v3 c = V(v[9], v[10], v[11]);
        uint32_t mat, seed, gx, gy;
        memcpy(&mat, v + 12, 4);
        memcpy(&seed, v + 13, 4);
        memcpy(&gx, v + 14, 4);
        memcpy(&gy, v + 15, 4);
        if (mat >= sc->material_count) return -1;
        h.mat = mat;
        const rt_material *m = sc->materials + mat;
        switch (routine) {
            case 0: reflect(&r, &c, &h, m); break;
            case 1: refract(&r, &c, &h, m); break;
            case 2: scatter(&wd, &r, &c, &h, m, seed, gx, gy); break;
            case 3: dielectric(&wd, &r, &c, &h, m, seed, gx, gy); break;
            default: return -1;
        }
        float *o = out + 9 * i;
        o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z;
        o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
        o[6] = c.x; o[7] = c.y; o[8] = c.z;
    }
    return 0;
}

uint64_t oracle_counters_bytes(const rt_counters *c) {
    return 32 * c->t_sphere + 48 * c->t_plane + 64 * c->t_lens + 12 * c->t_model + 16 * c->t_mesh + 60 * c->t_tri +
           36 * c->h_tri + 48 * c->h_bounce + 12 * c->n_scatter + 4 * c->n_dielectric + 64 * c->n_texfetch +
           (48 + 16) * c->samples + 16 * c->image_reads;
}

/* ---- deterministic random table (layout of src/raytracer.cpp:69-93) ---------
 * Philox-4x32-10, key = (seed lo, seed hi), counter = (entry, attempt, 0, 0).
 * Output words w0..w3 → x,y,z = 2·(w>>8)·2^-24 − 1 (rejection-sampled to the
 * open unit ball, attempt = 0,1,2,...), u = (w3>>8)·2^-24 of attempt 0. */
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t out[4]) {
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline float u24(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

int oracle_make_random_table(uint64_t seed, float *out, size_t n) {
    if (!out || n != RT_RANDOM_TABLE_FLOATS) return -1;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (uint32_t i = 0; i < RT_RANDOM_BUFFER_SIZE; i++) {
        uint32_t w[4];
        philox4x32_10(i, 0, 0, 0, k0, k1, w);
        out[3 * RT_RANDOM_BUFFER_SIZE + i] = u24(w[3]);
        for (uint32_t attempt = 0;; attempt++) {
            if (attempt) philox4x32_10(i, attempt, 0, 0, k0, k1, w);
            float x = 2.0f * u24(w[0]) - 1.0f, y = 2.0f * u24(w[1]) - 1.0f, z = 2.0f * u24(w[2]) - 1.0f;
            if ((x * x + y * y) + z * z < 1.0f) {
                out[3 * i] = x; out[3 * i + 1] = y; out[3 * i + 2] = z;
                break;
            }
        }
    }
    return 0;
}
