// rt_amd.hip — host side of librt_amd.so (include/rt_amd.h), gfx950 only: the C ABI, scene upload, the builders of
// the acceleration structures and the kernels that carry none of the reference's arithmetic (resolve, pack, unpack).
// The path-tracing kernels live in pt_kernels.hip, compiled once per arithmetic policy (pt_arith.hpp); this file
// picks a policy's launchers at run time (RT_OPT_ARITH).
//
// Build: __graft_entry__.build_hip().  No CPU fallback exists: without a gfx950 device rt_create() fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "rt_context.hpp"
#include "pt_kernels.hpp"
#include "mesh_bvh_build.hpp"

using namespace pt;
using namespace rtamd;

// =============================== policy-free kernels ===============================

// Multi-GPU exchange: the accumulator pixels a rank owns, packed in slot order (what
// travels over xGMI is 1/world of the frame instead of the whole frame) ...
__global__ __launch_bounds__(256) void pt_pack(FrameParams fp, const float4 *__restrict__ accum,
                                               float4 *__restrict__ packed, uint32_t n_slots) {
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= n_slots) return;
    uint32_t x, y;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (slot < fp.slot_end && slot_to_pixel(fp, slot, x, y)) v = accum[(size_t)y * fp.w + x];
    packed[slot] = v;
}
// ... and the inverse on the receiving rank: fp describes the SENDER's shard
__global__ __launch_bounds__(256) void pt_unpack(FrameParams fp, const float4 *__restrict__ packed,
                                                 float4 *__restrict__ accum) {
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    uint32_t x, y;
    if (slot < fp.slot_end && slot_to_pixel(fp, slot, x, y)) accum[(size_t)y * fp.w + x] = packed[slot];
}

// image = sqrt(accum.rgb / accum.w), alpha 1; pixels this rank does not own stay 0
__global__ __launch_bounds__(256) void pt_resolve(const float4 *__restrict__ accum, float4 *__restrict__ image,
                                                  uint32_t n, int linear_only) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 a = accum[i];
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (a.w > 0.0f) {
        float rx = a.x / a.w, ry = a.y / a.w, rz = a.z / a.w;
        o = linear_only ? make_float4(rx, ry, rz, 1.0f) : make_float4(sqrtf(rx), sqrtf(ry), sqrtf(rz), 1.0f);
    }
    image[i] = o;
}

// ================================== host side ==================================

namespace {

std::mutex g_err_mutex;
std::string g_create_error;

// Philox-4x32-10 random table — see rt_set_seed in rt_amd.h, DESIGN.md "random table"
void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
inline float u24(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

void make_table(uint64_t seed, float *out) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (uint32_t i = 0; i < RT_RANDOM_BUFFER_SIZE; i++) {
        uint32_t w[4];
        philox4x32_10(i, 0, 0, 0, k0, k1, w);
        out[3 * RT_RANDOM_BUFFER_SIZE + i] = u24(w[3]);
        for (uint32_t attempt = 0;; attempt++) {
            if (attempt) philox4x32_10(i, attempt, 0, 0, k0, k1, w);
            // volatile: keep each product/sum a separately rounded binary32 operation on the host
            volatile float x = 2.0f * u24(w[0]) - 1.0f, y = 2.0f * u24(w[1]) - 1.0f, z = 2.0f * u24(w[2]) - 1.0f;
            volatile float xx = x * x, yy = y * y, zz = z * z;
            volatile float s2 = xx + yy;
            volatile float s3 = s2 + zz;
            if (s3 < 1.0f) {
                out[3 * i] = x;
                out[3 * i + 1] = y;
                out[3 * i + 2] = z;
                break;
            }
        }
    }
}

// Sphere BVH (see hit_spheres_bvh): binned surface-area-heuristic splits (16 bins per axis over the
// centroids; median split when no bin boundary separates them), leaves of <= 4 spheres; children
// are allocated in adjacent pairs (left at an even index, lower coordinates along the split axis),
// every node records its split axis and, per direction octant, the node that follows its subtree in the
// near-before-far order of that octant (the walk is threaded: no stack, no way back up).
#ifndef SPHERE_BVH_LEAF
#define SPHERE_BVH_LEAF 4  // spheres per leaf (<= 7)
#endif
#define BVH_END 0x0FFFFFFFu
#define BVH_ROOT 1u
struct BvhBuild {
    const rt_sphere *sph;
    std::vector<uint32_t> order;
    std::vector<float4> nodes;   // 4 per node: (lo, A) (hi, B) and the eight skip links (pt_device.hpp hit_spheres_bvh)
    std::vector<float4> leaf_sph;
    std::vector<uint32_t> leaf_idx;

    void bounds(uint32_t b, uint32_t e, float lo[3], float hi[3]) const {
        for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
        for (uint32_t i = b; i < e; i++) {
            const rt_sphere &s = sph[order[i]];
            const float c[3] = {s.pos.x, s.pos.y, s.pos.z};
            float r = std::fabs(s.r);
            for (int k = 0; k < 3; k++) {
                lo[k] = std::fmin(lo[k], c[k] - r);
                hi[k] = std::fmax(hi[k], c[k] + r);
            }
        }
    }
    // skip[o]: the node that follows this subtree for a ray of direction octant o (bit k of o: d_k < 0), whose walk
    // enters the nearer child of every inner node first; BVH_END after the last one
    void fill(uint32_t me, const uint32_t skip[8], uint32_t b, uint32_t e, uint32_t depth = 0) {
        float lo[3], hi[3];
        bounds(b, e, lo, hi);
        uint32_t A = 0, B;
        if (e - b <= SPHERE_BVH_LEAF) {
            uint32_t first = (uint32_t)leaf_sph.size();
            for (uint32_t i = b; i < e; i++) {
                const rt_sphere &s = sph[order[i]];
                volatile float r2 = s.r * s.r;
                leaf_sph.push_back(make_float4(s.pos.x, s.pos.y, s.pos.z, r2));
                leaf_idx.push_back(order[i]);
            }
            B = 0x80000000u | ((e - b) << 28) | first;
        } else {
            float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = b; i < e; i++) {
                const rt_sphere &s = sph[order[i]];
                const float c[3] = {s.pos.x, s.pos.y, s.pos.z};
                for (int k = 0; k < 3; k++) { clo[k] = std::fmin(clo[k], c[k]); chi[k] = std::fmax(chi[k], c[k]); }
            }
            int ax = 0;
            for (int k = 1; k < 3; k++) if (chi[k] - clo[k] > chi[ax] - clo[ax]) ax = k;
            uint32_t mid = b + (e - b) / 2;
            const rt_sphere *sp = sph;
            // binned SAH: cost(split) = area(L)·|L| + area(R)·|R| over 15 boundaries × 3 axes
            constexpr int NB = 16;
            double best_cost = INFINITY;
            int best_ax = -1, best_bin = -1;
            for (int k = 0; k < 3 && depth < 32u; k++) {   // (median splits from level 32 on bound the depth)
                float ext = chi[k] - clo[k];
                if (!(ext > 0.0f)) continue;
                struct Bin { float lo[3], hi[3]; uint32_t n; } bins[NB];
                for (auto &bn : bins) { for (int q = 0; q < 3; q++) { bn.lo[q] = INFINITY; bn.hi[q] = -INFINITY; } bn.n = 0; }
                for (uint32_t i = b; i < e; i++) {
                    const rt_sphere &s = sph[order[i]];
                    const float c[3] = {s.pos.x, s.pos.y, s.pos.z};
                    int bi = (int)((c[k] - clo[k]) / ext * NB);
                    bi = bi < 0 ? 0 : (bi >= NB ? NB - 1 : bi);
                    float r = std::fabs(s.r);
                    for (int q = 0; q < 3; q++) { bins[bi].lo[q] = std::fmin(bins[bi].lo[q], c[q] - r); bins[bi].hi[q] = std::fmax(bins[bi].hi[q], c[q] + r); }
                    bins[bi].n++;
                }
                auto area = [](const float *l, const float *h) {
                    double dx = (double)h[0] - l[0], dy = (double)h[1] - l[1], dz = (double)h[2] - l[2];
                    return dx < 0 ? 0.0 : 2.0 * (dx * dy + dy * dz + dz * dx);
                };
                double right_cost[NB];
                float rl[3] = {INFINITY, INFINITY, INFINITY}, rh[3] = {-INFINITY, -INFINITY, -INFINITY};
                uint32_t rn = 0;
                for (int j = NB - 1; j > 0; j--) {
                    for (int q = 0; q < 3; q++) { rl[q] = std::fmin(rl[q], bins[j].lo[q]); rh[q] = std::fmax(rh[q], bins[j].hi[q]); }
                    rn += bins[j].n;
                    right_cost[j] = rn ? area(rl, rh) * rn : INFINITY;
                }
                float ll[3] = {INFINITY, INFINITY, INFINITY}, lh[3] = {-INFINITY, -INFINITY, -INFINITY};
                uint32_t ln = 0;
                for (int j = 0; j < NB - 1; j++) {
                    for (int q = 0; q < 3; q++) { ll[q] = std::fmin(ll[q], bins[j].lo[q]); lh[q] = std::fmax(lh[q], bins[j].hi[q]); }
                    ln += bins[j].n;
                    if (ln == 0 || ln == e - b) continue;
                    double cost = area(ll, lh) * ln + right_cost[j + 1];
                    if (cost < best_cost) { best_cost = cost; best_ax = k; best_bin = j; }
                }
            }
            if (best_ax >= 0) {
                ax = best_ax;
                float ext = chi[ax] - clo[ax], base = clo[ax];
                int bb = best_bin;
                auto it = std::partition(order.begin() + b, order.begin() + e, [sp, ax, ext, base, bb](uint32_t i) {
                    const float *ci = &sp[i].pos.x;
                    int bi = (int)((ci[ax] - base) / ext * NB);
                    bi = bi < 0 ? 0 : (bi >= NB ? NB - 1 : bi);
                    return bi <= bb;
                });
                mid = (uint32_t)(it - order.begin());
            }
            if (best_ax < 0 || mid == b || mid == e) {  // coincident centroids: median split by index
                mid = b + (e - b) / 2;
                std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e, [sp, ax](uint32_t i, uint32_t j) {
                    const float *ci = &sp[i].pos.x, *cj = &sp[j].pos.x;
                    return ci[ax] < cj[ax] || (ci[ax] == cj[ax] && i < j);
                });
            }
            uint32_t left = (uint32_t)(nodes.size() / 4);  // even: the root is node 1 and pairs follow, each in one 64-byte stretch
            nodes.resize(nodes.size() + 8);
            A = (uint32_t)ax << 28;
            B = left;
            uint32_t skip_l[8], skip_r[8];
            for (uint32_t o = 0; o < 8; o++) {   // the walk enters child B + ((o >> ax) & 1) first, its sibling next
                bool right_first = ((o >> ax) & 1u) != 0;
                skip_l[o] = right_first ? skip[o] : left + 1u;
                skip_r[o] = right_first ? left : skip[o];
            }
            fill(left, skip_l, b, mid, depth + 1u);
            fill(left + 1, skip_r, mid, e, depth + 1u);
        }
        float bc[3], bh[3];
        bvh_centre_half(lo, hi, bc, bh);
        nodes[4 * (size_t)me] = make_float4(bc[0], bc[1], bc[2], 0.0f);
        nodes[4 * (size_t)me + 1] = make_float4(bh[0], bh[1], bh[2], 0.0f);
        memcpy(&nodes[4 * (size_t)me].w, &A, 4);
        memcpy(&nodes[4 * (size_t)me + 1].w, &B, 4);
        memcpy(&nodes[4 * (size_t)me + 2], skip, 32);
    }
    void build(uint32_t b, uint32_t e) {
        nodes.assign(8, make_float4(0.0f, 0.0f, 0.0f, 0.0f));   // node 0 is padding, the root is node BVH_ROOT
        const uint32_t end[8] = {BVH_END, BVH_END, BVH_END, BVH_END, BVH_END, BVH_END, BVH_END, BVH_END};
        fill(BVH_ROOT, end, b, e);
    }
};

// per-face records (see DeviceScene::faces): the same binary32 operations hitTriangle performs.
// base[m] = first record of mesh m; fr = 3 float4 per face + one dummy record.
bool build_face_records(const rt_scene_desc *d, std::vector<uint32_t> &base, std::vector<float4> &fr) {
    base.assign(d->mesh_count ? d->mesh_count : 1, 0u);
    size_t total = 0;
    for (uint32_t m = 0; m < d->mesh_count; m++) { base[m] = (uint32_t)total; total += d->meshes[m].face_count; }
    if (total >= (1ull << 31)) return false;
    fr.assign(3 * (total + 1), make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    for (uint32_t m = 0; m < d->mesh_count; m++) {
        const rt_mesh &me = d->meshes[m];
        for (uint32_t f = 0; f < me.face_count; f++) {
            const uint32_t *ib = d->indices + me.index_anchor + 3u * f;
            const rt_float3 &A = d->vertices[me.vertex_anchor + ib[0]], &B = d->vertices[me.vertex_anchor + ib[1]],
                            &C = d->vertices[me.vertex_anchor + ib[2]];
            volatile float e1x = B.x - A.x, e1y = B.y - A.y, e1z = B.z - A.z;
            volatile float e2x = C.x - A.x, e2y = C.y - A.y, e2z = C.z - A.z;
            volatile float p1 = e1y * e2z, p2 = e1z * e2y, p3 = e1z * e2x, p4 = e1x * e2z, p5 = e1x * e2y, p6 = e1y * e2x;
            volatile float cx = p1 - p2, cy = p3 - p4, cz = p5 - p6;          // cross(e1, e2)
            volatile float xx = cx * cx, yy = cy * cy, zz = cz * cz;
            volatile float s2 = xx + yy;
            volatile float s3 = s2 + zz;                                        // dot = (x*x + y*y) + z*z
            volatile float len = std::sqrt((float)s3);
            volatile float nx = cx / len, ny = cy / len, nz = cz / len;          // normalize
            float4 *q = &fr[3 * ((size_t)base[m] + f)];
            q[0] = make_float4(A.x, A.y, A.z, e1x);
            q[1] = make_float4(e1y, e1z, e2x, e2y);
            q[2] = make_float4(e2z, nx, ny, nz);
        }
    }
    return true;
}

}  // namespace

namespace rtamd {
int fail(rt_context *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf;
    else {
        std::lock_guard<std::mutex> lk(g_err_mutex);
        g_create_error = buf;
    }
    return code;
}
}  // namespace rtamd

namespace {

int alloc_frame(rt_context *ctx, int w, int h) {
    if (ctx->d_image) (void)hipFree(ctx->d_image);
    if (ctx->d_accum) (void)hipFree(ctx->d_accum);
    ctx->d_image = ctx->d_accum = nullptr;
    if (ctx->d_recs) (void)hipFree(ctx->d_recs);
    if (ctx->d_live) (void)hipFree(ctx->d_live);
    if (ctx->d_trees) (void)hipFree(ctx->d_trees);
    if (ctx->d_glass) (void)hipFree(ctx->d_glass);
    if (ctx->d_tree_work) (void)hipFree(ctx->d_tree_work);
    ctx->d_tree_work = nullptr;
    ctx->d_recs = nullptr;
    ctx->d_live = nullptr;
    ctx->d_trees = nullptr;
    ctx->d_glass = nullptr;
    ctx->slot_capacity = 0;
    ctx->tree_capacity = 0;
    size_t bytes = (size_t)w * h * sizeof(float4);
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_image, bytes));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_accum, bytes));
    HIP_TRY(ctx, hipMemset(ctx->d_image, 0, bytes));
    HIP_TRY(ctx, hipMemset(ctx->d_accum, 0, bytes));
    ctx->width = w;
    ctx->height = h;
    ctx->accum_count = 0;
    ctx->sample_counter = 0;
    return RT_OK;
}

int check_ready(rt_context *ctx, const float *cam) {
    if (!ctx) return RT_EINVAL;
    if (!cam) return fail(ctx, RT_EINVAL, "camera block is NULL");
    if (!ctx->have_scene) return fail(ctx, RT_ESTATE, "no scene: call rt_set_scene first");
    if (ctx->scene_uses_textures && (ctx->tex_layers <= 0 || ctx->max_texture_id >= (uint32_t)ctx->tex_layers))
        return fail(ctx, RT_ERANGE, "scene has textured meshes (max texture id %u) but only %d texture layers are set",
                    ctx->max_texture_id, ctx->tex_layers);
    return RT_OK;
}

}  // namespace

namespace {

// Everything on the device that depends on the arithmetic policy (rt_set_scene, rt_set_option(RT_OPT_ARITH)):
//   * the spheres' test records (centre, w) of the brute-force loop (sph4: whole batches + one dummy batch that the
//     prefetch of the last iteration reads) and of the BVH leaves (bvh_sph): w = r*r, the one rounded binary32 product
//     hitSphere forms (:152), for policies 0 and 1; w = r for policy 2, where the product is contracted into
//     fma(-r, r, dot(oc, oc)) and never exists on its own.  Dummies: w = -inf (discriminant -inf) resp. NaN;
//   * the unit normal of every face record (A, e1, e2, n): host-computed IEEE operations for policy 0
//     (build_face_records), the policy's own cross / rsq-based normalize ON THE DEVICE for policies 1 and 2.
int apply_arith(rt_context *ctx) {
    const bool w_is_radius = ctx->arith == RT_ARITH_ROCM_OCL;
    auto rec = [&](const rt_sphere &s) {
        volatile float r2 = s.r * s.r;
        return make_float4(s.pos.x, s.pos.y, s.pos.z, w_is_radius ? s.r : (float)r2);
    };
    const uint32_t n = (uint32_t)ctx->h_spheres.size();
    const uint32_t batches = (n + PT_SPHERE_BATCH - 1) / PT_SPHERE_BATCH;
    std::vector<float4> v((size_t)(batches + 1) * PT_SPHERE_BATCH, make_float4(0.0f, 0.0f, 0.0f, w_is_radius ? NAN : -INFINITY));
    for (uint32_t i = 0; i < n; i++) v[i] = rec(ctx->h_spheres[i]);
    HIP_TRY(ctx, ctx->sph4.upload(v.data(), v.size()));
    ctx->sphere_batches = batches;
    if (!ctx->h_bvh_idx.empty()) {
        std::vector<float4> leaf(ctx->h_bvh_idx.size());
        for (size_t i = 0; i < leaf.size(); i++) leaf[i] = rec(ctx->h_spheres[ctx->h_bvh_idx[i]]);
        HIP_TRY(ctx, ctx->bvh_sph.upload(leaf.data(), leaf.size()));
    }
    HIP_TRY(ctx, ctx->faces.upload(ctx->h_faces.data(), ctx->h_faces.size()));
    HIP_TRY(ctx, ctx->mbvh_faces.upload(ctx->h_mbvh_faces.data(), ctx->h_mbvh_faces.size()));
    if (ctx->arith != RT_ARITH_IEEE) {
        int rc = ctx->ks->launch_face_normals(ctx, ctx->faces.p, (uint32_t)(ctx->h_faces.size() / 3));
        if (rc == RT_OK) rc = ctx->ks->launch_face_normals(ctx, ctx->mbvh_faces.p, (uint32_t)(ctx->h_mbvh_faces.size() / 3));
        if (rc != RT_OK) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RT_OK;
}

}  // namespace

// ================================== C ABI =====================================

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

const char *rt_last_error(const rt_context *ctx) {
    if (ctx) return ctx->error.c_str();
    std::lock_guard<std::mutex> lk(g_err_mutex);
    return g_create_error.c_str();
}

int rt_make_random_table(uint64_t seed, float *out, size_t n) {
    if (!out || n != RT_RANDOM_TABLE_FLOATS) return fail(nullptr, RT_EINVAL, "table must hold %d floats", RT_RANDOM_TABLE_FLOATS);
    make_table(seed, out);
    return RT_OK;
}

int rt_create(int device, int width, int height, rt_context **out) {
    if (!out) return fail(nullptr, RT_EINVAL, "out is NULL");
    *out = nullptr;
    if (width < 1 || height < 1 || width > RT_MAX_DIM || height > RT_MAX_DIM)
        return fail(nullptr, RT_EINVAL, "frame size %dx%d outside 1..%d", width, height, RT_MAX_DIM);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
        return fail(nullptr, RT_ENODEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(nullptr, RT_ENODEVICE, "device %d out of range (%d visible)", device, n);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, RT_EHIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, RT_ENODEVICE, "device %d is %s; librt_amd.so carries gfx950 code only", device,
                    prop.gcnArchName);
    rt_context *ctx = new (std::nothrow) rt_context();
    if (!ctx) return fail(nullptr, RT_EHIP, "out of host memory");
    ctx->device = device;
    ctx->dev_name = prop.name;
    ctx->dev_arch = prop.gcnArchName;
    ctx->cu_count = prop.multiProcessorCount;
    int rc = RT_OK;
    auto bail = [&](int code) {
        {
            std::lock_guard<std::mutex> lk(g_err_mutex);
            g_create_error = ctx->error;
        }
        rt_destroy(ctx);
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) { ctx->error = "hipSetDevice failed"; return bail(RT_EHIP); }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { ctx->error = "hipStreamCreate failed"; return bail(RT_EHIP); }
    ctx->stream = ctx->own_stream;
    for (int i = 0; i < rt_context::EV_RING; i++)
        if (hipEventCreate(&ctx->ev[i][0]) != hipSuccess || hipEventCreate(&ctx->ev[i][1]) != hipSuccess || hipEventCreate(&ctx->ev[i][2]) != hipSuccess) { ctx->error = "hipEventCreate failed"; return bail(RT_EHIP); }
    if (hipMalloc((void **)&ctx->d_counters, (COUNTER_REPLICAS * COUNTER_STRIDE + 32) * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(ctx->d_counters, 0, (COUNTER_REPLICAS * COUNTER_STRIDE + 32) * sizeof(unsigned long long)) != hipSuccess) { ctx->error = "counter allocation failed"; return bail(RT_EHIP); }
    ctx->d_walk_overflow = reinterpret_cast<uint32_t *>(ctx->d_counters + COUNTER_REPLICAS * COUNTER_STRIDE + 24);

    ctx->arith = RT_ARITH_IEEE;
    ctx->ks = kernel_set_a0();
    if ((rc = alloc_frame(ctx, width, height)) != RT_OK) return bail(rc);
    if ((rc = rt_set_seed(ctx, 0xC0FFEEull)) != RT_OK) return bail(rc);
    if ((rc = rt_set_textures(ctx, nullptr, 0, 0, 0)) != RT_OK) return bail(rc);
    *out = ctx;
    return RT_OK;
}

void rt_destroy(rt_context *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    ctx->materials.release(); ctx->spheres.release(); ctx->planes.release(); ctx->lenses.release();
    ctx->vertices.release(); ctx->uvs.release(); ctx->indices.release(); ctx->meshes.release(); ctx->models.release();
    ctx->table.release(); ctx->tex.release();
    if (ctx->d_image) (void)hipFree(ctx->d_image);
    if (ctx->d_accum) (void)hipFree(ctx->d_accum);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_recs) (void)hipFree(ctx->d_recs);
    if (ctx->d_live) (void)hipFree(ctx->d_live);
    if (ctx->d_trees) (void)hipFree(ctx->d_trees);
    if (ctx->d_glass) (void)hipFree(ctx->d_glass);
    if (ctx->d_tree_work) (void)hipFree(ctx->d_tree_work);
    ctx->sph4.release();
    ctx->faces.release();
    ctx->mesh_face_base.release();
    ctx->mbvh_nodes.release();
    ctx->mbvh_faces.release();
    ctx->mbvh_face_idx.release();
    ctx->walk_jobs.release();
    ctx->mesh_bvh_root.release();
    ctx->bvh_nodes.release();
    ctx->bvh_sph.release();
    ctx->bvh_idx.release();
    ctx->bvh_links.release();
    for (int i = 0; i < rt_context::EV_RING; i++)
        for (int k = 0; k < 3; k++)
            if (ctx->ev[i][k]) (void)hipEventDestroy(ctx->ev[i][k]);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int rt_resize(rt_context *ctx, int width, int height) {
    if (!ctx) return RT_EINVAL;
    if (width < 1 || height < 1 || width > RT_MAX_DIM || height > RT_MAX_DIM)
        return fail(ctx, RT_EINVAL, "frame size %dx%d outside 1..%d", width, height, RT_MAX_DIM);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return alloc_frame(ctx, width, height);
}

int rt_set_stream(rt_context *ctx, void *hip_stream) {
    if (!ctx) return RT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    ctx->ev_count = 0;
    return RT_OK;
}

int rt_set_scene(rt_context *ctx, const rt_scene_desc *d) {
    if (!ctx) return RT_EINVAL;
    if (!d) return fail(ctx, RT_EINVAL, "scene is NULL");
    ctx->walk_jobs.release();
    struct { const void *p; uint32_t n; const char *name; } arrs[] = {
        {d->materials, d->material_count, "materials"}, {d->spheres, d->sphere_count, "spheres"},
        {d->planes, d->plane_count, "planes"}, {d->lenses, d->lens_count, "lenses"},
        {d->vertices, d->vertex_count, "vertices"}, {d->uvs, d->uv_count, "uvs"},
        {d->indices, d->index_count, "indices"}, {d->meshes, d->mesh_count, "meshes"},
        {d->models, d->model_count, "models"}};
    for (auto &a : arrs)
        if (a.n && !a.p) return fail(ctx, RT_EINVAL, "%s: count %u but NULL pointer", a.name, a.n);
    if (d->uv_count != 0 && d->uv_count != d->vertex_count)
        return fail(ctx, RT_EINVAL, "uv_count (%u) must be 0 or vertex_count (%u)", d->uv_count, d->vertex_count);
    // validate every index the kernels dereference
    for (uint32_t i = 0; i < d->material_count; i++)
        if (d->materials[i].type < 0 || d->materials[i].type > RT_LIGHT)
            return fail(ctx, RT_EINVAL, "material %u has unknown type %d", i, d->materials[i].type);
    for (uint32_t i = 0; i < d->sphere_count; i++)
        if (d->spheres[i].mat_ID >= d->material_count) return fail(ctx, RT_ERANGE, "sphere %u: material %u does not exist", i, d->spheres[i].mat_ID);
    for (uint32_t i = 0; i < d->plane_count; i++)
        if (d->planes[i].mat_ID >= d->material_count) return fail(ctx, RT_ERANGE, "plane %u: material %u does not exist", i, d->planes[i].mat_ID);
    for (uint32_t i = 0; i < d->lens_count; i++)
        if (d->lenses[i].mat_ID >= d->material_count) return fail(ctx, RT_ERANGE, "lens %u: material %u does not exist", i, d->lenses[i].mat_ID);
    bool uses_tex = false;
    uint32_t max_tex = 0;
    for (uint32_t i = 0; i < d->model_count; i++) {
        const rt_model &mo = d->models[i];
        if (mo.mat_ID >= d->material_count) return fail(ctx, RT_ERANGE, "model %u: material %u does not exist", i, mo.mat_ID);
        if ((uint64_t)mo.mesh_anchor + mo.mesh_count > d->mesh_count) return fail(ctx, RT_ERANGE, "model %u: meshes [%u,+%u) outside the mesh array (%u)", i, mo.mesh_anchor, mo.mesh_count, d->mesh_count);
        bool textured = d->materials[mo.mat_ID].type == RT_TEXTURED;
        for (uint32_t k = 0; k < mo.mesh_count; k++) {
            const rt_mesh &me = d->meshes[mo.mesh_anchor + k];
            if (textured) {
                uses_tex = true;
                if (me.texture_ID > max_tex) max_tex = me.texture_ID;
            }
        }
    }
    for (uint32_t i = 0; i < d->mesh_count; i++) {
        const rt_mesh &me = d->meshes[i];
        if ((uint64_t)me.index_anchor + 3ull * me.face_count > d->index_count)
            return fail(ctx, RT_ERANGE, "mesh %u: indices [%u,+3*%u) outside the index array (%u)", i, me.index_anchor, me.face_count, d->index_count);
        for (uint64_t k = 0; k < 3ull * me.face_count; k++)
            if ((uint64_t)me.vertex_anchor + d->indices[me.index_anchor + k] >= d->vertex_count)
                return fail(ctx, RT_ERANGE, "mesh %u: vertex index %u (+anchor %u) outside the vertex array (%u)", i, d->indices[me.index_anchor + k], me.vertex_anchor, d->vertex_count);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_scene = false;
    HIP_TRY(ctx, ctx->materials.upload(d->materials, d->material_count));
    HIP_TRY(ctx, ctx->spheres.upload(d->spheres, d->sphere_count));
    ctx->h_spheres.assign(d->spheres, d->spheres + d->sphere_count);   // the test records follow in apply_arith()
    ctx->h_bvh_idx.clear();
    {
        std::vector<uint32_t> base;
        std::vector<float4> fr;
        if (!build_face_records(d, base, fr)) return fail(ctx, RT_EINVAL, "too many faces");
        HIP_TRY(ctx, ctx->mesh_face_base.upload(base.data(), d->mesh_count));
        // per-mesh BVHs for the "first front-facing hit in face order" rule (pt_mesh_bvh.hpp)
        std::vector<float4> nodes, lfaces;
        std::vector<uint32_t> lidx, roots(d->mesh_count ? d->mesh_count : 1, PT_MESH_BVH_NONE);
        ctx->have_mesh_bvh = false;
        for (uint32_t m = 0; m < d->mesh_count; m++) {
            if (d->meshes[m].face_count < MESH_BVH_MIN_FACES || d->meshes[m].face_count >= (1u << 27)) continue;
            MeshBvhBuilder mb;
            mb.rec = &fr[3 * (size_t)base[m]];
            mb.n_faces = d->meshes[m].face_count;
            mb.nodes = &nodes;
            mb.leaf_faces = &lfaces;
            mb.leaf_idx = &lidx;
            roots[m] = mb.build();
            ctx->have_mesh_bvh = true;
        }
        // the walks address these arrays with 32-bit byte offsets (at32): everything below 4 GiB, node indices below 2^26
        if (nodes.size() / 4 >= (1u << 26) || lidx.size() >= (1u << 26)) ctx->have_mesh_bvh = false;
        if (nodes.empty()) nodes.resize(4, make_float4(0, 0, 0, 0));
        if (lfaces.empty()) lfaces.resize(3, make_float4(0, 0, 0, 0));
        {   // device layout: 48 bytes per node (mesh_node_pack), wide-cone inner nodes spliced out of the links
            std::vector<uint8_t> is_root(nodes.size() / 4, 0);
            for (uint32_t m = 0; m < d->mesh_count; m++) if (roots[m] != PT_MESH_BVH_NONE && roots[m] < is_root.size()) is_root[roots[m]] = 1;
            mesh_collapse_links(nodes, is_root);
            std::vector<float4> packed(nodes.size() / 4 * 3);
            for (size_t n = 0; n < nodes.size() / 4; n++) mesh_node_pack(&nodes[4 * n], &packed[3 * n]);
            HIP_TRY(ctx, ctx->mbvh_nodes.upload(packed.data(), packed.size()));
        }
        ctx->h_faces.swap(fr);            // uploaded by apply_arith(): the normals depend on the arithmetic policy
        ctx->h_mbvh_faces.swap(lfaces);
        HIP_TRY(ctx, ctx->mbvh_face_idx.upload(lidx.data(), lidx.size()));
        HIP_TRY(ctx, ctx->mesh_bvh_root.upload(roots.data(), d->mesh_count));
        std::vector<uint2> jobs;
        bool all_bvh = ctx->have_mesh_bvh && d->model_count > 0;
        for (uint32_t mo = 0; mo < d->model_count && all_bvh; mo++)
            for (uint32_t k = 0; k < d->models[mo].mesh_count && all_bvh; k++) {
                uint32_t mi = d->models[mo].mesh_anchor + k;
                all_bvh = mi < d->mesh_count && roots[mi] != PT_MESH_BVH_NONE;
                jobs.push_back(make_uint2(mi, d->models[mo].mat_ID));
            }
        if (all_bvh && !jobs.empty() && jobs.size() < (1u << 16)) HIP_TRY(ctx, ctx->walk_jobs.upload(jobs.data(), jobs.size()));
    }
    ctx->bvh_node_count = 0;
    if (d->sphere_count > 0 && d->sphere_count < (1u << 24)) {   // (32-bit byte offsets into the node / link / leaf arrays: at32; a node's links are 128 bytes)
        bool finite = true;
        for (uint32_t i = 0; i < d->sphere_count && finite; i++)
            finite = std::isfinite(d->spheres[i].pos.x) && std::isfinite(d->spheres[i].pos.y) &&
                     std::isfinite(d->spheres[i].pos.z) && std::isfinite(d->spheres[i].r);
        if (finite) {  // non-finite spheres: no BVH, the brute-force loop handles them as the reference does
            BvhBuild bb;
            bb.sph = d->spheres;
            bb.order.resize(d->sphere_count);
            for (uint32_t i = 0; i < d->sphere_count; i++) bb.order[i] = i;
            bb.build(0, d->sphere_count);
            {   // device layout (hit_spheres_bvh): (centre, B) in 16 bytes; per octant (skip | axis << 28, half extent) in 16 bytes
                std::vector<float4> boxes(bb.nodes.size() / 4), links(bb.nodes.size() * 2);
                for (size_t n = 0; n < bb.nodes.size() / 4; n++) {
                    boxes[n] = bb.nodes[4 * n];
                    boxes[n].w = bb.nodes[4 * n + 1].w;
                    uint32_t A, s8[8];
                    memcpy(&A, &bb.nodes[4 * n].w, 4);
                    memcpy(s8, &bb.nodes[4 * n + 2], 32);
                    for (int o = 0; o < 8; o++) {
                        const uint32_t w = s8[o] | (A & 0x30000000u);
                        float4 &l = links[8 * n + o];
                        l = make_float4(0.0f, bb.nodes[4 * n + 1].x, bb.nodes[4 * n + 1].y, bb.nodes[4 * n + 1].z);
                        memcpy(&l.x, &w, 4);
                    }
                }
                HIP_TRY(ctx, ctx->bvh_nodes.upload(boxes.data(), boxes.size()));
                HIP_TRY(ctx, ctx->bvh_links.upload(links.data(), links.size()));
            }
            HIP_TRY(ctx, ctx->bvh_idx.upload(bb.leaf_idx.data(), bb.leaf_idx.size()));
            ctx->h_bvh_idx = bb.leaf_idx;   // the leaf records (centre, r*r or r) follow in apply_arith()
            // the walk addresses the link array with the 32-bit byte offset node << 7 and leaf slots with slot << 4
            // (at32): a degenerate set whose tree outgrows that takes the brute-force loop
            ctx->bvh_node_count = (bb.nodes.size() / 4 < (1u << 25) && bb.leaf_sph.size() < (1u << 28)) ? (uint32_t)(bb.nodes.size() / 4) : 0u;
            ctx->bvh_rmax = 0;
            for (int k = 0; k < 3; k++) { ctx->bvh_lo[k] = INFINITY; ctx->bvh_hi[k] = -INFINITY; }
            for (uint32_t i = 0; i < d->sphere_count; i++) {
                const float c[3] = {d->spheres[i].pos.x, d->spheres[i].pos.y, d->spheres[i].pos.z};
                for (int k = 0; k < 3; k++) { ctx->bvh_lo[k] = std::fmin(ctx->bvh_lo[k], c[k]); ctx->bvh_hi[k] = std::fmax(ctx->bvh_hi[k], c[k]); }
                ctx->bvh_rmax = std::fmax(ctx->bvh_rmax, std::fabs(d->spheres[i].r));
            }
        }
    }
    HIP_TRY(ctx, ctx->planes.upload(d->planes, d->plane_count));
    HIP_TRY(ctx, ctx->lenses.upload(d->lenses, d->lens_count));
    HIP_TRY(ctx, ctx->vertices.upload(d->vertices, d->vertex_count));
    if (d->uv_count) HIP_TRY(ctx, ctx->uvs.upload(d->uvs, d->uv_count));
    else {
        std::vector<rt_float2> zeros(d->vertex_count ? d->vertex_count : 1, rt_float2{0.0f, 0.0f});
        HIP_TRY(ctx, ctx->uvs.upload(zeros.data(), d->vertex_count));
    }
    HIP_TRY(ctx, ctx->indices.upload(d->indices, d->index_count));
    HIP_TRY(ctx, ctx->meshes.upload(d->meshes, d->mesh_count));
    HIP_TRY(ctx, ctx->models.upload(d->models, d->model_count));
    ctx->scene_uses_textures = uses_tex;
    ctx->max_texture_id = max_tex;
    {
        int rc = apply_arith(ctx);
        if (rc != RT_OK) return rc;
    }
    ctx->have_scene = true;
    ctx->sample_counter = 0;
    return RT_OK;
}

int rt_set_textures(rt_context *ctx, const float *rgba, int w, int h, int layers) {
    if (!ctx) return RT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (layers == 0) {
        float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        HIP_TRY(ctx, ctx->tex.upload(&zero, 1));
        ctx->tex_w = ctx->tex_h = 1;
        ctx->tex_layers = 0;
        return RT_OK;
    }
    if (!rgba || w < 1 || h < 1 || layers < 1 || w > 32768 || h > 32768 || layers > 2048)
        return fail(ctx, RT_EINVAL, "bad texture array %dx%dx%d", w, h, layers);
    HIP_TRY(ctx, ctx->tex.upload((const float4 *)rgba, (size_t)w * h * layers));
    ctx->tex_w = w;
    ctx->tex_h = h;
    ctx->tex_layers = layers;
    return RT_OK;
}

int rt_set_random_table(rt_context *ctx, const float *table, size_t n) {
    if (!ctx) return RT_EINVAL;
    if (!table || n != RT_RANDOM_TABLE_FLOATS) return fail(ctx, RT_EINVAL, "table must hold %d floats", RT_RANDOM_TABLE_FLOATS);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // 2 floats of slack: getVec reads idx..idx+2 with idx < 100000 — inside the table already
    HIP_TRY(ctx, ctx->table.upload(table, n));
    return RT_OK;
}

int rt_set_seed(rt_context *ctx, uint64_t seed) {
    if (!ctx) return RT_EINVAL;
    std::vector<float> t(RT_RANDOM_TABLE_FLOATS);
    make_table(seed, t.data());
    return rt_set_random_table(ctx, t.data(), t.size());
}

int rt_get_random_table(rt_context *ctx, float *out, size_t n) {
    if (!ctx) return RT_EINVAL;
    if (!out || n != RT_RANDOM_TABLE_FLOATS) return fail(ctx, RT_EINVAL, "table must hold %d floats", RT_RANDOM_TABLE_FLOATS);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->table.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_set_shard(rt_context *ctx, int rank, int world, int tile_w, int tile_h) {
    if (!ctx) return RT_EINVAL;
    auto log2_exact = [](int v) { int l = 0; while ((1 << l) < v) l++; return (1 << l) == v ? l : -1; };
    int lw = log2_exact(tile_w), lh = log2_exact(tile_h);
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, RT_EINVAL, "rank %d of %d", rank, world);
    if (tile_w < 1 || tile_h < 1 || lw < 0 || lh < 0 || lw + lh > 16)
        return fail(ctx, RT_EINVAL, "tile %dx%d: sides must be powers of two, area <= 65536", tile_w, tile_h);
    ctx->rank = rank;
    ctx->world = world;
    ctx->tile_w_log2 = lw;
    ctx->tile_h_log2 = lh;
    return RT_OK;
}

int rt_render(rt_context *ctx, const float camera[12]) {
    int rc = check_ready(ctx, camera);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->sample_counter = 0;  // src/raytracer.cpp:128
    if ((rc = ctx->ks->launch_render(ctx, MODE_TRACE, camera, 0, 1, 0)) != RT_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // queue.finish(), src/raytracer.cpp:140
    return RT_OK;
}

int rt_render_again(rt_context *ctx, const float camera[12]) {
    int rc = check_ready(ctx, camera);
    if (rc) return rc;
    if (ctx->sample_counter >= RT_MAX_SAMPLE) return fail(ctx, RT_EINVAL, "sample counter limit %u reached", RT_MAX_SAMPLE);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->sample_counter++;  // src/raytracer.cpp:147
    if ((rc = ctx->ks->launch_render(ctx, MODE_RETRACE, camera, ctx->sample_counter, 1, 0)) != RT_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_sample_counter(const rt_context *ctx, uint32_t *out) {
    if (!ctx || !out) return RT_EINVAL;
    *out = ctx->sample_counter;
    return RT_OK;
}

int rt_clear(rt_context *ctx) {
    if (!ctx) return RT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_accum, 0, (size_t)ctx->width * ctx->height * sizeof(float4), ctx->stream));
    ctx->accum_count = 0;
    return RT_OK;
}

int rt_render_spp(rt_context *ctx, const float camera[12], uint32_t first_sample, uint32_t n_samples) {
    int rc = check_ready(ctx, camera);
    if (rc) return rc;
    if (n_samples == 0) return RT_OK;
    if ((uint64_t)first_sample + n_samples - 1 > RT_MAX_SAMPLE)
        return fail(ctx, RT_EINVAL, "samples %u..+%u exceed the limit %u", first_sample, n_samples, RT_MAX_SAMPLE);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // A call of more than RT_SPP_PER_LAUNCH samples per pixel runs as consecutive launches of that many (the capacity of a
    // wave's sample queue: beyond it a frame would fall back to the fixed-lane kernel, 1.8 x slower — 520 samples took
    // 18.4 ms where 512 take 10.2); the accumulator then is the sum of those launches' sums, on every path alike.
    for (uint32_t done = 0; done < n_samples;) {
        const uint32_t c = n_samples - done < RT_SPP_PER_LAUNCH ? n_samples - done : RT_SPP_PER_LAUNCH;
        rc = ctx->prefix_sharing ? ctx->ks->launch_fused(ctx, camera, first_sample + done, c, group_log2_for(c))
                                 : ctx->ks->launch_render(ctx, MODE_ACCUM, camera, first_sample + done, c, group_log2_for(c));
        if (rc != RT_OK) return rc;
        done += c;
        ctx->accum_count += c;
    }
    return RT_OK;
}

static int resolve_into(rt_context *ctx, int linear_only) {
    uint32_t n = (uint32_t)ctx->width * ctx->height;
    hipLaunchKernelGGL(pt_resolve, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_accum, ctx->d_image, n,
                       linear_only);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

int rt_resolve(rt_context *ctx) {
    if (!ctx) return RT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return resolve_into(ctx, 0);
}

int rt_sync(rt_context *ctx) {
    if (!ctx) return RT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_trace_samples(rt_context *ctx, const float camera[12], const uint32_t *x, const uint32_t *y,
                     const uint32_t *sample, size_t n, float *out_rgb) {
    int rc = check_ready(ctx, camera);
    if (rc) return rc;
    if (n == 0) return RT_OK;
    if (!x || !y || !sample || !out_rgb || n > (1u << 28)) return fail(ctx, RT_EINVAL, "bad probe arrays");
    for (size_t i = 0; i < n; i++)
        if (x[i] >= (uint32_t)ctx->width || y[i] >= (uint32_t)ctx->height || sample[i] > RT_MAX_SAMPLE)
            return fail(ctx, RT_EINVAL, "probe %zu (%u,%u,%u) outside the frame / sample range", i, x[i], y[i], sample[i]);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t *d_in = nullptr;
    float *d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d_in, 3 * n * sizeof(uint32_t)));
    if (hipMalloc((void **)&d_out, 3 * n * sizeof(float)) != hipSuccess) { (void)hipFree(d_in); return fail(ctx, RT_EHIP, "hipMalloc failed"); }
    hipError_t e = hipMemcpyAsync(d_in, x, n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in + n, y, n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in + 2 * n, sample, n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        FrameParams fp = frame_params(ctx, camera, 0, 1, 0);
        DeviceScene sc = device_scene(ctx);
        if (ctx->ks->launch_probe(ctx, fp, sc, d_in, (uint32_t)n, d_out) != RT_OK) e = hipErrorLaunchFailure;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_rgb, d_out, 3 * n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ctx, RT_EHIP, "probe: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_read_image(rt_context *ctx, float *rgba, size_t bytes) {
    if (!ctx) return RT_EINVAL;
    size_t need = (size_t)ctx->width * ctx->height * sizeof(float4);
    if (!rgba || bytes != need) return fail(ctx, RT_EINVAL, "image buffer must be %zu bytes", need);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(rgba, ctx->d_image, need, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_read_linear(rt_context *ctx, float *rgba, size_t bytes) {
    if (!ctx) return RT_EINVAL;
    size_t need = (size_t)ctx->width * ctx->height * sizeof(float4);
    if (!rgba || bytes != need) return fail(ctx, RT_EINVAL, "image buffer must be %zu bytes", need);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float4 *tmp = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&tmp, need));
    uint32_t n = (uint32_t)ctx->width * ctx->height;
    hipLaunchKernelGGL(pt_resolve, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_accum, tmp, n, 1);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(rgba, tmp, need, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(ctx, RT_EHIP, "read_linear: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_device_image(rt_context *ctx, void **d_rgba) {
    if (!ctx || !d_rgba) return RT_EINVAL;
    *d_rgba = ctx->d_image;
    return RT_OK;
}

int rt_device_accum(rt_context *ctx, void **d_rgba) {
    if (!ctx || !d_rgba) return RT_EINVAL;
    *d_rgba = ctx->d_accum;
    return RT_OK;
}

int rt_enable_counters(rt_context *ctx, int enable) {
    if (!ctx) return RT_EINVAL;
    ctx->count_enabled = enable != 0;
    return RT_OK;
}

// ---- host-only self-check of the acceleration structures ------------------------------------
extern "C++" {
namespace {
struct BvhWalkStats { uint64_t nodes = 0, leaves = 0, prims = 0, max_depth = 0; };

// walk the sphere tree the way the kernel does for direction octant `oct` with every box "hit": nearer child
// after an inner node, the octant's skip link after a leaf
std::string host_walk_octant(const std::vector<float4> &nodes, uint32_t oct, std::vector<uint32_t> &leaf_visits, BvhWalkStats &st) {
    uint32_t n_nodes = (uint32_t)(nodes.size() / 4), cur = BVH_ROOT;
    for (uint64_t guard = 0; cur != BVH_END; guard++) {
        if (guard > (uint64_t)n_nodes + 8) return "walk does not terminate";
        if (cur >= n_nodes) return "node index out of range";
        uint32_t A, B, sk[8];
        memcpy(&A, &nodes[4 * (size_t)cur].w, 4);
        memcpy(&B, &nodes[4 * (size_t)cur + 1].w, 4);
        memcpy(sk, &nodes[4 * (size_t)cur + 2], 32);
        st.nodes++;
        if (B & 0x80000000u) {
            uint32_t first = B & 0x0FFFFFFFu, cnt = (B >> 28) & 7u;
            st.leaves++;
            for (uint32_t k = 0; k < cnt; k++) {
                if (first + k >= leaf_visits.size()) return "leaf slot out of range";
                leaf_visits[first + k]++;
            }
            cur = sk[oct];
        } else {
            if (((A >> 28) & 3u) > 2) return "bad split axis";
            if (B & 1u) return "left child at an odd index";
            cur = B + ((oct >> ((A >> 28) & 3u)) & 1u);
        }
    }
    return "";
}

// recursive structural check of the sphere tree: child boxes inside the parent's, the eight skip links, depth
std::string host_check_sphere_tree(const std::vector<float4> &nodes, uint32_t me, uint32_t parent, const uint32_t skip[8],
                                   bool is_root, uint32_t depth, BvhWalkStats &st) {
    const float4 &lo = nodes[4 * (size_t)me], &hi = nodes[4 * (size_t)me + 1];
    uint32_t A, B, sk[8];
    memcpy(&A, &lo.w, 4);
    memcpy(&B, &hi.w, 4);
    memcpy(sk, &nodes[4 * (size_t)me + 2], 32);
    for (int o = 0; o < 8; o++) if (sk[o] != skip[o]) return "wrong skip link";
    st.max_depth = std::max<uint64_t>(st.max_depth, depth);
    if (depth > 60) return "tree deeper than 60 levels";
    (void)parent; (void)is_root;   // (boxes: every primitive is checked against the box of EVERY node above it, see rt_debug_check_accel)
    if (B & 0x80000000u) return "";
    uint32_t ax = (A >> 28) & 3u, sl[8], sr[8];
    for (uint32_t o = 0; o < 8; o++) {
        bool right_first = ((o >> ax) & 1u) != 0;
        sl[o] = right_first ? skip[o] : B + 1u;
        sr[o] = right_first ? B : skip[o];
    }
    std::string e = host_check_sphere_tree(nodes, B, me, sl, false, depth + 1, st);
    if (e.empty()) e = host_check_sphere_tree(nodes, B + 1u, me, sr, false, depth + 1, st);
    return e;
}

// the mesh trees are threaded (pt_mesh_bvh.hpp): follow "left child after an inner node, skip link after a leaf"
std::string host_walk_threaded(const std::vector<float4> &nodes, uint32_t root, std::vector<uint32_t> &leaf_visits, BvhWalkStats &st) {
    uint32_t n_nodes = (uint32_t)(nodes.size() / 4), cur = root;
    for (uint64_t guard = 0; cur != 0x0FFFFFFFu; guard++) {
        if (guard > (uint64_t)n_nodes + 8) return "walk does not terminate";
        if (cur >= n_nodes) return "node index out of range";
        uint32_t A, B;
        memcpy(&A, &nodes[4 * (size_t)cur].w, 4);
        memcpy(&B, &nodes[4 * (size_t)cur + 1].w, 4);
        st.nodes++;
        if (B & 0x80000000u) {
            uint32_t first = B & 0x0FFFFFFFu, cnt = (B >> 28) & 7u;
            st.leaves++;
            for (uint32_t k = 0; k < cnt; k++) {
                if (first + k >= leaf_visits.size()) return "leaf slot out of range";
                leaf_visits[first + k]++;
            }
            cur = A & 0x0FFFFFFFu;
        } else {
            cur = B;
        }
    }
    return "";
}

// recursive structural check of a threaded tree: child boxes inside the parent's, skip links, depth
std::string host_check_threaded(const std::vector<float4> &nodes, uint32_t me, uint32_t parent, uint32_t skip, bool is_root,
                                uint32_t depth, BvhWalkStats &st) {
    const float4 &lo = nodes[4 * (size_t)me], &hi = nodes[4 * (size_t)me + 1];
    uint32_t A, B;
    memcpy(&A, &lo.w, 4);
    memcpy(&B, &hi.w, 4);
    if ((A & 0x0FFFFFFFu) != skip) return "wrong skip link";
    st.max_depth = std::max<uint64_t>(st.max_depth, depth);
    if (depth > 60) return "tree deeper than 60 levels";
    (void)parent; (void)is_root;   // (boxes: every primitive is checked against the box of EVERY node above it, see rt_debug_check_accel)
    if (B & 0x80000000u) return "";
    uint32_t right;   // = the left child's skip link
    memcpy(&right, &nodes[4 * (size_t)B].w, 4);
    right &= 0x0FFFFFFFu;
    if (right >= nodes.size() / 4) return "left child's skip link out of range";
    std::string e = host_check_threaded(nodes, B, me, right, false, depth + 1, st);
    if (e.empty()) e = host_check_threaded(nodes, right, me, skip, false, depth + 1, st);
    return e;
}

}  // namespace
}  // extern "C++"

int rt_debug_check_accel(const rt_scene_desc *d, uint64_t stats[8], char *err, size_t err_len) {
    auto bad = [&](const std::string &m) {
        if (err && err_len) snprintf(err, err_len, "%s", m.c_str());
        return RT_EINVAL;
    };
    if (!d || !stats) return RT_EINVAL;
    memset(stats, 0, 8 * sizeof(uint64_t));
    // ---- sphere BVH
    if (d->sphere_count) {
        BvhBuild bb;
        bb.sph = d->spheres;
        bb.order.resize(d->sphere_count);
        for (uint32_t i = 0; i < d->sphere_count; i++) bb.order[i] = i;
        bb.build(0, d->sphere_count);
        BvhWalkStats st;
        const uint32_t end[8] = {BVH_END, BVH_END, BVH_END, BVH_END, BVH_END, BVH_END, BVH_END, BVH_END};
        std::string e = host_check_sphere_tree(bb.nodes, BVH_ROOT, BVH_ROOT, end, true, 0, st);
        if (!e.empty()) return bad("sphere bvh: " + e);
        for (uint32_t oct = 0; oct < 8; oct++) {
            std::vector<uint32_t> visits(bb.leaf_idx.size(), 0);
            BvhWalkStats w;
            e = host_walk_octant(bb.nodes, oct, visits, w);
            if (!e.empty()) return bad("sphere bvh: " + e);
            for (uint32_t v : visits) if (v != 1) return bad("sphere bvh: a leaf slot is not visited exactly once");
            st.nodes = w.nodes; st.leaves = w.leaves;
        }
        std::vector<uint32_t> seen(d->sphere_count, 0);
        if (bb.leaf_idx.size() != d->sphere_count) return bad("sphere bvh: leaf slot count != sphere count");
        for (uint32_t i : bb.leaf_idx) { if (i >= d->sphere_count || seen[i]++) return bad("sphere bvh: sphere missing or duplicated"); }
        // every sphere inside the box (centre ± half extent) of its leaf and of every node above it
        std::function<std::string(uint32_t, std::vector<uint32_t> &)> gather = [&](uint32_t n, std::vector<uint32_t> &sph) -> std::string {
            uint32_t B;
            memcpy(&B, &bb.nodes[4 * (size_t)n + 1].w, 4);
            if (B & 0x80000000u) {
                uint32_t first = B & 0x0FFFFFFFu, cnt = (B >> 28) & 7u;
                for (uint32_t k = 0; k < cnt; k++) sph.push_back(bb.leaf_idx[first + k]);
            } else {
                for (uint32_t k = 0; k < 2; k++) {
                    std::vector<uint32_t> sub;
                    std::string er = gather(B + k, sub);
                    if (!er.empty()) return er;
                    sph.insert(sph.end(), sub.begin(), sub.end());
                }
            }
            const float4 &c = bb.nodes[4 * (size_t)n], &h = bb.nodes[4 * (size_t)n + 1];
            for (uint32_t i : sph) {
                const rt_sphere &sp = d->spheres[i];
                const float r = std::fabs(sp.r), p[3] = {sp.pos.x, sp.pos.y, sp.pos.z};
                for (int k = 0; k < 3; k++) {   // (the sphere's extent in binary32, as BvhBuild::bounds forms it)
                    const double lo = (double)(&c.x)[k] - (double)(&h.x)[k], hi = (double)(&c.x)[k] + (double)(&h.x)[k];
                    const float slo = p[k] - r, shi = p[k] + r;
                    if ((double)slo < lo || (double)shi > hi) return std::string("sphere outside the box of a node above it");
                }
            }
            return "";
        };
        {
            std::vector<uint32_t> all;
            e = gather(BVH_ROOT, all);
            if (!e.empty()) return bad("sphere bvh: " + e);
        }
        stats[0] = st.nodes; stats[1] = st.leaves; stats[2] = st.max_depth;
    }
    // ---- mesh BVHs
    std::vector<uint32_t> base;
    std::vector<float4> fr;
    if (!build_face_records(d, base, fr)) return bad("too many faces");
    for (uint32_t m = 0; m < d->mesh_count; m++) {
        uint32_t nf = d->meshes[m].face_count;
        if (nf < MESH_BVH_MIN_FACES) continue;
        std::vector<float4> nodes, lfaces;
        std::vector<uint32_t> lidx;
        MeshBvhBuilder mb;
        mb.rec = &fr[3 * (size_t)base[m]];
        mb.n_faces = nf;
        mb.nodes = &nodes;
        mb.leaf_faces = &lfaces;
        mb.leaf_idx = &lidx;
        uint32_t root = mb.build();
        BvhWalkStats st;
        std::string e = host_check_threaded(nodes, root, root, 0x0FFFFFFFu, true, 0, st);
        if (!e.empty()) return bad("mesh bvh: " + e);
        {
            std::vector<uint32_t> visits(lidx.size(), 0);
            BvhWalkStats w;
            e = host_walk_threaded(nodes, root, visits, w);
            if (!e.empty()) return bad("mesh bvh: " + e);
            for (uint32_t v : visits) if (v != 1) return bad("mesh bvh: a leaf slot is not visited exactly once");
            st.nodes = w.nodes; st.leaves = w.leaves;
        }
        {   // the links the device walks (wide-cone inner nodes spliced out): still every leaf exactly once
            std::vector<float4> dev = nodes;
            std::vector<uint8_t> is_root(dev.size() / 4, 0);
            is_root[root] = 1;
            mesh_collapse_links(dev, is_root);
            std::vector<uint32_t> visits(lidx.size(), 0);
            BvhWalkStats w;
            e = host_walk_threaded(dev, root, visits, w);
            if (!e.empty()) return bad("mesh bvh (collapsed links): " + e);
            for (uint32_t v : visits) if (v != 1) return bad("mesh bvh (collapsed links): a leaf slot is not visited exactly once");
        }
        if (lidx.size() != nf) return bad("mesh bvh: leaf slot count != face count");
        std::vector<uint32_t> seen(nf, 0);
        for (uint32_t i : lidx) { if (i >= nf || seen[i]++) return bad("mesh bvh: face missing or duplicated"); }
        // per node: min face, normal cone, edge and quality bounds, faces inside leaf boxes — from the leaves up
        std::function<std::string(uint32_t, std::vector<uint32_t> &)> collect = [&](uint32_t n, std::vector<uint32_t> &faces) -> std::string {
            uint32_t B;
            memcpy(&B, &nodes[4 * (size_t)n + 1].w, 4);
            if (B & 0x80000000u) {
                uint32_t first = B & 0x0FFFFFFFu, cnt = (B >> 28) & 7u;
                for (uint32_t k = 0; k < cnt; k++) faces.push_back(lidx[first + k]);
            } else {
                uint32_t right;
                memcpy(&right, &nodes[4 * (size_t)B].w, 4);
                for (uint32_t k = 0; k < 2; k++) {
                    std::vector<uint32_t> sub;
                    std::string er = collect(k ? (right & 0x0FFFFFFFu) : B, sub);
                    if (!er.empty()) return er;
                    faces.insert(faces.end(), sub.begin(), sub.end());
                }
            }
            const float4 &lo = nodes[4 * (size_t)n], &hi = nodes[4 * (size_t)n + 1], &cone = nodes[4 * (size_t)n + 2], &ex = nodes[4 * (size_t)n + 3];
            uint32_t min_face;
            memcpy(&min_face, &ex.y, 4);
            uint32_t true_min = 0xFFFFFFFFu;
            for (uint32_t f : faces) {
                true_min = std::min(true_min, f);
                const float4 *q = mb.rec + 3 * (size_t)f;
                double A[3] = {q[0].x, q[0].y, q[0].z}, e1[3] = {q[0].w, q[1].x, q[1].y}, e2[3] = {q[1].z, q[1].w, q[2].x};
                for (int k = 0; k < 3; k++) {   // (lo, hi here: the node's centre and half extent)
                    const double l = (double)(&lo.x)[k] - (double)(&hi.x)[k], h = (double)(&lo.x)[k] + (double)(&hi.x)[k];
                    double p[3] = {A[k], A[k] + e1[k], A[k] + e2[k]};
                    for (double v : p) if (v < l || v > h) return std::string("face outside its node box");
                }
                double nl = std::sqrt((double)q[2].y * q[2].y + (double)q[2].z * q[2].z + (double)q[2].w * q[2].w);
                if (std::isfinite(nl) && nl > 0.5 && ex.w > 0.0f) {
                    double dt = (cone.x * q[2].y + cone.y * q[2].z + cone.z * q[2].w) / nl;
                    double cos_a = cone.w, sin_a = ex.x;
                    // the face normal must lie inside the cone (cos/sin describe the half angle; 0/1 = hemisphere or wider)
                    if (!(cos_a == 0.0 && sin_a == 1.0) && dt < cos_a - 1e-5) return std::string("face normal outside the node's cone");
                }
                double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
                if (l1 > ex.z * 1.0001 || l2 > ex.z * 1.0001) return std::string("edge longer than the node's bound");
            }
            if (true_min != min_face) return std::string("wrong smallest face index");
            return "";
        };
        std::vector<uint32_t> all;
        e = collect(root, all);
        if (!e.empty()) return bad("mesh bvh: " + e);
        // the device's 48-byte form of every node (mesh_node_pack): binary16 fields rounded to the safe side
        for (size_t n = root; n < nodes.size() / 4; n++) {
            const float4 *nd = &nodes[4 * n];
            float4 pk[3];
            mesh_node_pack(nd, pk);
            uint32_t p[3];
            memcpy(p, &pk[2], 12);
            const float hx = bvh_half_value(p[0] & 0xFFFFu), hy = bvh_half_value(p[0] >> 16), hz = bvh_half_value(p[1] & 0xFFFFu);
            const float sn = bvh_half_value(p[1] >> 16), em = bvh_half_value(p[2] & 0xFFFFu), q = bvh_half_value(p[2] >> 16);
            if (!(hx >= nd[1].x && hy >= nd[1].y && hz >= nd[1].z)) return bad("mesh bvh: packed half extent below the node's");
            if (!(sn >= nd[3].x) || !(em >= nd[3].z) || !(q <= nd[3].w)) return bad("mesh bvh: packed cone / edge / quality on the wrong side");
            if (memcmp(&pk[0], &nd[0], 16) || memcmp(&pk[1], &nd[2], 12) || memcmp(&pk[1].w, &nd[1].w, 4) || memcmp(&pk[2].w, &nd[3].y, 4))
                return bad("mesh bvh: packed node differs in an unpacked field");
        }
        stats[3] += st.nodes; stats[4] += st.leaves; stats[5] = std::max(stats[5], st.max_depth); stats[6]++;
    }
    return RT_OK;
}

extern "C++" {
namespace {
// upload `bytes` of input, run `launch(d_in, d_out)`, download `out_bytes`
template <class F>
int debug_roundtrip(rt_context *ctx, const void *in, size_t bytes, void *out, size_t out_bytes, F launch) {
    void *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(ctx, hipMalloc(&d_in, bytes ? bytes : 16));
    if (hipMalloc(&d_out, out_bytes ? out_bytes : 16) != hipSuccess) { (void)hipFree(d_in); return fail(ctx, RT_EHIP, "hipMalloc failed"); }
    hipError_t e = hipMemcpyAsync(d_in, in, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) { launch(d_in, d_out); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ctx, RT_EHIP, "debug probe: %s", hipGetErrorString(e));
    return RT_OK;
}
}  // namespace
}  // extern "C++"

int rt_debug_hit(rt_context *ctx, int kind, const float *rays, const uint32_t *prim, const uint32_t *face, size_t n,
                 float *out12) {
    if (!ctx) return RT_EINVAL;
    if (!ctx->have_scene) return fail(ctx, RT_ESTATE, "no scene: call rt_set_scene first");
    if (n == 0) return RT_OK;
    if (!rays || !out12 || kind < 0 || kind > 4 || n > (1u << 26) || (kind != 3 && !prim) || (kind == 4 && !face))
        return fail(ctx, RT_EINVAL, "bad unit-probe arguments");
    const size_t counts[5] = {ctx->spheres.n, ctx->planes.n, ctx->lenses.n, 0, ctx->meshes.n};
    std::vector<uint32_t> pf(2 * n, 0u);
    for (size_t i = 0; i < n; i++) {
        if (kind != 3) {
            if (prim[i] >= counts[kind]) return fail(ctx, RT_ERANGE, "unit probe %zu: primitive %u does not exist", i, prim[i]);
            pf[i] = prim[i];
        }
        if (kind == 4) pf[n + i] = face[i];
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (kind == 4) {  // face indices against the meshes' face counts (host copy of the mesh array)
        std::vector<rt_mesh> meshes(ctx->meshes.n);
        HIP_TRY(ctx, hipMemcpy(meshes.data(), ctx->meshes.p, meshes.size() * sizeof(rt_mesh), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++)
            if (face[i] >= meshes[prim[i]].face_count) return fail(ctx, RT_ERANGE, "unit probe %zu: face %u does not exist", i, face[i]);
    }
    // rays and (prim, face) travel in one buffer: 6 floats + 2 words per record
    std::vector<uint32_t> in(8 * n);
    memcpy(in.data(), rays, 6 * n * sizeof(float));
    memcpy(in.data() + 6 * n, pf.data(), 2 * n * sizeof(uint32_t));
    DeviceScene sc = device_scene(ctx);
    return debug_roundtrip(ctx, in.data(), in.size() * 4, out12, 12 * n * sizeof(float), [&](void *d_in, void *d_out) {
        const float *d_rays = (const float *)d_in;
        const uint32_t *d_prim = (const uint32_t *)d_in + 6 * n, *d_face = d_prim + n;
        (void)ctx->ks->launch_debug_hit(ctx, sc, kind, d_rays, d_prim, d_face, (uint32_t)n, (float *)d_out);
    });
}

int rt_debug_material(rt_context *ctx, int routine, const float *in16, size_t n, float *out9) {
    if (!ctx) return RT_EINVAL;
    if (!ctx->have_scene) return fail(ctx, RT_ESTATE, "no scene: call rt_set_scene first");
    if (n == 0) return RT_OK;
    if (!in16 || !out9 || routine < 0 || routine > 3 || n > (1u << 26)) return fail(ctx, RT_EINVAL, "bad unit-probe arguments");
    for (size_t i = 0; i < n; i++) {
        uint32_t w[4];
        memcpy(w, in16 + 16 * i + 12, sizeof w);
        if (w[0] >= ctx->materials.n) return fail(ctx, RT_ERANGE, "unit probe %zu: material %u does not exist", i, w[0]);
        if (w[1] > RT_MAX_SAMPLE + RT_DEPTH || w[2] >= RT_MAX_DIM || w[3] >= RT_MAX_DIM)
            return fail(ctx, RT_EINVAL, "unit probe %zu: seed / pixel outside the supported range", i);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DeviceScene sc = device_scene(ctx);
    return debug_roundtrip(ctx, in16, 16 * n * sizeof(float), out9, 9 * n * sizeof(float), [&](void *d_in, void *d_out) {
        (void)ctx->ks->launch_debug_material(ctx, sc, routine, (const float *)d_in, (uint32_t)n, (float *)d_out);
    });
}

int rt_debug_div3(rt_context *ctx, const float *in4, size_t n, float *out6) {
    if (!ctx || !in4 || !out6 || n > (1u << 28)) return RT_EINVAL;
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return debug_roundtrip(ctx, in4, 4 * n * sizeof(float), out6, 6 * n * sizeof(float), [&](void *d_in, void *d_out) {
        (void)ctx->ks->launch_debug_div3(ctx, (const float *)d_in, (uint32_t)n, (float *)d_out);
    });
}

int rt_shard_slots(rt_context *ctx, int world, uint32_t *slots_out) {
    if (!ctx || !slots_out || world < 1) return RT_EINVAL;
    uint32_t tw = 1u << ctx->tile_w_log2, th = 1u << ctx->tile_h_log2;
    uint32_t tiles = ((ctx->width + tw - 1) / tw) * ((ctx->height + th - 1) / th);
    *slots_out = ((tiles + world - 1) / world) * tw * th;  // the same for every rank of `world`
    return RT_OK;
}

int rt_pack_accum(rt_context *ctx, void *d_packed, size_t bytes) {
    if (!ctx || !d_packed) return RT_EINVAL;
    uint32_t n = 0;
    rt_shard_slots(ctx, ctx->world, &n);
    if (bytes != (size_t)n * sizeof(float4)) return fail(ctx, RT_EINVAL, "packed buffer must be %zu bytes", (size_t)n * sizeof(float4));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float cam0[12] = {0};
    FrameParams fp = frame_params(ctx, cam0, 0, 0, 0);
    hipLaunchKernelGGL(pt_pack, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, fp, ctx->d_accum, (float4 *)d_packed, n);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

int rt_unpack_accum(rt_context *ctx, const void *d_packed, size_t bytes, int src_rank, int world) {
    if (!ctx || !d_packed) return RT_EINVAL;
    if (world < 1 || src_rank < 0 || src_rank >= world) return fail(ctx, RT_EINVAL, "rank %d of %d", src_rank, world);
    uint32_t n = 0;
    rt_shard_slots(ctx, world, &n);
    if (bytes != (size_t)n * sizeof(float4)) return fail(ctx, RT_EINVAL, "packed buffer must be %zu bytes", (size_t)n * sizeof(float4));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float cam0[12] = {0};
    FrameParams fp = frame_params(ctx, cam0, 0, 0, 0, src_rank, world);  // the SENDER's shard
    if (fp.slot_end)
        hipLaunchKernelGGL(pt_unpack, dim3((fp.slot_end + 255) / 256), dim3(256), 0, ctx->stream, fp, (const float4 *)d_packed, ctx->d_accum);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

int rt_set_option(rt_context *ctx, int option, int value) {
    if (!ctx) return RT_EINVAL;
    switch (option) {
        case RT_OPT_PREFIX_SHARING: ctx->prefix_sharing = value != 0; return RT_OK;
        case RT_OPT_SAMPLE_QUEUE: ctx->sample_queue = value != 0; return RT_OK;
        case RT_OPT_WALK_SLICES: ctx->walk_slices = value != 0; return RT_OK;
        case RT_OPT_WAVE_FILL: ctx->wave_fill = value != 0; return RT_OK;
        case RT_OPT_PREFIX_TREE:
            if (value < 0 || value > 2) return fail(ctx, RT_EINVAL, "RT_OPT_PREFIX_TREE takes 0, 1 or 2");
            ctx->prefix_tree = value;
            return RT_OK;
        case RT_OPT_ACCEL:
            if (value < 0 || value > 2) return fail(ctx, RT_EINVAL, "RT_OPT_ACCEL takes 0, 1 or 2");
            ctx->accel = value;
            return RT_OK;
        case RT_OPT_ARITH: {
            const KernelSet *ks = value == RT_ARITH_IEEE ? kernel_set_a0() : value == RT_ARITH_ROCM_OCL_NOCONTRACT ? kernel_set_a1()
                                  : value == RT_ARITH_ROCM_OCL ? kernel_set_a2() : nullptr;
            if (!ks) return fail(ctx, RT_EINVAL, "RT_OPT_ARITH takes RT_ARITH_IEEE (0), RT_ARITH_ROCM_OCL_NOCONTRACT (1) or RT_ARITH_ROCM_OCL (2)");
            HIP_TRY(ctx, hipSetDevice(ctx->device));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            ctx->arith = value;
            ctx->ks = ks;
            ctx->sample_counter = 0;
            return ctx->have_scene ? apply_arith(ctx) : RT_OK;
        }
        case RT_OPT_MAX_THREADS_PER_LAUNCH:
            if (value < 256) return fail(ctx, RT_EINVAL, "max threads per launch must be >= 256");
            ctx->max_threads_per_launch = (uint32_t)value;
            return RT_OK;
        default: return fail(ctx, RT_EINVAL, "unknown option %d", option);
    }
}

int rt_reset_counters(rt_context *ctx) {
    if (!ctx) return RT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, COUNTER_REPLICAS * COUNTER_STRIDE * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_walk_overflow, 0, sizeof(uint32_t), ctx->stream));
    return RT_OK;
}

int rt_get_counters(rt_context *ctx, rt_counters *out) {
    if (!ctx || !out) return RT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<unsigned long long> h(COUNTER_REPLICAS * COUNTER_STRIDE);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->d_counters, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t *o = (uint64_t *)out;
    for (int i = 0; i < 14; i++) {
        o[i] = 0;
        for (int r = 0; r < COUNTER_REPLICAS; r++) o[i] += h[(size_t)r * COUNTER_STRIDE + i];
    }
    return RT_OK;
}

int rt_get_debug_counters(rt_context *ctx, uint64_t out[2]) {
    if (!ctx || !out) return RT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<unsigned long long> h(COUNTER_REPLICAS * COUNTER_STRIDE);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->d_counters, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    out[0] = out[1] = 0;
    for (int r = 0; r < COUNTER_REPLICAS; r++) {
        out[0] += h[(size_t)r * COUNTER_STRIDE + 14];
        out[1] += h[(size_t)r * COUNTER_STRIDE + 15];
    }
#ifdef PT_WSTAT
    {   // diagnostic build: lane census of pt_samples_w (tools/wstat.py)
        unsigned long long v[12];
        if (hipMemcpy(v, ctx->d_counters + COUNTER_REPLICAS * COUNTER_STRIDE + 8, sizeof v, hipMemcpyDeviceToHost) == hipSuccess) {
            const char *names[12] = {"outer iterations", "active lanes", "phase-0 lanes", "phase-1 lanes", "phase-2 lanes", "walk slices",
                                     "walk steps", "node-test lanes", "(unused)", "leaf phases", "leaf lanes", "finished (idle) lanes"};
            for (int k = 0; k < 12; k++) fprintf(stderr, "[wstat] %-24s %llu\n", names[k], v[k]);
            (void)hipMemset(ctx->d_counters + COUNTER_REPLICAS * COUNTER_STRIDE + 8, 0, sizeof v);
        }
    }
#endif
#if PT_STAMPS
    {   // diagnostic build: print the s_memtime shares of pt_samples_q's sections
        unsigned long long st[8];
        if (hipMemcpy(st, ctx->d_counters + COUNTER_REPLICAS * COUNTER_STRIDE, sizeof st, hipMemcpyDeviceToHost) == hipSuccess) {
            const char *names[6] = {"refill", "scatter", "hit:spheres", "hit:planes/lenses/models", "hit:rebuild+material", "loop"};
            double tot = 0;
            for (int k = 0; k < 6; k++) tot += (double)st[k];
            for (int k = 0; k < 6; k++) fprintf(stderr, "[stamps] %-26s %5.1f %%\n", names[k], tot > 0 ? 100.0 * st[k] / tot : 0.0);
        }
    }
#endif
    return RT_OK;
}

int rt_walk_overflow(rt_context *ctx, uint32_t *flags_out) {
    if (!ctx || !flags_out) return RT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(flags_out, ctx->d_walk_overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_debug_builtin(rt_context *ctx, int op, const float *in8, size_t n, float *out4) {
    if (!ctx || !in8 || !out4 || op < 0 || op > 9 || n > (1u << 26)) return RT_EINVAL;
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return debug_roundtrip(ctx, in8, 8 * n * sizeof(float), out4, 4 * n * sizeof(float), [&](void *d_in, void *d_out) {
        (void)ctx->ks->launch_debug_builtin(ctx, op, (const float *)d_in, (uint32_t)n, (float *)d_out);
    });
}

uint64_t rt_counters_bytes(const rt_counters *c) {
    if (!c) return 0;
    return 32 * c->t_sphere + 48 * c->t_plane + 64 * c->t_lens + 12 * c->t_model + 16 * c->t_mesh + 60 * c->t_tri +
           36 * c->h_tri + 48 * c->h_bounce + 12 * c->n_scatter + 4 * c->n_dielectric + 64 * c->n_texfetch +
           (48 + 16) * c->samples + 16 * c->image_reads;
}

int rt_kernel_ms_history(rt_context *ctx, float *ms, size_t cap, size_t *n_out) {
    if (!ctx || !ms || !n_out) return RT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t have = ctx->ev_count < (uint64_t)rt_context::EV_RING ? (size_t)ctx->ev_count : (size_t)rt_context::EV_RING;
    size_t n = have < cap ? have : cap;
    for (size_t i = 0; i < n; i++) {  // oldest of the last n first
        hipEvent_t *evp = ctx->ev[(ctx->ev_count - n + i) % rt_context::EV_RING];
        HIP_TRY(ctx, hipEventSynchronize(evp[1]));
        HIP_TRY(ctx, hipEventElapsedTime(&ms[i], evp[0], evp[1]));
    }
    *n_out = n;
    return RT_OK;
}

int rt_stage_ms_history(rt_context *ctx, float *first_ms, float *second_ms, size_t cap, size_t *n_out) {
    if (!ctx || !first_ms || !second_ms || !n_out) return RT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t have = ctx->ev_count < (uint64_t)rt_context::EV_RING ? (size_t)ctx->ev_count : (size_t)rt_context::EV_RING;
    size_t n = have < cap ? have : cap;
    for (size_t i = 0; i < n; i++) {  // oldest of the last n first
        hipEvent_t *evp = ctx->ev[(ctx->ev_count - n + i) % rt_context::EV_RING];
        HIP_TRY(ctx, hipEventSynchronize(evp[1]));
        HIP_TRY(ctx, hipEventElapsedTime(&first_ms[i], evp[0], evp[2]));
        HIP_TRY(ctx, hipEventElapsedTime(&second_ms[i], evp[2], evp[1]));
    }
    *n_out = n;
    return RT_OK;
}

int rt_last_kernel_ms(rt_context *ctx, float *ms) {
    if (!ctx || !ms) return RT_EINVAL;
    if (ctx->ev_count == 0) return fail(ctx, RT_ESTATE, "no render call has been timed yet");
    size_t n = 0;
    return rt_kernel_ms_history(ctx, ms, 1, &n);
}

int rt_device_info(rt_context *ctx, char *name, size_t name_len, int *cu_count, char *arch, size_t arch_len) {
    if (!ctx) return RT_EINVAL;
    if (name && name_len) snprintf(name, name_len, "%s", ctx->dev_name.c_str());
    if (arch && arch_len) snprintf(arch, arch_len, "%s", ctx->dev_arch.c_str());
    if (cu_count) *cu_count = ctx->cu_count;
    return RT_OK;
}

}  // extern "C"
