// rt_context.hpp — the context behind the C ABI (include/rt_amd.h) and the helpers every translation unit of
// librt_amd.so shares: rt_amd.hip (host side: C ABI, scene upload, BVH builders, policy-free kernels) and
// pt_kernels.hip (the path-tracing kernels and their launchers, compiled once per arithmetic policy).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "pt_types.hpp"

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t upload(const T *src, size_t count) {
        release();
        size_t alloc = count ? count : 1;  // empty arrays become 1-element dummies (src/scene.cpp:41-44)
        hipError_t e = hipMalloc((void **)&p, alloc * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return e; }
        n = count;
        if (count) e = hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
        else e = hipMemset(p, 0, sizeof(T));
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

namespace pt { struct KernelSet; }

#define ACCEL_MIN_SPHERES 64
#define RT_SPP_PER_LAUNCH 512u   // samples per pixel of one trace launch (rt_render_spp splits larger calls); = QUEUE_SLOTS of pt_kernels.hip
#ifndef MESH_BVH_MIN_FACES
#define MESH_BVH_MIN_FACES 32
#endif

struct rt_context {
    int device = 0;
    int width = 0, height = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    static constexpr int EV_RING = 64;   // event pairs of the last EV_RING render calls
    hipEvent_t ev[EV_RING][3] = {};      // [0] before the call, [1] after it, [2] between the fused call's two stages
    uint64_t ev_count = 0;
    std::string error;
    std::string dev_name, dev_arch;
    int cu_count = 0;

    DevBuf<rt_material> materials;
    DevBuf<rt_sphere> spheres;
    DevBuf<float4> sph4;
    uint32_t sphere_batches = 0;
    DevBuf<float4> faces;
    DevBuf<uint32_t> mesh_face_base;
    DevBuf<float4> mbvh_nodes, mbvh_faces;
    DevBuf<uint32_t> mbvh_face_idx, mesh_bvh_root;
    bool have_mesh_bvh = false;
    DevBuf<float4> bvh_nodes, bvh_sph;
    DevBuf<uint32_t> bvh_idx;
    DevBuf<float4> bvh_links;
    uint32_t bvh_node_count = 0;
    float bvh_lo[3] = {0, 0, 0}, bvh_hi[3] = {0, 0, 0}, bvh_rmax = 0;
    int accel = 1;  // RT_OPT_ACCEL: 0 brute force, 1 BVH for >= ACCEL_MIN_SPHERES spheres, 2 always BVH
    int arith = 0;  // RT_OPT_ARITH: the arithmetic policy (pt_arith.hpp) whose kernels render
    const pt::KernelSet *ks = nullptr;   // that policy's launchers
    // host copies of what depends on the policy (re-derived by rt_set_option(RT_OPT_ARITH)): the spheres' test records
    // hold r*r (policies 0, 1) or r (policy 2), the per-face normals are computed ON THE DEVICE with the policy's
    // cross / normalize for policies 1, 2
    std::vector<rt_sphere> h_spheres;
    std::vector<uint32_t> h_bvh_idx;
    std::vector<float4> h_faces, h_mbvh_faces;
    DevBuf<rt_plane> planes;
    DevBuf<rt_lens> lenses;
    DevBuf<rt_float3> vertices;
    DevBuf<rt_float2> uvs;
    DevBuf<uint32_t> indices;
    DevBuf<rt_mesh> meshes;
    DevBuf<rt_model> models;
    DevBuf<float> table;
    DevBuf<float4> tex;
    int tex_w = 1, tex_h = 1, tex_layers = 0;
    bool have_scene = false;
    bool scene_uses_textures = false;
    uint32_t max_texture_id = 0;

    float4 *d_image = nullptr;
    float4 *d_accum = nullptr;
    unsigned long long *d_counters = nullptr;
    uint32_t *d_walk_overflow = nullptr;   // one word of sticky PT_OVF_* bits
    pt::PixelRec *d_recs = nullptr;      // per owned pixel slot: shared path prefix (fused path)
    uint32_t *d_live = nullptr;      // slots that need per-sample work + [capacity] = their count
    size_t slot_capacity = 0;
    pt::PixelTree *d_trees = nullptr;   // shared decision trees of dielectric-first pixels (pt_types.hpp), tree_capacity of them
    uint32_t *d_glass = nullptr;        // live-list positions of those pixels (pt_prefix → pt_tree_pass)
    pt::TreeWork *d_tree_work = nullptr;  // glass vertices waiting for the next level: two queues of tree_capacity entries
    size_t tree_capacity = 0;
    bool wave_fill = true;              // RT_OPT_WAVE_FILL
    int prefix_tree = 1;                // RT_OPT_PREFIX_TREE: 0 off, 1 from PT_TREE_MIN_SAMPLES samples per call on, 2 always
    bool prefix_sharing = true;
    bool sample_queue = true;
    DevBuf<uint2> walk_jobs;         // (mesh, model material) of every model's meshes in hit order; empty unless
                                     // every one of them has a BVH (pt_samples_w)
    bool walk_slices = true;         // RT_OPT_WALK_SLICES
    uint32_t accum_count = 0;
    uint32_t sample_counter = 0;
    bool count_enabled = false;

    int rank = 0, world = 1, tile_w_log2 = 3, tile_h_log2 = 3;
    uint32_t max_threads_per_launch = 1u << 30;
};

namespace rtamd {
using namespace pt;

// records the message on the context (or, for ctx == nullptr, as the last rt_create failure) and returns `code`
int fail(rt_context *ctx, int code, const char *fmt, ...);

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(ctx, RT_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline DeviceScene device_scene(const rt_context *ctx) {
    DeviceScene s;
    s.materials = ctx->materials.p;
    s.spheres = ctx->spheres.p;
    s.sph4 = ctx->sph4.p;
    s.sphere_batches = ctx->sphere_batches;
    s.material_count = (uint32_t)ctx->materials.n;
    s.faces = ctx->faces.p;
    s.mesh_face_base = ctx->mesh_face_base.p;
    s.mbvh_nodes = ctx->mbvh_nodes.p;
    s.mbvh_faces = ctx->mbvh_faces.p;
    s.mbvh_face_idx = ctx->mbvh_face_idx.p;
    s.mesh_bvh_root = (ctx->have_mesh_bvh && ctx->accel != 0) ? ctx->mesh_bvh_root.p : nullptr;
    bool use_bvh = ctx->bvh_node_count && (ctx->accel == 2 || (ctx->accel == 1 && ctx->spheres.n >= ACCEL_MIN_SPHERES));
    s.bvh_nodes = ctx->bvh_nodes.p;
    s.bvh_sph = ctx->bvh_sph.p;
    s.bvh_idx = ctx->bvh_idx.p;
    s.bvh_links = ctx->bvh_links.p;
    s.bvh_node_count = use_bvh ? ctx->bvh_node_count : 0;
    for (int k = 0; k < 3; k++) { s.bvh_lo[k] = ctx->bvh_lo[k]; s.bvh_hi[k] = ctx->bvh_hi[k]; }
    s.bvh_rmax = ctx->bvh_rmax;
    s.walk_overflow = ctx->d_walk_overflow;
    s.planes = ctx->planes.p;
    s.lenses = ctx->lenses.p;
    s.vertices = ctx->vertices.p;
    s.uvs = ctx->uvs.p;
    s.indices = ctx->indices.p;
    s.meshes = ctx->meshes.p;
    s.models = ctx->models.p;
    s.table = ctx->table.p;
    s.tex = ctx->tex.p;
    s.tex_w = ctx->tex_w;
    s.tex_h = ctx->tex_h;
    s.tex_layers = ctx->tex_layers;
    s.tex_wf = (float)ctx->tex_w;
    s.tex_hf = (float)ctx->tex_h;
    s.sphere_count = (uint32_t)ctx->spheres.n;
    s.plane_count = (uint32_t)ctx->planes.n;
    s.lens_count = (uint32_t)ctx->lenses.n;
    s.model_count = (uint32_t)ctx->models.n;
    return s;
}

struct Shard {
    uint32_t tiles_x, tiles_total, owned_tiles, slots;
};
inline Shard shard_of(const rt_context *ctx, int rank, int world) {
    Shard s;
    uint32_t tw = 1u << ctx->tile_w_log2, th = 1u << ctx->tile_h_log2;
    s.tiles_x = (ctx->width + tw - 1) / tw;
    uint32_t tiles_y = (ctx->height + th - 1) / th;
    s.tiles_total = s.tiles_x * tiles_y;
    s.owned_tiles = s.tiles_total > (uint32_t)rank ? (s.tiles_total - rank + world - 1) / world : 0;
    s.slots = s.owned_tiles * tw * th;
    return s;
}

// frame parameters for the shard (rank of world); rank < 0 → the context's own shard
inline FrameParams frame_params(const rt_context *ctx, const float cam[12], uint32_t first, uint32_t count, uint32_t glog2,
                         int rank = -1, int world = 1) {
    if (rank < 0) { rank = ctx->rank; world = ctx->world; }
    FrameParams fp;
    memcpy(fp.cam, cam, sizeof fp.cam);
    Shard sh = shard_of(ctx, rank, world);
    fp.w = ctx->width;
    fp.h = ctx->height;
    fp.tile_w_log2 = ctx->tile_w_log2;
    fp.tile_h_log2 = ctx->tile_h_log2;
    fp.tiles_x = sh.tiles_x;
    fp.tiles_total = sh.tiles_total;
    fp.rank = (uint32_t)rank;
    fp.world = (uint32_t)world;
    fp.slot_begin = 0;
    fp.slot_end = sh.slots;
    fp.first = first;
    fp.count = count;
    fp.group_log2 = glog2;
    fp.seg_cap = 0;
    fp.trees = nullptr;
    fp.glass = nullptr;
    fp.tree_count = nullptr;
    fp.tree_cap = 0;
    fp.lds_face_f4 = 0;
    {
        volatile float c = (float)count;
        volatile float q = 1.0f / c;
        fp.inv_count = count ? q : 0.0f;
    }
    return fp;
}

// kernels come in (COUNT, ACCEL) instantiations; scenes without any BVH run the ACCEL = false ones
inline bool scene_has_accel(const DeviceScene &sc) { return sc.bvh_node_count != 0 || sc.mesh_bvh_root != nullptr; }

inline int ensure_slots(rt_context *ctx, size_t slots) {
    if (slots <= ctx->slot_capacity) return RT_OK;
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
    // segmented live list: LIVE_SEGMENTS segments of whole workgroups' worth of entries, then the segment counters
    size_t entries = slots + (size_t)LIVE_SEGMENTS * 256u;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_recs, entries * sizeof(PixelRec)));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_live, (entries + (size_t)LIVE_SEGMENTS * LIVE_COUNT_STRIDE) * sizeof(uint32_t)));
    // decision trees for a quarter of the slots (a frame with more dielectric-first pixels keeps plain records for the rest)
    ctx->tree_capacity = slots / 4 + 256;
    if (hipMalloc((void **)&ctx->d_trees, ctx->tree_capacity * sizeof(PixelTree)) != hipSuccess ||
        hipMalloc((void **)&ctx->d_glass, ctx->tree_capacity * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void **)&ctx->d_tree_work, 2 * ctx->tree_capacity * sizeof(TreeWork)) != hipSuccess) {
        (void)hipGetLastError();
        if (ctx->d_trees) (void)hipFree(ctx->d_trees);
        ctx->d_trees = nullptr;       // not fatal: the frame renders without shared trees
        ctx->tree_capacity = 0;
    }
    ctx->slot_capacity = slots;
    return RT_OK;
}

inline uint32_t group_log2_for(uint32_t count) {
    uint32_t g = 0;
    while ((1u << g) < count && g < 6) g++;
    return g;
}

}  // namespace rtamd
