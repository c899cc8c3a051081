// pt_types.hpp — plain data shared by every translation unit of librt_amd.so: what the kernels read (DeviceScene),
// the frame / launch parameters, the per-pixel prefix record, the work-counter slots and the LDS sizing rules.
// No arithmetic lives here: the device functions are compiled once per ARITHMETIC POLICY (pt_arith.hpp, pt_device.hpp,
// pt_kernels.hip) in namespaces of their own, and all of them share these layouts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_amd.h"

namespace pt {

#define PT_DEV __device__ __forceinline__
#define PT_HD __host__ __device__ inline

#define PT_LDS_MATERIALS 64
#define PT_LDS_WINNERS 64    // spheres whose (pos, r, mat) are also staged in LDS: the winner's record is
                             // a per-lane fetch on the critical path of every bounce (global: ~600 cycles)  // materials staged in LDS (the .scene grammar allows 10, src/scene.cpp:455)
#define PT_SPHERE_BATCH 4

// Everything the kernels read.  Passed by value (kernarg segment → SGPRs); this
// replaces the reference's device-resident Scene struct and its createScene
// pointer-stashing kernel (:74-91, :541-558).
struct DeviceScene {
    const rt_material *materials;
    const rt_sphere *spheres;
    const float4 *sph4;  // (cx, cy, cz, r*r) per sphere, padded to whole batches + one dummy batch
    const rt_plane *planes;
    const rt_lens *lenses;
    const rt_float3 *vertices;
    const rt_float2 *uvs;
    const uint32_t *indices;
    const rt_mesh *meshes;
    const rt_model *models;
    const float *table;  // 400 000 floats
    const float4 *tex;   // layers × h × w texels
    int tex_w, tex_h, tex_layers;
    float tex_wf, tex_hf;  // (float)tex_w, (float)tex_h: scalar operands of the sampler (converted per lane they end up hoisted into VGPRs)
    uint32_t material_count, sphere_count, sphere_batches, plane_count, lens_count, model_count;
    // per-face records of all meshes, 3 float4 each: (A.xyz, e1.x), (e1.yz, e2.xy), (e2.z, n.xyz) with
    // e1 = B−A, e2 = C−A, n = normalize(cross(e1,e2)) computed once at upload with the same binary32
    // operations hitTriangle performs per test (:264-265,285); faces of mesh m start at mesh_face_base[m]
    const float4 *faces;
    const uint32_t *mesh_face_base;
    // optional per-mesh BVHs (pt_mesh_bvh.hpp); mesh_bvh_root == nullptr or root == NONE → face scan
    const float4 *mbvh_nodes;       // 3 float4 per node (pt_mesh_bvh.hpp)
    const float4 *mbvh_faces;       // face records (as `faces`) in leaf order
    const uint32_t *mbvh_face_idx;  // their face index inside the mesh
    const uint32_t *mesh_bvh_root;  // per mesh
    // optional sphere BVH (see hit_spheres_bvh); bvh_node_count == 0 → brute force
    const float4 *bvh_nodes;   // 1 float4 per node: (box centre.xyz, left child | leaf)
    const float4 *bvh_sph;     // spheres in leaf order: (cx, cy, cz, r*r)
    const uint32_t *bvh_idx;   // their original indices
    const float4 *bvh_links;   // 8 per node, one per direction octant o: (skip link | split axis << 28, box half extent.xyz) (hit_spheres_bvh)
    uint32_t bvh_node_count;
    float bvh_lo[3], bvh_hi[3];  // bounds of all sphere CENTRES
    float bvh_rmax;              // largest radius
    uint32_t *walk_overflow;     // sticky PT_OVF_* bits (one atomicOr on the cold exit path of a walk)
};

// per-lane work counters (only in COUNT builds)
#define PT_N_COUNTERS 16
struct LaneCounters {
    uint32_t c[PT_N_COUNTERS];
};
enum {
    CN_SAMPLES, CN_BOUNCES, CN_T_SPHERE, CN_T_PLANE, CN_T_LENS, CN_T_MODEL, CN_T_MESH, CN_T_TRI, CN_H_TRI,
    CN_H_BOUNCE, CN_N_SCATTER, CN_N_DIELECTRIC, CN_N_TEXFETCH, CN_IMAGE_READS,
    CN_DBG_BVH_NODES,   // diagnostics (rt_get_debug_counters): BVH nodes entered
    CN_DBG_BVH_TESTS    // sphere tests actually executed (BVH leaves + brute-force fallback)
};

#define PT_LDS_PLANES 16     // planes whose (normal, mat) are staged likewise
#define PT_LDS_STATIC_FLOAT4 (2 * PT_LDS_MATERIALS + 2 * PT_LDS_WINNERS + PT_LDS_PLANES)  // stage_materials' LDS footprint

// ---- materials in LDS ----------------------------------------------------------
// Compact layout, sized by the scene: materials [0, 2M), sphere winner records [2M, 2M + 2S), plane
// records after them; a set that exceeds its cap is not staged (0 entries, read from global memory).
// pt_samples_q sizes its dynamic LDS by lds_static_used(), the other kernels hold the maximum statically.
PT_HD uint32_t lds_mat_n(uint32_t material_count) { return material_count <= PT_LDS_MATERIALS ? 2u * material_count : 0u; }
PT_HD uint32_t lds_win_n(uint32_t sphere_count) { return sphere_count <= PT_LDS_WINNERS ? 2u * sphere_count : 0u; }
PT_HD uint32_t lds_pln_n(uint32_t plane_count) { return plane_count <= PT_LDS_PLANES ? plane_count : 0u; }
PT_HD uint32_t lds_static_used(uint32_t material_count, uint32_t sphere_count, uint32_t plane_count) {
    return lds_mat_n(material_count) + lds_win_n(sphere_count) + lds_pln_n(plane_count);  // float4 units
}

// ---- shared deterministic prefix ---------------------------------------------------
// The reference does not jitter the primary ray (:500-505), so all samples of a
// pixel follow the SAME path until the first random event (a diffuse / textured /
// dielectric surface).  That prefix is traced once per pixel (kernel pt_prefix)
// and stored as a PixelRec; the per-sample kernel continues from it.  Bits are
// unchanged: the same operations are merely not repeated per sample.
struct PixelRec {        // 80 bytes
    float4 p_kind;       // hit point, w = bits: kind (0 final colour, 1 stochastic vertex) | depth << 8 | type << 16
    float4 n_extra;      // normal, material extra_data
    float4 d;            // incoming ray direction, w = material id bits
    float4 out;          // path colour so far (kind 1) or the pixel's radiance for every sample (kind 0)
    float4 col;          // material colour / texel
};
enum { REC_FINAL = 0, REC_VERTEX = 1, REC_TREE = 2 };

// ---- shared decision tree of a pixel -------------------------------------------------------------------------------
// A dielectric surface is a random event with only TWO outcomes — refract or reflect (raytracer.cl:407-435): which one a
// sample takes depends on its own table entry u (reflect_prob < u → refract), but the two continuations are the same
// two rays for every sample of the pixel.  They are therefore followed ONCE per pixel, each up to its next random event,
// and the result is kept as a small binary tree (PT_TREE_LEVELS levels of decisions, heap-indexed: node h has the
// children 2h and 2h + 1): decision nodes hold what a sample needs to choose — the hash of the incoming direction and
// the bounce index that form its table index, and the reflect probability — leaves are ordinary pixel records (a
// diffuse / textured vertex, a final colour, or — below the last level — a dielectric vertex the samples continue
// from on their own).  The sample kernels walk the decisions per sample (one table read each) and start from the
// chosen leaf: the nearest-hit searches behind the glass are done once per pixel instead of once per sample.  A record
// of kind REC_TREE names its tree in col.w.  Bits are unchanged: the same operations, not repeated per sample.
// Built level by level (pt_tree_pass): two work-items per waiting glass vertex, one per continuation.
#ifndef PT_TREE_LEVELS
#define PT_TREE_LEVELS 2      // A/B on MI355X, whole fused call (profiles/r03_experiments.md): C2 off 1.927 ms, 1 level ?, 2 levels 1.822, 3 levels 1.867; C5 at 1080p x 512 spp 174.0 / 149.5 / 149.7
#endif
#define PT_TREE_DECISIONS ((1 << PT_TREE_LEVELS) - 1)
#define PT_TREE_LEAVES (1 << PT_TREE_LEVELS)   // a binary tree with 2^L - 1 inner nodes has at most 2^L leaves
struct TreeDecision {   // 16 bytes
    uint32_t hsh;       // dir_hash of the direction the ray arrives with (random(), :121); [0]: the tree's leaf counter
    float prob;         // Schlick's reflect probability (:420); a sample refracts iff prob < u
    uint32_t depth;     // bounce index i of the vertex (s_seed = i + sample, :471)
    uint16_t child[2];  // [0]: where a refracting sample goes, [1]: a reflecting one — 0x8000 | leaf, or the decision's heap index
};
struct PixelTree {      // 8 x 16 + 8 x 80 = 768 bytes at 3 levels
    TreeDecision dec[PT_TREE_DECISIONS + 1];   // heap-indexed from 1
    PixelRec leaf[PT_TREE_LEAVES];
};
struct TreeWork {       // a glass vertex waiting to become node `heap` of tree `tree`
    PixelRec rec;
    uint32_t tree, heap, _pad[2];
};

enum RenderMode { MODE_ACCUM = 0, MODE_TRACE = 1, MODE_RETRACE = 2 };

struct FrameParams {
    float cam[12];
    int w, h;
    int tile_w_log2, tile_h_log2;
    uint32_t tiles_x, tiles_total;
    uint32_t rank, world;
    uint32_t slot_begin, slot_end;  // owned pixel slots handled by this launch
    uint32_t first, count;          // samples first .. first+count-1
    uint32_t group_log2;            // lanes per pixel = 1 << group_log2 (<= 64)
    uint32_t seg_cap;               // live list: entries per segment (see LIVE_SEGMENTS)
    float inv_count;                // 1 / count, the IEEE quotient computed on the host: a scalar operand of the queue kernels
    PixelTree *trees;               // shared decision trees of this launch (pt_tree writes, the sample kernels read)
    uint32_t *glass;                // live-list positions of the pixels whose first random event is a dielectric surface
    uint32_t *tree_count;           // how many pt_prefix listed (tree i belongs to glass[i])
    uint32_t tree_cap;              // capacity of `trees` (0: trees off — RT_OPT_PREFIX_TREE, counting builds)
    uint32_t lds_face_f4;           // pt_samples_q: float4 of DeviceScene::faces staged in LDS after the static tables (0: none; see launch_fused)
};

// The live list (pixels that need per-sample work) can be kept in LIVE_SEGMENTS independent segments, workgroup b
// of pt_prefix appending to segment b mod LIVE_SEGMENTS and the sample kernels dealing their waves over the
// segments.  Built to take the append counter off a single address; measured on MI355X (profiles/r02_experiments.md):
// pt_prefix 0.198 → 0.075 ms, but pt_samples_q 2.35 → 2.75 (4 segments) … 3.07 ms (64) on C2 and 10.8 → 14.8 ms on
// C3 — the waves of a workgroup (and neighbouring workgroups) then work on distant parts of the image, finish at
// different times and hold their workgroup's LDS and wave slots until the slowest is through.  The list's ORDER
// is a performance property: 1 segment ships, and the counter is relieved by one atomic per workgroup instead.
#ifndef LIVE_SEGMENTS
#define LIVE_SEGMENTS 1u
#endif
#define LIVE_COUNT_STRIDE 32u   // counters 128 bytes apart: one L2 line each
#define LIVE_HEAVY_COUNTER 24u   // word of the live-count block: how many HEAVY pixels pt_prefix stored from the end of the list downwards
#define LIVE_TREE_COUNTER 16u   // words of the live-count block (zeroed with it): [16] glass-first pixels = trees, [17], [18] the work of levels 1, 2

// Counters are spread over COUNTER_REPLICAS rows (one per workgroup residue) so
// that two million waves do not serialise on 14 addresses; the host sums the rows.
#define COUNTER_REPLICAS 512
#define COUNTER_STRIDE 16

// owned pixel slot → frame coordinates.  Slots enumerate this rank's tiles
// (t = rank, rank+world, ...) tile after tile, row-major inside a tile.
PT_DEV bool slot_to_pixel(const FrameParams &fp, uint32_t slot, uint32_t &x, uint32_t &y) {
    uint32_t tpix_log2 = fp.tile_w_log2 + fp.tile_h_log2;
    uint32_t k = slot >> tpix_log2, in = slot & ((1u << tpix_log2) - 1u);
    uint32_t t = fp.rank + k * fp.world;
    if (t >= fp.tiles_total) return false;
    uint32_t tx = t % fp.tiles_x, ty = t / fp.tiles_x;
    x = (tx << fp.tile_w_log2) + (in & ((1u << fp.tile_w_log2) - 1u));
    y = (ty << fp.tile_h_log2) + (in >> fp.tile_w_log2);
    return x < (uint32_t)fp.w && y < (uint32_t)fp.h;
}

#define PT_MESH_BVH_NONE 0xFFFFFFFFu   // mesh_bvh_root[m]: mesh m has no BVH (face scan)

// Sticky "a walk ran out of its loop bound" bits (DeviceScene::walk_overflow; rt_walk_overflow()): a walk that ends
// on its bound instead of at the end of the tree returns a possibly wrong nearest hit — it must never happen, and
// every GPU test asserts that it has not.
#define PT_OVF_SPHERE_WALK 1u
#define PT_OVF_MESH_WALK 2u
#define PT_OVF_WALK_SLICES 4u

}  // namespace pt
