// pt_device.hpp — device-side path tracing for gfx950 (MI355X), wave64.
//
// Hand-written HIP equivalent of everything kernels `trace` / `retrace` of the
// reference reach (kernels/raytracer.cl:93-494; line numbers below are that
// file's).  Every float operation below is the reference's operation at that place, in the
// reference's evaluation order, with the builtins, `/`, sqrt and the contraction of a*b+c as the
// translation unit's ARITHMETIC POLICY defines them (pt_arith.hpp): policy 0 is bit-identical per
// pixel-sample to the reference compiled without FMA contraction and with plain IEEE builtins (the CPU
// oracle's pin), policies 1 and 2 to the reference as ROCm's own OpenCL tool chain builds it for gfx950
// (oracle/_ref_gfx950/*.hsaco).  Compile with -ffp-contract=off and without any fast-math flag.
//
// Mapping to the hardware:
//   * the 64 lanes of a wave are 64 samples of one pixel (or 64/g pixels × g
//     samples), so primitive indices are wave-uniform: primitive records are
//     fetched with SCALAR loads (s_load → scalar cache → L2), four spheres per
//     batch with the next batch prefetched, and cost no VGPRs and no LDS traffic;
//   * materials (per-lane index), their glass constants and the winner records of small sphere
//     sets are staged in LDS once per workgroup;
//   * large sphere sets and large meshes go through conservative BVHs that return the
//     brute-force answer bit for bit (hit_spheres_bvh, pt_mesh_bvh.hpp);
//   * the nearest-hit search keeps only (t, id) per lane and rebuilds the hit
//     record of the winner afterwards (same arithmetic → same bits);
//   * every material kind ends in ONE shared "new direction" tail, so lanes on
//     different materials do not serialise a normalize() each;
//   * table / texture gathers are the only divergent global-memory accesses.
#pragma once
#include "pt_arith.hpp"

namespace PT_NS {

// tuning switches (A/B-tested on MI355X; results never depend on them)
#ifndef PT_RNG_PREFETCH
#define PT_RNG_PREFETCH 0  // 1: issue the table gathers before the nearest-hit search; 2: right after it, for lanes that go on
#endif
#ifndef PT_BEHIND_SKIP
#define PT_BEHIND_SKIP 1   // skip the square root for spheres behind the ray origin
#endif

#ifndef PT_LDS_SPHERES
#define PT_LDS_SPHERES 0   // A/B switch: 1 = the brute-force sphere loop reads an LDS copy of the spheres
#endif                     // (ds_read_b128 broadcast) instead of scalar loads; measured slower, DESIGN.md §5
#define PT_LDS_SPHERE_CAP 256

#ifndef PT_FACE_MASK
#define PT_FACE_MASK 1  // face-scanned meshes of <= 32 faces: facing test first, per-lane candidate lists (hit_models)
#endif
#ifndef PT_NO_MODELS
#define PT_NO_MODELS 0  // experiment: compile the mesh code out
#endif
#ifndef PT_STAMPS
#define PT_STAMPS 0   // diagnostic build: s_memtime shares of the queue kernel's sections (never timed)
#endif
#if PT_STAMPS
#define PT_STAMP(c, k)                                                         \
    do {                                                                       \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();                \
        (c).st[k] += now_ - (c).st_last;                                       \
        (c).st_last = now_;                                                    \
    } while (0)
#else
#define PT_STAMP(c, k) do { } while (0)
#endif


// What a device function needs besides the scene: the workgroup's LDS copy of
// the materials and the lane's counters.
// The staged tables are addressed through LDS-qualified pointers: an `if (table) … else (global record) …` then
// stays a ds_read and a global load in two branches.  Through generic pointers the compiler merges the two reads into
// ONE flat load of a selected pointer — a dozen v_cndmask to build the address, and a vector-memory instruction
// (the queue kernels are bound by their number as much as by ALU work) where an LDS read would do.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(3))) *LdsV4;
PT_DEV float4 lds_ld(LdsV4 p, uint32_t i) {
    v4f t = p[i];
    return make_float4(t.x, t.y, t.z, t.w);
}
PT_DEV LdsV4 lds_ptr(const float4 *generic) { return (LdsV4)generic; }

struct Ctx {
    const DeviceScene &sc;
    LdsV4 lmat;  // LDS: [2i] = (r,g,b,extra), [2i+1].x = type bits; nullptr → read global
    LaneCounters *cn;
    const float4 *lsph = nullptr;  // PT_LDS_SPHERES: LDS copy of sph4 (or nullptr)
    LdsV4 lwin = nullptr;  // LDS winner records of small sphere sets (stage_materials), or nullptr
    LdsV4 lpln = nullptr;  // LDS (normal, mat) of small plane sets, or nullptr
    LdsV4 lfaces = nullptr;  // LDS copy of DeviceScene::faces (scenes of a few small meshes; pt_samples_q), or nullptr
#if PT_STAMPS
    mutable unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    mutable unsigned long long st_last = 0;
#endif
};

// :402-403 — r0 = ((1 − ratio)/(1 + ratio))²
PT_DEV float schlick_r0(float ratio) {
    float r0 = (1.0f - ratio) / (1.0f + ratio);
    r0 *= r0;
    return r0;
}

// call at kernel start by every thread of the workgroup (contains a barrier)
PT_DEV LdsV4 stage_materials(const DeviceScene &sc, float4 *lds) {
    const uint32_t nm = lds_mat_n(sc.material_count), nw = lds_win_n(sc.sphere_count), np = lds_pln_n(sc.plane_count);
    if (nm)
        for (uint32_t i = threadIdx.x; i < sc.material_count; i += blockDim.x) {
            const rt_material &m = sc.materials[i];
            lds[2 * i] = make_float4(m.color.x, m.color.y, m.color.z, m.extra_data);
            // per-material constants of the glass interactions (:379,:402-403), computed once per
            // workgroup by the same operations the bounce would perform: 1/extra and Schlick's r0²
            // for both index ratios
            float inv = 1.0f / m.extra_data;
            lds[2 * i + 1] = make_float4(__int_as_float(m.type), inv, schlick_r0(m.extra_data), schlick_r0(inv));
        }
    // winner records of small sphere sets, after the materials: [2i] = (pos.xyz, r), [2i+1].x = mat_ID bits
    if (nw)
        for (uint32_t i = threadIdx.x; i < sc.sphere_count; i += blockDim.x) {
            const rt_sphere &s = sc.spheres[i];
            lds[nm + 2 * i] = make_float4(s.pos.x, s.pos.y, s.pos.z, s.r);
            lds[nm + 2 * i + 1] = make_float4(__uint_as_float(s.mat_ID), 0.0f, 0.0f, 0.0f);
        }
    // plane winner records: (normal.xyz, mat_ID bits)
    if (np)
        for (uint32_t i = threadIdx.x; i < sc.plane_count; i += blockDim.x) {
            const rt_plane &p = sc.planes[i];
            lds[nm + nw + i] = make_float4(p.normal.x, p.normal.y, p.normal.z, __uint_as_float(p.mat_ID));
        }
    __syncthreads();
    return nm ? lds_ptr(lds) : (LdsV4) nullptr;
}
PT_DEV LdsV4 staged_winners(const DeviceScene &sc, const float4 *lds) {
    return lds_win_n(sc.sphere_count) ? lds_ptr(lds + lds_mat_n(sc.material_count)) : (LdsV4) nullptr;
}
PT_DEV LdsV4 staged_planes(const DeviceScene &sc, const float4 *lds) {
    return lds_pln_n(sc.plane_count) ? lds_ptr(lds + lds_mat_n(sc.material_count) + lds_win_n(sc.sphere_count)) : (LdsV4) nullptr;
}
// PT_LDS_SPHERES experiment: stage the sphere test data of small scenes (after stage_materials' barrier
// has been passed by every thread; contains its own barrier)
PT_DEV const float4 *stage_spheres(const DeviceScene &sc, float4 *lds) {
    uint32_t n = (sc.sphere_batches + 1u) * PT_SPHERE_BATCH;
    if (!PT_LDS_SPHERES || n > PT_LDS_SPHERE_CAP) return nullptr;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) lds[i] = sc.sph4[i];
    __syncthreads();
    return lds;
}

PT_DEV void load_material(const Ctx &c, uint32_t id, int &type, float &extra, V3 &col) {
    if (c.lmat) {
        float4 a = lds_ld(c.lmat, 2 * id);
        type = __float_as_int(lds_ld(c.lmat, 2 * id + 1).x);
        extra = a.w;
        col = mk(a.x, a.y, a.z);
    } else {
        const rt_material &m = c.sc.materials[id];
        type = m.type;
        extra = m.extra_data;
        col = ld3(m.color);
    }
}

// ---- table index (:113-125) --------------------------------------------------
// fp64 hash of the direction; the index sum is below 2^32 for w,h <= RT_MAX_DIM
// and sample <= RT_MAX_SAMPLE + RT_DEPTH, so 32-bit arithmetic equals the
// reference's 64-bit size_t arithmetic.
PT_DEV uint32_t dir_hash(V3 d) {
    float dp = dot(d, mk(123.9898f, 348.233f, 433.3314f));
    return (uint32_t)fabs((double)dp * 438.5453);
}
// Both table reads of a bounce depend only on the INCOMING direction and the
// seed, so they are issued before the nearest-hit search and consumed after it
// (the gather latency hides behind the intersection arithmetic).  randomVec (:113)
// feeds t_diffuse / t_textured, random (:120) feeds t_dielectric.
struct Rnd {
    V3 v;     // three consecutive FLOATS at float offset idx (:109-111,117)
    float u;
};
PT_DEV Rnd fetch_rnd(const float *__restrict__ table, V3 dir, uint32_t s_seed, uint32_t gx, uint32_t gy) {
    uint32_t hsh = dir_hash(dir);
    uint32_t iv = (hsh + (s_seed * 2683u + gx * 3931u + gy * 2504u) * 3u) % RT_RANDOM_BUFFER_SIZE;
    uint32_t iu = (hsh + (s_seed * 2683u + gx * 3931u + gy)) % RT_RANDOM_BUFFER_SIZE;
    const float *t = table + iv;
    Rnd r;
    r.v = mk(t[0], t[1], t[2]);
    r.u = table[3 * RT_RANDOM_BUFFER_SIZE + iu];
    return r;
}

// The same two reads with the per-sample part of the index sums precomputed (queue kernels): for
// s_seed = depth + sample,   (s_seed·2683 + gx·3931 + gy·2504)·3 = depth·8049 + bv   and
// s_seed·2683 + gx·3931 + gy = depth·2683 + bu   in uint32 arithmetic (the sums stay below 2^32, see above).
PT_DEV uint32_t rnd_base_v(uint32_t sample, uint32_t gx, uint32_t gy) { return (sample * 2683u + gx * 3931u + gy * 2504u) * 3u; }
PT_DEV uint32_t rnd_base_u(uint32_t sample, uint32_t gx, uint32_t gy) { return sample * 2683u + gx * 3931u + gy; }
PT_DEV Rnd fetch_rnd_b(const float *__restrict__ table, V3 dir, uint32_t depth, uint32_t bv, uint32_t bu) {
    uint32_t hsh = dir_hash(dir);
    uint32_t iv = (hsh + bv + depth * 8049u) % RT_RANDOM_BUFFER_SIZE;
    uint32_t iu = (hsh + bu + depth * 2683u) % RT_RANDOM_BUFFER_SIZE;
    const float *t = table + iv;
    Rnd r;
    r.v = mk(t[0], t[1], t[2]);
    r.u = table[3 * RT_RANDOM_BUFFER_SIZE + iu];
    return r;
}

// ---- nearest-hit search --------------------------------------------------------
// id of the winning primitive: kind in the top 2 bits
enum : uint32_t { K_SPHERE = 0u << 30, K_PLANE = 1u << 30, K_LENS = 2u << 30, K_MESH = 3u << 30, K_MASK = 3u << 30 };
#define PT_NO_HIT 0xFFFFFFFFu

// The intersection routines return the accepted parameter t, or PT_MISS = +infinity: a caller looking for the
// nearest hit then needs ONE comparison, `t < best_t` (best_t <= MAX_DISTANCE), instead of `t > 0 && t < best_t`
// — a v_cmp costs as much as two multiplies on this chip (profiles/r02_valu_microbench.md).
#define PT_MISS INFINITY
// :149-174.  s = (centre, r²) with r² the same float product, computed once at upload — or, under policy 2,
// (centre, r): there `dot(oc,oc) - r*r` and `b*b - c` are contracted (:152-153), so the product never exists on
// its own.  Returns the accepted root, or PT_MISS.
PT_DEV float sphere_root(float b, float cc, float dis);
PT_DEV void sphere_disc(const Ray &r, float4 s, float &b, float &cc, float &dis) {
    V3 oc = xyz(s) - r.o;
    b = dot(oc, r.d);
#if PT_CONTRACT
    cc = __builtin_fmaf(-s.w, s.w, dot(oc, oc));
    dis = __builtin_fmaf(b, b, -cc);
#else
    cc = dot(oc, oc) - s.w;
    dis = b * b - cc;
#endif
}
PT_DEV float sphere_t(const Ray &r, float4 s) {
    float b, cc, dis;
    sphere_disc(r, s, b, cc, dis);
    return sphere_root(b, cc, dis);
}
// what the host stores in the fourth component of a sphere's test record (sph4, bvh_sph)
#define PT_SPHERE_W_IS_RADIUS PT_CONTRACT
#define PT_BVH_END 0x0FFFFFFFu
#ifndef PT_SPHERE_LEAF_EVERY
#define PT_SPHERE_LEAF_EVERY 4u  // node steps between leaf phases of the sphere BVH walk (power of two)
#endif
// Does this sphere need its roots?  Centre behind the origin (b < 0) and origin outside the sphere
// (cc > 0): the far root is b + sqrt(b*b - cc) <= |b|·2^-23 < MIN_DISTANCE for |b| < 4096, the near
// root is negative — the reference rejects both, so the square root is skipped.  Exact.  (Policies 1, 2: the
// 3-ulp square root makes that |b|·3.6·2^-24; the skip is taken for |b| < 1024 — far root < 2.2e-4.)
PT_DEV bool sphere_needs_roots(float b, float cc, float dis) {
    return dis > 0 && !(PT_BEHIND_SKIP && b < 0.0f && cc > 0.0f && b > (PT_OCL ? -1024.0f : -4096.0f));
}
// :155-171 — the accepted root for dis > 0, or PT_MISS
PT_DEV float sphere_roots(float b, float dis) {
    float d = sqrt1(dis);
    float t = PT_MISS;
    float t0 = b - d;
    if (in_range(t0)) t = t0;
    else {
        float t1 = b + d;
        if (in_range(t1)) t = t1;
    }
    return t;
}
// second half of :149-174: the accepted root, or PT_MISS
PT_DEV float sphere_root(float b, float cc, float dis) {
    return sphere_needs_roots(b, cc, dis) ? sphere_roots(b, dis) : PT_MISS;
}
// ---- sphere BVH -------------------------------------------------------------------
// The reference tests every sphere on every bounce (:327-333).  For large sphere counts
// the same answer — min over spheres of (t, index), t from the very same sphere_t()
// arithmetic — is found through a bounding-volume hierarchy.  It is an acceleration
// only if it can never drop a sphere the reference would have hit, so the culling is
// conservative by construction (DESIGN.md "sphere BVH"):
//   * hitSphere can only succeed when its computed discriminant b² − (|oc|² − r²) is > 0.
//     Its rounding error is below 16·2^-24·|oc|², and for a direction of length² dd ≠ 1 the
//     formula is the reference's own (non-geometric) one: it succeeds iff the LINE passes
//     within sqrt(r²/dd + |oc|²(1 − 1/dd)) of the centre.  Both are covered by inflating a
//     node's box by m = sqrt(c·2)·dfar with c = 2e-6 + 2|dd − 1|/min(dd, 1) and dfar the
//     largest distance from the origin to the box (|oc| and r of every sphere inside are
//     <= dfar), plus rounding slack of the slab test itself;
//   * a node is skipped on distance only if its entry is beyond the best t by a margin;
//   * candidates are compared by (t, original index), so ties resolve as in the
//     reference's in-order scan with strict '<';
//   * rays whose direction is far from unit length (or NaN) take the brute-force loop.
// Traversal is ORDERED (near child first, so the best-t bound prunes the far side) and keeps neither a
// stack nor a way back up: the tree is threaded per direction octant (see hit_spheres_bvh).
//   builder's node = 4 float4: (box centre.xyz, A), (box half extent.xyz, B), 8 skip links;  A = split_axis << 28;  siblings
//   adjacent, the left one at an even index;  B = left child index, or for a leaf 0x80000000 | count << 28 | first sphere.
//   device: bvh_nodes[node] = (centre, B), bvh_links[8·node + octant] = (skip link | A, half extent).
// base + a 32-BIT byte offset: the compiler then uses the scalar-base form of the global load (one 32-bit shift
// instead of 64-bit address arithmetic per access).  The host keeps every BVH array below 4 GiB (rt_set_scene).
template <class T>
PT_DEV const T *at32(const T *base, uint32_t byte_offset) {
    return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_offset);
}

// 1 / d per axis for the slab tests of the BVH walks: the hardware reciprocal, its magnitude capped at 1e30 so that
// o · (1/d) stays finite for an axis-parallel ray (±0 → ±1e30: every slab distance is then astronomically large or an
// exact 0 — the same verdicts as with infinities, without their inf − inf)
PT_DEV V3 cull_inverse(V3 d) {
    V3 i = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    return mk(fminf(fmaxf(i.x, -1.0e30f), 1.0e30f), fminf(fmaxf(i.y, -1.0e30f), 1.0e30f), fminf(fmaxf(i.z, -1.0e30f), 1.0e30f));
}
template <bool COUNT>
PT_DEV void hit_spheres_bvh(const DeviceScene &sc, const Ray &r, float &best_t, uint32_t &best_id, LaneCounters *cn) {
    // per-ray constants of the culling tests: hardware rcp / sqrt (1 ulp) scaled to the safe side — the margins
    // below carry 2 % (k_ray) and 1e-5 relative (slab test) of slack, the sphere tests themselves stay IEEE
    float dd = dot(r.d, r.d);
    float c_ray = 2.0e-6f + 2.0f * fabsf(dd - 1.0f) * (__builtin_amdgcn_rcpf(fminf(dd, 1.0f)) * 1.000002f);
    float k_ray = __builtin_amdgcn_sqrtf(2.0f * c_ray) * 1.02f;   // margin per unit of distance to the node
    float o_max = fmaxf(fmaxf(fabsf(r.o.x), fabsf(r.o.y)), fabsf(r.o.z));
    V3 inv = cull_inverse(r.d);
    // the ray's direction octant: bit k set → along axis k the RIGHT child (higher coordinates) is the nearer one
    const uint32_t oct = (r.d.x < 0.0f ? 1u : 0u) | (r.d.y < 0.0f ? 2u : 0u) | (r.d.z < 0.0f ? 4u : 0u);

    // The tree is THREADED per octant: a node carries, for each of the 8 sign patterns of a direction, the node that
    // follows its subtree in that octant's near-before-far order (the sibling for the child entered first, the
    // parent's link for the other, PT_BVH_END after the last) — so "next" is the nearer child after a hit on an
    // inner node and the octant's skip link otherwise: no way back up, no state, every box tested at most once.
    // "While-while": a node per step for every lane that is neither parked at a leaf nor through; the sphere tests
    // of the parked lanes only every PT_SPHERE_LEAF_EVERY steps, or when every lane is parked or through.  In
    // place, the leaf block (up to four sphere tests with their square roots) ran on almost every step for one or
    // two lanes.
    uint32_t cur = 1u;   // (the root; node 0 is padding so that sibling pairs start at even indices)
    uint32_t best_slot = PT_NO_HIT;   // this walk's nearest sphere so far (the spheres come first in the search: best_id is PT_NO_HIT on entry)
    bool at_leaf = false;
    uint32_t leaf_b = 0, leaf_skip = 0;
    // (the bound counts ITERATIONS: a lane tests every node at most once, and every leaf it meets costs it up to
    // PT_SPHERE_LEAF_EVERY more — parked until the next leaf phase.  A ray whose direction is far from unit length
    // (|d|² up to 1.25: the margin then inflates every box by about the distance to it) visits the whole tree.)
    for (uint32_t guard = 0; guard < (PT_SPHERE_LEAF_EVERY + 1u) * sc.bvh_node_count + 8u; guard++) {
      if (cur != PT_BVH_END && !at_leaf) {
        // TWO loads per node: (centre, B) and the octant's (skip link | split axis << 28, half extent).  The half extent is
        // repeated in each of a node's eight octant records so that the link and it arrive together: the walks are bound
        // by the NUMBER of vector memory instructions (one per ~18 cycles per CU on C4).  Links inside a 64-byte node:
        // C4 at 16 spp 132.3 ms; 32-byte box + a 4-byte link from an array of its own (three loads): 117.8; this: 10 % less again.
        float4 a = *at32(sc.bvh_nodes, cur << 4);
        float4 b = *at32(sc.bvh_links, (cur << 7) + (oct << 4));
        uint32_t skip = __float_as_uint(b.x) & PT_BVH_END;
        {
            const uint32_t axis_bits = __float_as_uint(b.x) & 0x30000000u, child = __float_as_uint(a.w);
            a.w = __uint_as_float(axis_bits);
            b = make_float4(b.y, b.z, b.w, __uint_as_float(child));
        }
        uint32_t A = __float_as_uint(a.w), B = __float_as_uint(b.w);
        // (both header words are pinned here: left to itself the compiler fetches the box as two 12-byte loads and
        // the header words only after a hit, one after the other — two more trips to memory on the way down)
        // (not `volatile`: that would also stop the hoisting and scalarisation of every other load of the kernel)
        asm("" : "+v"(A), "+v"(B), "+v"(skip));
        if (COUNT) cn->c[CN_DBG_BVH_NODES]++;
        // slab test against the inflated box (fminf/fmaxf drop the NaN of 0·inf)
        // (a = the box's centre, b = its half extent: the farthest corner, and below the slab distances, without min / max)
        const V3 dc = mk(a.x - r.o.x, a.y - r.o.y, a.z - r.o.z);
        float fx = fabsf(dc.x) + b.x, fy = fabsf(dc.y) + b.y, fz = fabsf(dc.z) + b.z;
        // (culling arithmetic, not the reference's: fused multiply-adds — fewer instructions, smaller rounding)
        float dfar = __builtin_amdgcn_sqrtf(__builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx))) * 1.001f;  // approximate sqrt, rounded up
        float m = __builtin_fmaf(k_ray, dfar, __builtin_fmaf(1.0e-5f, dfar + o_max, 1.0e-6f));
        // per axis the line is inside the inflated slab for t in [tc − te, tc + te], tc = (c − o)/d, te = (h + m)/|d|
        // (the rounding of this form, 2^-24·(3·dfar + o_max) in units of distance, is 50 × below the slack in m)
        float tcx = dc.x * inv.x, tcy = dc.y * inv.y, tcz = dc.z * inv.z;
        float tex = (b.x + m) * fabsf(inv.x), tey = (b.y + m) * fabsf(inv.y), tez = (b.z + m) * fabsf(inv.z);
        float tmin = fmaxf(fmaxf(tcx - tex, tcy - tey), tcz - tez), tmax = fminf(fminf(tcx + tex, tcy + tey), tcz + tez);
        bool miss = tmin > __builtin_fmaf(fabsf(tmax), 1.0e-5f, tmax) + 1.0e-4f   // the line misses the box
                    || tmax < -1.0e-2f                                             // box entirely behind the origin
                    || tmin > __builtin_fmaf(best_t, 1.00001f, 1.0e-2f);           // box entirely beyond the best hit
        if (miss) {
            cur = skip;
        } else if (B & 0x80000000u) {  // park here: the spheres are tested in the leaf phase below
            at_leaf = true;
            leaf_b = B;
            leaf_skip = skip;
        } else {                       // descend to the nearer child
            cur = B + ((oct >> ((A >> 28) & 3u)) & 1u);
        }
      }
        // ---- leaf phase (wave-uniform decision)
        bool flush = (guard & (PT_SPHERE_LEAF_EVERY - 1u)) == PT_SPHERE_LEAF_EVERY - 1u || __all(at_leaf || cur == PT_BVH_END);
        if (flush && __any(at_leaf)) {
            if (at_leaf) {
                uint32_t first = leaf_b & 0x0FFFFFFFu, cnt = (leaf_b >> 28) & 7u;
                for (uint32_t k = 0; k < cnt; k++) {
                    if (COUNT) cn->c[CN_DBG_BVH_TESTS]++;
                    // ONE load per sphere: the winner is kept as a leaf slot and translated to the sphere's own index
                    // once, after the walk; two spheres at exactly the same t (the smaller index wins, as in the
                    // reference's in-order scan with strict '<') fetch their indices on the spot
                    float t = sphere_t(r, *at32(sc.bvh_sph, (first + k) << 4));
                    bool take = t < best_t;
                    if (t == best_t && best_slot != PT_NO_HIT)
                        take = *at32(sc.bvh_idx, (first + k) << 2) < *at32(sc.bvh_idx, best_slot << 2);
                    if (take) {
                        best_t = t;
                        best_slot = first + k;
                    }
                }
                at_leaf = false;
                cur = leaf_skip;
            }
        }
        if (__all(cur == PT_BVH_END)) break;
    }
    // (cold path: the loop above can only end by its break — a lane still inside the tree here means the iteration
    // bound was too small and a nearer sphere may have been missed: sticky flag, asserted zero by the tests)
    if (cur != PT_BVH_END || at_leaf) atomicOr(sc.walk_overflow, PT_OVF_SPHERE_WALK);
    if (best_slot != PT_NO_HIT) best_id = K_SPHERE | *at32(sc.bvh_idx, best_slot << 2);
}

// :176-194
PT_DEV float plane_t(const Ray &r, V3 p0, V3 n) {
    float a = dot(r.d, n);
    float b = dot(p0 - r.o, n);
    float t = b / a;
    return in_range(t) ? t : PT_MISS;
}

// :196-255 — intersection of two spheres; which = 0 → surface 1 (p1,r1), 1 → surface 2
PT_DEV float lens_t(const Ray &r, const rt_lens &l, int *which) {
    V3 oc = ld3(l.p1) - r.o;
    float b1 = dot(oc, r.d);
    float c = nmad(l.r1, l.r1, dot(oc, oc));   // dot(oc, oc) - r1 * r1   (:199)
    float dis1 = msub(b1, b1, c);              // b1 * b1 - c            (:200)
    oc = ld3(l.p2) - r.o;
    float b2 = dot(oc, r.d);
    c = nmad(l.r2, l.r2, dot(oc, oc));
    float dis2 = msub(b2, b2, c);
    if (dis1 > 0 && dis2 > 0) {
        float d1 = sqrt1(dis1), d2 = sqrt1(dis2);
        float t1A = b1 - d1, t1B = b1 + d1, t2A = b2 - d2, t2B = b2 + d2;
        float t;
        int w;
        if ((t1B < t2A) || (t2B < t1A)) return PT_MISS;
        else if (RT_MIN_DISTANCE <= t1A || RT_MIN_DISTANCE <= t2A) {
            if (t2A <= t1A) { w = 0; t = t1A; } else { w = 1; t = t2A; }
        } else if (RT_MIN_DISTANCE <= t1B && RT_MIN_DISTANCE <= t2B) {
            if (t1B <= t2B) { w = 0; t = t1B; } else { w = 1; t = t2B; }
        } else return PT_MISS;
        if (t <= RT_MAX_DISTANCE) {
            *which = w;
            return t;
        }
    }
    return PT_MISS;
}

// :257-289 — Möller–Trumbore without culling.  Returns t on a hit, PT_MISS otherwise.
PT_DEV float triangle_t(const Ray &r, V3 A, V3 e1, V3 e2, float *u_out, float *v_out) {
    V3 h = cross(r.d, e2);
    float a = dot(e1, h);
    if (a > -RT_TRIANGLE_EPSILON && a < RT_TRIANGLE_EPSILON) return PT_MISS;
    float f = 1.0f / a;
    V3 s = r.o - A;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return PT_MISS;
    V3 q = cross(s, e1);
    float v = f * dot(r.d, q);
    if (v < 0.0f || u + v > 1.0f) return PT_MISS;
    float t = f * dot(e2, q);
    if (!in_range(t)) return PT_MISS;
    *u_out = u;
    *v_out = v;
    return t;
}

#include "pt_mesh_bvh.hpp"

struct Hit {
    V3 p, n;
    float u, v;  // texture coordinates (mesh hits)
    uint32_t tex;
    uint32_t mat;
};

// :322-360 hitScene, :305-320 hitModel, :291-303 hitMeshOut.
// Nearest over spheres → planes → lenses → models with strict '<' (the earlier
// primitive keeps a tie); inside a mesh the FIRST front-facing hit in face
// order wins, not the nearest.
// ACCEL = false compiles the BVH walks out (scenes without a BVH keep the lean kernel).
// The nearest-hit search in three parts, so that a kernel can interleave the (long, divergent) mesh walks
// of part 2 with other work: 1 spheres / planes / lenses, 2 models, 3 counters + winner's record.
struct Nearest {
    float t = RT_MAX_DISTANCE;
    uint32_t id = PT_NO_HIT;
    uint32_t face = 0, mat = 0;
    float u = 0.0f, v = 0.0f;
};
// LENSES = false compiles the lens loop out (kernels specialised for scenes of spheres and planes only)
template <bool COUNT, bool ACCEL, bool LENSES = true>
PT_DEV void hit_primitives(const Ctx &c, const Ray &r, Nearest &nb) {
    const DeviceScene &sc = c.sc;
    float best_t = nb.t;
    uint32_t best_id = nb.id;

    // spheres: through the BVH when one was built, for lanes whose direction is (nearly) unit length
    bool brute = true;
    if (ACCEL && sc.bvh_node_count) {
        float dd = dot(r.d, r.d);
        brute = !(fabsf(dd - 1.0f) < 0.25f);  // also catches NaN; the margin m covers any smaller deviation
        if (!brute) hit_spheres_bvh<COUNT>(sc, r, best_t, best_id, c.cn);
    }
    // brute force: wave-uniform index → scalar loads; a batch of 4 in flight while 4 are tested
#ifndef PT_QSTAT
    if (COUNT && brute) c.cn->c[CN_DBG_BVH_TESTS] += sc.sphere_count;
#endif
    if (brute && sc.sphere_batches) {
        const float4 *sp = (PT_LDS_SPHERES && c.lsph) ? c.lsph : sc.sph4;
        float4 a0 = sp[0], a1 = sp[1], a2 = sp[2], a3 = sp[3];
        for (uint32_t b = 0; b < sc.sphere_batches; b++) {
            sp += PT_SPHERE_BATCH;  // the array ends with one dummy batch, so this prefetch is always in bounds
            float4 n0_ = sp[0], n1_ = sp[1], n2_ = sp[2], n3_ = sp[3];
            uint32_t i = b * PT_SPHERE_BATCH;
            float t0, t1, t2, t3;
            t0 = sphere_t(r, a0);
            t1 = sphere_t(r, a1);
            t2 = sphere_t(r, a2);
            t3 = sphere_t(r, a3);
            if (t0 < best_t) { best_t = t0; best_id = K_SPHERE | i; }
            if (t1 < best_t) { best_t = t1; best_id = K_SPHERE | (i + 1); }
            if (t2 < best_t) { best_t = t2; best_id = K_SPHERE | (i + 2); }
            if (t3 < best_t) { best_t = t3; best_id = K_SPHERE | (i + 3); }
            a0 = n0_; a1 = n1_; a2 = n2_; a3 = n3_;
        }
    }
    PT_STAMP(c, 2);
    for (uint32_t i = 0; i < sc.plane_count; i++) {
        const rt_plane &p = sc.planes[i];
        float t = plane_t(r, ld3(p.pos), ld3(p.normal));
        if (t < best_t) {
            best_t = t;
            best_id = K_PLANE | i;
        }
    }
    for (uint32_t i = 0; LENSES && i < sc.lens_count; i++) {
        int which;
        float t = lens_t(r, sc.lenses[i], &which);
        if (t < best_t) {
            best_t = t;
            best_id = K_LENS | i;
        }
    }
    nb.t = best_t;
    nb.id = best_id;
}

template <bool COUNT, bool ACCEL>
PT_DEV void hit_models(const Ctx &c, const Ray &r, Nearest &nb) {
    const DeviceScene &sc = c.sc;
    float best_t = nb.t;
    uint32_t best_id = nb.id;
    uint32_t best_face = nb.face, best_mat = nb.mat;
    float best_u = nb.u, best_v = nb.v;
    for (uint32_t mo = 0; mo < (PT_NO_MODELS ? 0u : sc.model_count); mo++) {
        const rt_model &model = sc.models[mo];
        float model_best = RT_MAX_DISTANCE;  // hitModel's own hit_min (:307)
        uint32_t m_id = PT_NO_HIT, m_face = 0;
        float m_u = 0.0f, m_v = 0.0f;
        for (uint32_t k = 0; k < model.mesh_count; k++) {
            uint32_t mi = model.mesh_anchor + k;
            const rt_mesh &mesh = sc.meshes[mi];
            if (COUNT) c.cn->c[CN_T_MESH]++;
            // hitMeshOut: scan faces until this lane has its first front-facing hit.  The face
            // records come through scalar loads (wave-uniform index), the next one in flight
            // while the current one is tested.
            bool found = false;
            float ft = 0.0f, fu = 0.0f, fv = 0.0f;
            uint32_t fface = 0;
            uint32_t root = (ACCEL && sc.mesh_bvh_root) ? sc.mesh_bvh_root[mi] : PT_MESH_BVH_NONE;
            if (ACCEL && root != PT_MESH_BVH_NONE) {
                // smallest face index with a valid front-facing hit, through the mesh's BVH
                uint32_t best = mesh.face_count;
                (void)mesh_bvh_walk<0>(sc, r, root, best, ft, fu, fv, COUNT ? c.cn : nullptr);
                found = best < mesh.face_count;
                fface = best;
                if (COUNT) {  // the reference's scan: faces 0..best, and the valid hits it stepped over
                    c.cn->c[CN_T_TRI] += found ? best + 1u : mesh.face_count;
                    float x0, x1, x2;
                    uint32_t lim = best;
                    c.cn->c[CN_H_TRI] += mesh_bvh_walk<1>(sc, r, root, lim, x0, x1, x2) + (found ? 1u : 0u);
                }
            }
            const uint32_t fbase = sc.mesh_face_base[mi];
            const float4 *fr = sc.faces + 3u * (size_t)fbase;
            bool scan = root == PT_MESH_BVH_NONE;
#if PT_FACE_MASK
            if (!COUNT && !ACCEL && scan && mesh.face_count <= 32u) {   // (instantiations that also carry the mesh walk keep the scan: registers)
                // Small mesh: the facing test first.  `dot(n, d) < 0` (:298) needs only the stored normal — one scalar
                // load and a dot product per face — and rules out every face on the far side of a closed mesh (half of
                // a cube's).  Each lane then runs hitTriangle on ITS OWN front-facing faces in ascending order and stops
                // at its first hit: the same face as the scan's (both conditions must hold, in either order), in half
                // the passes.  The records of a lane's face are gathered per lane (a few hundred bytes, cache resident).
                scan = false;
                uint32_t cand = 0;
                for (uint32_t f = 0; f < mesh.face_count; f++) {
                    const float4 n4 = fr[3u * f + 2u];
                    if (dot(mk(n4.y, n4.z, n4.w), r.d) < 0.0f) cand |= 1u << f;
                }
                while (cand) {
                    const uint32_t f = (uint32_t)__builtin_ctz(cand);
                    cand &= cand - 1u;
                    float4 g0, g1;
                    float e2z;
                    if (c.lfaces) {   // staged: three LDS reads (a per-lane gather from memory costs as much address-path time as it saves)
                        const uint32_t at = 3u * (fbase + f);
                        g0 = lds_ld(c.lfaces, at);
                        g1 = lds_ld(c.lfaces, at + 1u);
                        e2z = lds_ld(c.lfaces, at + 2u).x;
                    } else {
                        const float4 *q = fr + 3u * f;
                        g0 = q[0];
                        g1 = q[1];
                        e2z = q[2].x;
                    }
                    float u, v;
                    float t = triangle_t(r, mk(g0.x, g0.y, g0.z), mk(g0.w, g1.x, g1.y), mk(g1.z, g1.w, e2z), &u, &v);
                    if (t < PT_MISS) {
                        found = true;
                        ft = t; fu = u; fv = v; fface = f;
                        cand = 0;
                    }
                }
            }
#endif
            float4 q0 = fr[0], q1 = fr[1], q2 = fr[2];
            for (uint32_t f = 0; scan && f < mesh.face_count; f++) {
                fr += 3;  // the array ends with a dummy record, so this prefetch stays in bounds
                float4 p0 = fr[0], p1 = fr[1], p2 = fr[2];
                if (!found) {  // a lane that has its face idles while the others keep scanning
                    if (COUNT) c.cn->c[CN_T_TRI]++;
                    V3 A = mk(q0.x, q0.y, q0.z), e1 = mk(q0.w, q1.x, q1.y), e2 = mk(q1.z, q1.w, q2.x);
                    float u, v;
                    float t = triangle_t(r, A, e1, e2, &u, &v);
                    if (t < PT_MISS) {
                        if (COUNT) c.cn->c[CN_H_TRI]++;
                        V3 n = mk(q2.y, q2.z, q2.w);
                        if (dot(n, r.d) < 0.0f) {
                            found = true;
                            ft = t; fu = u; fv = v; fface = f;
                        }
                    }
                }
                q0 = p0; q1 = p1; q2 = p2;
            }
            if (found && ft < model_best) {
                model_best = ft;
                m_id = K_MESH | mi;
                m_face = fface;
                m_u = fu;
                m_v = fv;
            }
        }
        if (m_id != PT_NO_HIT && model_best < best_t) {
            best_t = model_best;
            best_id = m_id;
            best_face = m_face;
            best_mat = model.mat_ID;
            best_u = m_u;
            best_v = m_v;
        }
    }

    nb.t = best_t;
    nb.id = best_id;
    nb.face = best_face;
    nb.mat = best_mat;
    nb.u = best_u;
    nb.v = best_v;
}

// SIMPLE = true: the winner can only be a sphere or a plane (no lens / mesh code)
template <bool COUNT, bool SIMPLE = false>
PT_DEV bool hit_finish(const Ctx &c, const Ray &r, const Nearest &nb, Hit &hit) {
    const DeviceScene &sc = c.sc;
    const float best_t = nb.t;
    const uint32_t best_id = nb.id, best_face = nb.face, best_mat = nb.mat;
    const float best_u = nb.u, best_v = nb.v;
    if (COUNT) {
        c.cn->c[CN_BOUNCES]++;
        c.cn->c[CN_T_SPHERE] += sc.sphere_count;
        c.cn->c[CN_T_PLANE] += sc.plane_count;
        c.cn->c[CN_T_LENS] += sc.lens_count;
        c.cn->c[CN_T_MODEL] += sc.model_count;
    }
    PT_STAMP(c, 3);
    if (best_id == PT_NO_HIT) return false;

    // rebuild the winner's record with the reference's arithmetic
    uint32_t kind = best_id & K_MASK, idx = best_id & ~K_MASK;
    hit.p = point_at(r, best_t);
    hit.u = hit.v = 0.0f;
    hit.tex = 0;
    if (kind == K_PLANE) {
        V3 n;
        if (c.lpln) {
            float4 w = lds_ld(c.lpln, idx);
            n = xyz(w);
            hit.mat = __float_as_uint(w.w);
        } else {
            const rt_plane &p = sc.planes[idx];
            n = ld3(p.normal);
            hit.mat = p.mat_ID;
        }
        hit.n = neg(n) * sign1(dot(r.d, n));  // :187
    } else if (!SIMPLE && kind == K_MESH) {
        const rt_mesh &mesh = sc.meshes[idx];
        const uint32_t *ib = sc.indices + mesh.index_anchor + 3u * best_face;
        uint32_t ia = mesh.vertex_anchor + ib[0], ibx = mesh.vertex_anchor + ib[1], ic = mesh.vertex_anchor + ib[2];
        float4 fq = sc.faces[3u * ((size_t)sc.mesh_face_base[idx] + best_face) + 2];
        hit.n = mk(fq.y, fq.z, fq.w);  // normalize(cross(e1, e2)), :285, precomputed per face
        rt_float2 ua = sc.uvs[ia], ub = sc.uvs[ibx], uc = sc.uvs[ic];
        float wgt = 1.0f - best_u - best_v;  // :102
        hit.u = mad(uc.x, best_v, mad(ua.x, wgt, ub.x * best_u));   // (A·w + B·u) + C·v
        hit.v = mad(uc.y, best_v, mad(ua.y, wgt, ub.y * best_u));
        hit.tex = mesh.texture_ID;
        hit.mat = best_mat;
    } else {
        // sphere and lens share normal = (p - centre) / radius (:160, :248)
        V3 centre;
        float rad;
        if (SIMPLE || kind == K_SPHERE) {
            if (c.lwin) {
                float4 w = lds_ld(c.lwin, 2 * idx);
                centre = xyz(w);
                rad = w.w;
                hit.mat = __float_as_uint(lds_ld(c.lwin, 2 * idx + 1).x);
            } else {
                const rt_sphere &s = sc.spheres[idx];
                centre = ld3(s.pos);
                rad = s.r;
                hit.mat = s.mat_ID;
            }
        } else {
            const rt_lens &l = sc.lenses[idx];
            int which = 0;
            (void)lens_t(r, l, &which);
            centre = which == 0 ? ld3(l.p1) : ld3(l.p2);
            rad = which == 0 ? l.r1 : l.r2;
            hit.mat = l.mat_ID;
        }
        hit.n = div3(hit.p - centre, rad);
    }
    return true;
}


// :322-360
template <bool COUNT, bool ACCEL>
PT_DEV bool hit_scene(const Ctx &c, const Ray &r, Hit &hit) {
    Nearest nb;
    hit_primitives<COUNT, ACCEL>(c, r, nb);
    hit_models<COUNT, ACCEL>(c, r, nb);
    return hit_finish<COUNT>(c, r, nb, hit);
}

// ---- materials -------------------------------------------------------------------
// :401-405
PT_DEV float schlick(float cosine, float r0) { return mad(1.0f - r0, pow5(1.0f - cosine), r0); }   // r0 + (1 - r0) * pow(1 - cai, 5)

// :105-107 with the bilinear definition of DESIGN.md (OpenCL 1.2 §8.2, edge clamp)
PT_DEV int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
PT_DEV V3 texture_rgb(const DeviceScene &sc, float s, float t, uint32_t tex_id) {
    if (sc.tex_layers <= 0) return mk(0.0f, 0.0f, 0.0f);
    int W = sc.tex_w, H = sc.tex_h;
    int layer = tex_id < (uint32_t)sc.tex_layers ? (int)tex_id : 0;
    float u = s * sc.tex_wf - 0.5f, v = t * sc.tex_hf - 0.5f;
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

// One material interaction at a hit whose material is NOT a light
// (:454-479 with rayScatter :393, rayReflect :362, rayRefract :369,
// rayRefractDielectric :407).  `col` is the material colour, or the texel for
// t_textured.  All kinds end in the same tail: new origin = hit point, new
// direction = v, normalised unless it is a refraction (:386,:428 do not
// renormalise).
// set_origin = false: the caller has already moved the ray's origin to the hit point (the queue kernels do so
// right after the hit search, so that the hit point is not carried in registers of its own).
// The sample-independent terms of a glass interaction (:371-381, :409-424): facing normal, index ratio, cosine of the
// incident angle (made negative), Schlick's r0 for that side.  One definition, used by scatter() and by the shared
// decision tree of pt_prefix (which needs the reflect probability and the discriminant without taking a branch).
struct GlassTerms {
    V3 n;
    float ratio, cai, r0;
};
PT_DEV GlassTerms glass_terms(const Ctx &c, V3 d, V3 hn, uint32_t mat, float extra) {
    // 1/extra and Schlick's r0² per material: from the workgroup's LDS table when there is one
    float inv_extra, r0_extra, r0_inv;
    if (c.lmat) {
        float4 x = lds_ld(c.lmat, 2 * mat + 1);
        inv_extra = x.y; r0_extra = x.z; r0_inv = x.w;
    } else {
        inv_extra = 1.0f / extra;
        r0_extra = schlick_r0(extra);
        r0_inv = schlick_r0(inv_extra);
    }
    GlassTerms g;
    g.cai = dot(d, hn);  // cos of the incident angle
    if (g.cai > 0) {
        g.n = neg(hn);
        g.ratio = extra;
        g.r0 = r0_extra;
        g.cai = -g.cai;
    } else {
        g.n = hn;
        g.ratio = inv_extra;
        g.r0 = r0_inv;
    }
    return g;
}
PT_DEV float glass_disc(const GlassTerms &g) { return nmad(g.ratio * g.ratio, nmad(g.cai, g.cai, 1.0f), 1.0f); }   // 1 - ratio * ratio * (1 - cai * cai)   (:381,:424)

template <bool COUNT>
PT_DEV void scatter(const Ctx &c, Ray &r, V3 &out, const Hit &h, int type, float extra, V3 col, const Rnd &rnd,
                    bool set_origin = true) {
    V3 v;
    bool renorm = true;
    if (type == RT_DIFFUSE || type == RT_TEXTURED) {
        if (COUNT) c.cn->c[CN_N_SCATTER]++;
        v = h.n + rnd.v;
        out = out * extra;
    } else if (type == RT_REFLECTIVE) {
        float k = 2.0f * dot(r.d, h.n);
        v = nmad(h.n, k, r.d);   // dir - 2 dot(dir, n) * n
        out = out * extra;  // :366 — only for t_reflective
    } else if (type == RT_REFRACTIVE || type == RT_DIELECTRIC) {
        const GlassTerms g = glass_terms(c, r.d, h.n, h.mat, extra);
        const V3 n = g.n;
        const float ratio = g.ratio, cai = g.cai;
        bool want = true;
        if (type == RT_DIELECTRIC) {
            if (COUNT) c.cn->c[CN_N_DIELECTRIC]++;
            float prob = schlick(-cai, g.r0);
            want = prob < rnd.u;
        }
        float disc = glass_disc(g);
        if (want && disc > 0.0f) {
            // ratio * dir - n * (ratio * cai + sqrt(disc))   (:385,:428): the LEFT product is the one clang contracts
            V3 nk = n * mad(ratio, cai, sqrt1(disc));
            v = PT_CONTRACT ? mk(__builtin_fmaf(ratio, r.d.x, -nk.x), __builtin_fmaf(ratio, r.d.y, -nk.y), __builtin_fmaf(ratio, r.d.z, -nk.z))
                            : r.d * ratio - nk;
            renorm = false;
        } else {  // (total internal) reflection about the facing normal
            float k = 2.0f * dot(r.d, n);
            v = nmad(n, k, r.d);
        }
    } else {
        return;  // unknown type: the reference's switch has no default (rejected by rt_set_scene)
    }
    if (set_origin) r.o = h.p;
    if (renorm) v = normalize(v);
    r.d = v;
    out = vmin(out, col);  // mixCol is min(), :437
}

// :444-486 getCol from bounce i0 on — the sky is black, a path that survives
// DEPTH bounces returns what it has.
template <bool COUNT, bool ACCEL>
PT_DEV V3 trace_from(const Ctx &c, Ray r, V3 out, uint32_t i0, uint32_t sample, uint32_t gx, uint32_t gy) {
    for (uint32_t i = i0; i < RT_DEPTH; i++) {
        Rnd rnd = fetch_rnd(c.sc.table, r.d, i + sample, gx, gy);
        Hit h;
        if (!hit_scene<COUNT, ACCEL>(c, r, h)) return mk(0.0f, 0.0f, 0.0f);
        if (COUNT) c.cn->c[CN_H_BOUNCE]++;
        int type;
        float extra;
        V3 col;
        load_material(c, h.mat, type, extra, col);
        if (type == RT_LIGHT) return vmin(out, col);
        if (type == RT_TEXTURED) {
            if (COUNT) c.cn->c[CN_N_TEXFETCH]++;
            col = texture_rgb(c.sc, h.u, h.v, h.tex);
        }
        scatter<COUNT>(c, r, out, h, type, extra, col, rnd);
    }
    return out;
}

template <bool COUNT, bool ACCEL>
PT_DEV V3 radiance(const Ctx &c, Ray r, uint32_t sample, uint32_t gx, uint32_t gy) {
    return trace_from<COUNT, ACCEL>(c, r, mk(1.0f, 1.0f, 1.0f), 0, sample, gx, gy);
}

// :129-139, :500-505 — no pixel jitter
PT_DEV Ray primary_ray(const float *cam, uint32_t x, uint32_t y, int w, int h) {
    float s = (float)(int)x / (float)w;
    float t = (float)(int)y / (float)h;
    Ray r;
    r.o = mk(cam[0], cam[1], cam[2]);
    V3 llc = mk(cam[3], cam[4], cam[5]), hor = mk(cam[6], cam[7], cam[8]), ver = mk(cam[9], cam[10], cam[11]);
    r.d = normalize(mad(ver, t, mad(hor, s, llc)));   // llc + s * hor + t * ver
    return r;
}


// trace_branch: the deterministic stretch of a path from bounce i0 on (ray r, path colour out) up to its next random
// event — the record of that vertex (REC_VERTEX: a diffuse / textured / dielectric surface) or the path's final
// colour (REC_FINAL: sky, a light, or DEPTH bounces).  trace_prefix is the stretch that starts at the camera.
// max_stretch: after that many deterministic bounces (mirror / glass chains) the stretch is cut short at the next hit,
// which is returned as a vertex of whatever material it has — the samples take it from there (their kernels'
// interaction step handles every material).  pt_prefix follows its stretch to the end; the tree builder, whose
// launches last as long as their longest stretch, does not.
template <bool COUNT, bool ACCEL>
PT_DEV PixelRec trace_branch(const Ctx &c, Ray r, V3 out, uint32_t i0, uint32_t max_stretch = RT_DEPTH) {
    PixelRec rec;
    rec.p_kind = rec.n_extra = rec.d = rec.col = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t bounces = 0u;
    for (uint32_t i = i0; i < RT_DEPTH; i++) {
        Hit h;
        if (!hit_scene<COUNT, ACCEL>(c, r, h)) {
            out = mk(0.0f, 0.0f, 0.0f);
            break;
        }
        bounces = i - i0 + 1u;
        if (COUNT) c.cn->c[CN_H_BOUNCE]++;
        int type;
        float extra;
        V3 col;
        load_material(c, h.mat, type, extra, col);
        if (type == RT_LIGHT) {
            out = vmin(out, col);
            break;
        }
        if (type == RT_DIFFUSE || type == RT_TEXTURED || type == RT_DIELECTRIC || i - i0 >= max_stretch) {
            if (type == RT_TEXTURED) {
                if (COUNT) c.cn->c[CN_N_TEXFETCH]++;
                col = texture_rgb(c.sc, h.u, h.v, h.tex);
            }
            uint32_t bits = REC_VERTEX | (i << 8) | ((uint32_t)type << 16);
            rec.p_kind = make_float4(h.p.x, h.p.y, h.p.z, __uint_as_float(bits));
            rec.n_extra = make_float4(h.n.x, h.n.y, h.n.z, extra);
            rec.d = make_float4(r.d.x, r.d.y, r.d.z, __uint_as_float(h.mat));
            rec.out = make_float4(out.x, out.y, out.z, 0.0f);
            rec.col = make_float4(col.x, col.y, col.z, 0.0f);
            return rec;
        }
        Rnd none;  // mirror / glass: no random numbers
        none.v = mk(0.0f, 0.0f, 0.0f);
        none.u = 0.0f;
        scatter<COUNT>(c, r, out, h, type, extra, col, none);
    }
    rec.p_kind.w = __uint_as_float((uint32_t)REC_FINAL | (bounces << 8));   // (how many hits the stretch went through: pt_prefix's load estimate)
    rec.out = make_float4(out.x, out.y, out.z, 0.0f);
    return rec;
}
#ifndef PT_PREFIX_STRETCH
#define PT_PREFIX_STRETCH RT_DEPTH   // A/B: deterministic bounces pt_prefix follows before the samples take over (RT_DEPTH: to the end)
#endif
template <bool COUNT, bool ACCEL>
PT_DEV PixelRec trace_prefix(const Ctx &c, Ray r, uint32_t gx, uint32_t gy) {
    return trace_branch<COUNT, ACCEL>(c, r, mk(1.0f, 1.0f, 1.0f), 0u, PT_PREFIX_STRETCH);
}

// radiance of one sample continuing from its pixel's record
PT_DEV const float4 *tree_leaf(const PixelTree *tree, const float *__restrict__ table, uint32_t bu);
template <bool COUNT, bool ACCEL>
PT_DEV V3 radiance_from_rec(const Ctx &c, const PixelRec &rec0, uint32_t sample, uint32_t gx, uint32_t gy,
                            const PixelTree *trees = nullptr) {
    PixelRec rec = rec0;
    if ((__float_as_uint(rec.p_kind.w) & 0xFFu) == REC_TREE) {   // the pixel has a shared decision tree: this sample's leaf
        const float4 *lf = tree_leaf(trees + __float_as_uint(rec.col.w), c.sc.table, rnd_base_u(sample, gx, gy));
        rec.p_kind = lf[0]; rec.n_extra = lf[1]; rec.d = lf[2]; rec.out = lf[3]; rec.col = lf[4];
    }
    uint32_t bits = __float_as_uint(rec.p_kind.w);
    if ((bits & 0xFFu) == REC_FINAL) return xyz(rec.out);
    uint32_t depth = (bits >> 8) & 0xFFu;
    int type = (int)(bits >> 16);
    Ray r;
    r.o = xyz(rec.p_kind);
    r.d = xyz(rec.d);
    Hit h;
    h.p = xyz(rec.p_kind);
    h.n = xyz(rec.n_extra);
    h.u = h.v = 0.0f;
    h.tex = 0;
    h.mat = __float_as_uint(rec.d.w);
    V3 out = xyz(rec.out);
    Rnd rnd = fetch_rnd(c.sc.table, r.d, depth + sample, gx, gy);
    scatter<COUNT>(c, r, out, h, type, rec.n_extra.w, xyz(rec.col), rnd);
    return trace_from<COUNT, ACCEL>(c, r, out, depth + 1, sample, gx, gy);
}

// ---- shared decision tree (pt_types.hpp PixelTree) ------------------------------------------------------------------
#ifndef PT_TREE_STRETCH
#define PT_TREE_STRETCH 3u   // deterministic bounces a continuation is followed for before the samples take over (see trace_branch)
#endif
PT_DEV bool is_glass_vertex(const PixelRec &rec) {
    const uint32_t bits = __float_as_uint(rec.p_kind.w);
    return (bits & 0xFFu) == REC_VERTEX && (int)(bits >> 16) == RT_DIELECTRIC;
}
// One step of the builder for the glass vertex `rec`: what every sample does there regardless of its random number.
// A vertex whose refraction is impossible (discriminant <= 0: total internal reflection, `reflect_prob < rand &&
// discriminant > 0` is false for every rand, :426-433) is no decision — the path simply goes on to its next random
// event, and so on: on return `rec` is either a glass vertex that IS a decision (true; prob = its reflect probability)
// or some other record (false).
template <bool ACCEL>
PT_DEV bool tree_settle(const Ctx &c, PixelRec &rec, float &prob) {
    // (at most PT_TREE_STRETCH total reflections in a row are followed here; a glass vertex left unexamined after
    // that is simply a leaf the samples continue from)
    for (uint32_t k = 0; k <= PT_TREE_STRETCH && is_glass_vertex(rec); k++) {
        const GlassTerms g = glass_terms(c, xyz(rec.d), xyz(rec.n_extra), __float_as_uint(rec.d.w), rec.n_extra.w);
        if (glass_disc(g) > 0.0f) {
            prob = schlick(-g.cai, g.r0);
            return true;
        }
        Ray r;
        r.o = xyz(rec.p_kind);
        r.d = xyz(rec.d);
        Hit h;
        h.p = r.o;
        h.n = xyz(rec.n_extra);
        h.u = h.v = 0.0f;
        h.tex = 0;
        h.mat = __float_as_uint(rec.d.w);
        Rnd force;
        force.v = mk(0.0f, 0.0f, 0.0f);
        force.u = -INFINITY;
        V3 out = xyz(rec.out);
        scatter<false>(c, r, out, h, RT_DIELECTRIC, rec.n_extra.w, xyz(rec.col), force);
        rec = trace_branch<false, ACCEL>(c, r, out, ((__float_as_uint(rec.p_kind.w) >> 8) & 0xFFu) + 1u, PT_TREE_STRETCH);
    }
    return false;
}
// One continuation of the decision vertex `rec` — which = 0: the refracted ray, 1: the reflected one — up to its next
// random event, traced with the very scatter() a sample runs: the branch is forced through the comparison
// `prob < u` itself (u = +inf: refract if any u can; u = -inf: reflect).
template <bool ACCEL>
PT_DEV PixelRec tree_branch(const Ctx &c, const PixelRec &rec, uint32_t which) {
    Ray r;
    r.o = xyz(rec.p_kind);
    r.d = xyz(rec.d);
    Hit h;
    h.p = r.o;
    h.n = xyz(rec.n_extra);
    h.u = h.v = 0.0f;
    h.tex = 0;
    h.mat = __float_as_uint(rec.d.w);
    Rnd force;
    force.v = mk(0.0f, 0.0f, 0.0f);
    force.u = which ? -INFINITY : INFINITY;
    V3 out = xyz(rec.out);
    scatter<false>(c, r, out, h, RT_DIELECTRIC, rec.n_extra.w, xyz(rec.col), force);
    return trace_branch<false, ACCEL>(c, r, out, ((__float_as_uint(rec.p_kind.w) >> 8) & 0xFFu) + 1u, PT_TREE_STRETCH);
}

// The leaf a sample continues from: walk the decisions with the sample's own table entries (random(), :120-125:
// float 300000 + (hash + s_seed·2683 + gid0·3931 + gid1) mod 100000, s_seed = bounce index + sample; bu = the
// sample's part of that sum, rnd_base_u).  → pointer to the leaf's five float4.
PT_DEV const float4 *tree_leaf(const PixelTree *tree, const float *__restrict__ table, uint32_t bu) {
    uint32_t node = 1u, leaf = 0u;
#pragma unroll 1
    for (uint32_t guard = 0; guard < PT_TREE_LEVELS; guard++) {
        const uint4 d = *reinterpret_cast<const uint4 *>(&tree->dec[node]);
        const uint32_t iu = (d.x + bu + d.z * 2683u) % RT_RANDOM_BUFFER_SIZE;
        const float u = table[3 * RT_RANDOM_BUFFER_SIZE + iu];
        const uint32_t ch = (__uint_as_float(d.y) < u ? d.w : d.w >> 16) & 0xFFFFu;
        if (ch & 0x8000u) {
            leaf = ch & 0x7FFFu;
            break;
        }
        node = ch;
    }
    return reinterpret_cast<const float4 *>(&tree->leaf[leaf]);
}

}  // namespace PT_NS
