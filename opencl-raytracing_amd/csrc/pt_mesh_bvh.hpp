// pt_mesh_bvh.hpp — conservative BVH for the reference's mesh rule (device side).
//
// hitMeshOut (kernels/raytracer.cl:291-303) scans the faces of a mesh in index order
// and returns the FIRST face whose hitTriangle succeeds and whose normal faces the
// ray — not the nearest.  The same answer is "the smallest face index among valid
// front-facing hits", which a spatial hierarchy can find without scanning 50 000
// faces — provided it never drops a face the reference would have accepted.
//
// Why culling is safe (DESIGN.md "mesh BVH").  hitTriangle is Möller–Trumbore in
// binary32 (:257-289), every operation rounded once (no FMA), u = 2^-24.  Forward error
// analysis of its three numerators against the exact values for the stored vertices:
//     |δa|  <= 8.25 u |d| |e1| |e2|        a  = e1 · (d × e2)
//     |δnu| <= 8.25 u |s| |d| |e2|         nu = s · (d × e2),  s = origin − A
//     |δnv| <= 8.66 u |s| |d| |e1|         nv = d · (s × e1)
// A face the reference accepts therefore has exact barycentrics (of the point where the
// ray's LINE meets the face's plane; equivalently of the line seen along d) with
//     l_B >= −|δnu|/|a|,  l_C >= −|δnv|/|a|,  l_A >= −(|δnu| + |δnv| + |δa|)/|a| − 4u.
// Three margins follow, all used below with |s| <= dfar <= reach = dfar + 2·emax (dfar = largest
// distance origin → node box, emax = longest edge of the subtree):
//  steep   |a| = |d| |e1| |e2| sin(phi) |cos(theta)|: the accepted region is the face grown by at most
//          u·G·|s| / (sin(phi)·|cos(theta)|), G a shape factor (25.6 equilateral; mesh_bvh_build.hpp).
//          m_steep = K·reach / (q·cosmin), q = min sin(phi)·25.6/G over the subtree, cosmin = the
//          smallest |cos(theta)| over the subtree's NORMAL CONE; K = 5e-6 = 3.3 × 25.6 u.  Only when
//          cosmin > TAU and q > 1e-3 (then |δa| << |a| and the sign of a is right).
//  cap     the reference rejects |a_computed| < 1e-7 (TRIANGLE_EPSILON), so for emax²·|d| <= 0.04
//          (|δa| <= 8.25u·0.04 = 0.197e-7) every accepted face has |a| >= a_min = 0.8033e-7 whatever the angle.
//          Every vertex lies in the node's box, so |s| <= dfar; with u/a_min = 0.742:
//              -l_B <= 8.25u|s||d||e2|/a_min <= 6.12·|d|·dfar·emax,      -l_C <= 6.43·|d|·dfar·emax,
//              -l_A <= (6.12 + 6.43)·|d|·dfar·emax + 8.25u|d|emax²/a_min + 5u = 12.55·|d|·dfar·emax + 6.12·|d|·emax² + 5u.
//          The barycentrics sum to 1, so at most two are negative and their negative parts sum to
//          S <= |d|·emax·(18.98·dfar + 6.12·emax) (+ 5u, covered by the additive slack of the slab test); writing the
//          point as a combination of the vertices, it lies within S·(longest edge) of the face.
//          m_cap = |d|·emax²·(PT_MESH_CAP_D·dfar + PT_MESH_CAP_E·emax) with the two bounds + 9 % (20.7, 6.7); dlen, dfar
//          and emax are themselves rounded up (1e-6, 1e-3, 1e-4 + the binary16 rounding).  Round 1 carried
//          40·|d|·(dfar + 2·emax)·emax² here — a safety factor of 1.8 (and three times that for the nodes next to a
//          ray's origin) that cost C5 more than a third of its time: 39.1 ms → 27.2 (24·(dfar + 2·emax)) → 24.9 (this form)
//          at 1080p × 64 spp.
//  slab    same N, but along a direction x perpendicular to d the line leaves the face's own extent
//          along x by at most N·(that extent): grazing faces are thin along the part of the cone axis
//          perpendicular to d, which removes the "silhouette band" (see the code).
// A node is skipped when the line misses its box inflated by min(m_steep, m_cap) or fails the slab
// test; with neither steep nor cap it is only index-pruned.  Candidates are tested with the very
// same triangle_t() and the very same precomputed normal as the brute-force scan, so a visited
// face gives the reference's verdict bit for bit.
// tests/test_gpu_properties.py::test_mesh_bvh_keeps_faces_accepted_far_from_the_ray drives rays the
// reference accepts 2–5 edge lengths away from the face through this walk.
//
// Builder's node = 4 float4: (box centre.xyz, A) (box half extent.xyz, B) (cone axis.xyz, cos alpha) (sin alpha, min face, longest edge, q)
//   A = skip link (the node after this subtree; PT_MESH_END = none);  B = left child, or leaf: 0x80000000 | count << 28 | first face slot
// Device node = 3 float4 (mesh_node_pack in mesh_bvh_build.hpp): (centre.xyz, A) (cone axis.xyz, B)
//   (half extent.x | .y << 16, half extent.z | sin alpha << 16, longest edge | q << 16, min face) — the pairs are binary16.
// (included inside namespace pt by pt_device.hpp, after triangle_t and DeviceScene)
#pragma once

#define PT_MESH_TAU 2.0e-3f
#define PT_MESH_K 5.0e-6f
#ifndef PT_MESH_CAP_D
#define PT_MESH_CAP_D 20.7f
#endif
#ifndef PT_MESH_CAP_E
#define PT_MESH_CAP_E 6.7f
#endif
#ifndef PT_MESH_BACKFACE_CULL
#define PT_MESH_BACKFACE_CULL 1
#endif
#ifndef PT_MESH_SLAB
#define PT_MESH_SLAB 1  // extra slab test along the projected cone axis for grazing, narrow-cone subtrees
#endif
#ifndef PT_MESH_SLAB_SCALE
#define PT_MESH_SLAB_SCALE 1.0f  // test hook: < 1 must break tests/test_gpu_properties.py::test_mesh_bvh_grazing_rays
#endif
#ifndef PT_MESH_SLAB_SIN
#define PT_MESH_SLAB_SIN 0.9f  // A/B on C5 (cap 24): 0.3 -> 27.9 ms, 0.5 -> 27.4, 0.7 -> 27.1, 0.9 -> 27.0
#endif
#ifndef PT_MESH_LEAF_EVERY
#define PT_MESH_LEAF_EVERY 8u  // node steps between leaf phases of the mesh walk (power of two; 1 = faces tested in place).  A/B on C5 1080p x 64 spp: 1 -> 70.1 ms, 2 -> 68.1, 4 -> 68.2, 8 -> 67.5
#endif

// MODE 0: search — smallest face index < best_face with a valid front-facing hit.
// MODE 1: count  — number of faces with index < best_face whose hitTriangle succeeds (any
//                  facing): the reference's H_tri counter needs the back-facing hits it
//                  stepped over before its first front-facing one.
// The walk keeps neither a stack nor a way back up: the tree is THREADED — every node carries the index of the node
// that follows its subtree in the fixed left-before-right order (its "skip" link: the right sibling for a left child,
// the parent's skip for a right child, PT_MESH_END for the root), so "next" is the left child after a hit on an inner
// node and the skip link otherwise.  The rule wants the SMALLEST INDEX among the accepted faces, not the nearest, so
// the order in which subtrees are visited changes nothing but how early `best_face` can prune by index (1 % of the
// node tests on C5) — an ordered near-first walk paid for its order with a parent chase on the way back up.
// MeshWalk is the walk's whole position; mesh_bvh_steps advances it by at most max_steps nodes and returns true once
// the tree is exhausted — a kernel can interleave walks of different lengths with other work (pt_samples_w).
#ifdef PT_WSTAT  // diagnostic build (tools/wstat.py): wave-level lane census of the walk, never timed
struct WalkStat { unsigned long long steps, node_lanes, leaf_runs, leaf_lanes, idle_lanes; };
#define PT_WSTAT_ARG , WalkStat *ws = nullptr
#else
#define PT_WSTAT_ARG
#endif

#define PT_MESH_END 0x0FFFFFFFu
#define PT_MESH_PARKED 0x80000000u
PT_DEV float half_lo(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xFFFFu)); }
PT_DEV float half_hi(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }
struct MeshWalk {
    uint32_t cur;   // the node to be tested next | PT_MESH_PARKED: met leaf whose faces are still to be tested;  PT_MESH_END: through
};
PT_DEV MeshWalk mesh_walk_start(uint32_t root) { return MeshWalk{root}; }

// Per-ray constants of the CULLING tests only (never of a face test): hardware sqrt / rcp (1 ulp) instead of the
// correctly rounded expansions (54 / 43 issue cycles each, profiles/r02_valu_microbench.md) — every use below
// carries a relative slack of 1e-5 or more, and dlen is rounded UP by 2^-20 where a larger value is the safe side.
struct MeshCull {
    float dlen, o_max;
    V3 dh, inv;
};
PT_DEV MeshCull mesh_cull_setup(const Ray &r) {
    MeshCull k;
    k.dlen = __builtin_amdgcn_sqrtf(dot(r.d, r.d)) * 1.000001f;
    k.dh = r.d * __builtin_amdgcn_rcpf(k.dlen);
    k.inv = cull_inverse(r.d);   // (the slab test keeps the subtract form here: three more live registers for -o·inv would spill)
    k.o_max = fmaxf(fmaxf(fabsf(r.o.x), fabsf(r.o.y)), fabsf(r.o.z));
    return k;
}

// May the subtree of this node be skipped?  (a, b, cn, ex = the node's four float4; see the proof sketch above)
template <int MODE>
PT_DEV bool mesh_node_miss(const Ray &r, const MeshCull &k, float4 a, float4 b, float4 cn, float4 ex, uint32_t best_face,
                           LaneCounters *dbg) {
    const float dlen = k.dlen, o_max = k.o_max;
    const V3 dh = k.dh, inv = k.inv;
    if (dbg) dbg->c[CN_DBG_BVH_NODES]++;
    bool miss = __float_as_uint(ex.y) >= best_face;  // every face below has a larger index than the best
    if (!miss) {
        // smallest |cos(theta)| over the subtree's normal cone
        float x = dot(dh, xyz(cn));
        float sb = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - x * x));
        float cosmin = fabsf(x) * cn.w - sb * ex.x - 1.0e-5f;
        bool steep = cosmin > PT_MESH_TAU && ex.w > 1.0e-3f;    // no face of the subtree can be grazed
        // search mode wants FRONT-facing hits only (dot(n, d) < 0, :298): a subtree whose whole normal
        // cone points along the ray (n·d̂ >= cos(psi + alpha) > 0, far above the rounding of the
        // reference's own dot product) holds back faces only — the exit side of a closed mesh
#ifdef PT_MESH_STAT  // diagnostic: which kind of node is entered (read through the face-test debug counter)
        if (dbg) {
            bool wide = ex.x >= PT_MESH_SLAB_SIN;
            int kind = wide ? 1 : (steep ? 3 : 2);
            if (kind == PT_MESH_STAT) dbg->c[CN_DBG_BVH_TESTS]++;
        }
#endif
        bool backside = MODE == 0 && PT_MESH_BACKFACE_CULL && x * cn.w - sb * ex.x > 1.0e-3f;
        if (backside) miss = true;
        bool capped = ex.z * ex.z * dlen <= 0.04f;              // the epsilon test bounds the displacement
        if (!backside && (steep || capped)) {
            // (a = the box's centre, b = its half extent: the farthest corner, and below the slab distances, without min / max)
            const V3 dc = mk(a.x - r.o.x, a.y - r.o.y, a.z - r.o.z);
            float fx = fabsf(dc.x) + b.x, fy = fabsf(dc.y) + b.y, fz = fabsf(dc.z) + b.z;
            float dfar = __builtin_amdgcn_sqrtf(__builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx))) * 1.001f;
            float reach = __builtin_fmaf(2.0f, ex.z, dfar);
            // (a margin: the reciprocal is the hardware's, scaled up by 2^-19 — K itself carries a factor 3.3 of slack)
            float m_steep = steep ? PT_MESH_K * reach * (__builtin_amdgcn_rcpf(ex.w * cosmin) * 1.000002f) : INFINITY;
            float m_cap = capped ? dlen * ex.z * ex.z * __builtin_fmaf(PT_MESH_CAP_D, dfar, PT_MESH_CAP_E * ex.z) : INFINITY;
            float m = fminf(m_steep, m_cap) + __builtin_fmaf(1.0e-5f, dfar + o_max, 1.0e-6f);
            // per axis the line is inside the inflated slab for t in [tc − te, tc + te], tc = (c − o)/d, te = (h + m)/|d|
            // (the rounding of this form, 2^-24·(3·dfar + o_max) in units of distance, is 50 × below the slack in m)
            float tcx = dc.x * inv.x, tcy = dc.y * inv.y, tcz = dc.z * inv.z;
            float tex = (b.x + m) * fabsf(inv.x), tey = (b.y + m) * fabsf(inv.y), tez = (b.z + m) * fabsf(inv.z);
            float tmin = fmaxf(fmaxf(tcx - tex, tcy - tey), tcz - tez), tmax = fminf(fminf(tcx + tex, tcy + tey), tcz + tez);
            // the LINE misses the inflated box, or the box lies wholly behind / beyond the range of t
            miss = tmin > __builtin_fmaf(fabsf(tmax), 1.0e-5f, tmax) + 1.0e-4f || tmax < -(m + 1.0f) ||
                   tmin > RT_MAX_DISTANCE * 1.001f + m + 1.0f;
            if (PT_MESH_SLAB && !miss && !steep && ex.x < PT_MESH_SLAB_SIN && sb > 0.5f && ex.w > 1.0e-3f) {
                // Grazing subtree with a narrow cone: the displacement m_cap is real, but only ALONG the
                // sliver a grazing face projects to.  With barycentrics l_k >= -eta_k (eta_total =
                // m_cap / emax) the line's coordinate along any x perpendicular to d leaves the face's
                // own extent along x by at most eta_total * (that extent).  Along n' = the part of the
                // cone axis perpendicular to d every face of the subtree is thin: an edge e is
                // perpendicular to its normal, so |e . n'| <= |e| (sin alpha + |cos psi|) / sin psi.
                V3 np = (xyz(cn) - dh * x) * __builtin_amdgcn_rcpf(sb);   // |x| <= sin(alpha) + tau here (sb > 0.5)
                float rn = fabsf(np.x) * b.x + fabsf(np.y) * b.y + fabsf(np.z) * b.z;
                float dist = dc.x * np.x + dc.y * np.y + dc.z * np.z;   // (its sign is not needed)
                float mn = PT_MESH_SLAB_SCALE * m_cap * (ex.x + fabsf(x) + 2.0e-4f) * (__builtin_amdgcn_rcpf(sb) * 1.000002f) +
                           1.0e-5f * (dfar + o_max) + 1.0e-5f;
                if (fabsf(dist) > rn * 1.0001f + mn) miss = true;
            }
        }
    }
    return miss;
}

template <int MODE>
PT_DEV bool mesh_bvh_steps(const DeviceScene &sc, const Ray &r, MeshWalk &w, uint32_t &best_face, float &ft, float &fu,
                           float &fv, uint32_t max_steps, uint32_t &hits, LaneCounters *dbg = nullptr PT_WSTAT_ARG) {
    const MeshCull k = mesh_cull_setup(r);
    uint32_t cur = w.cur;
    // "While-while": a node per step for every lane that is neither parked nor through; the face tests of the lanes
    // parked at a leaf only every PT_MESH_LEAF_EVERY steps, at the end of the slice, or when every lane is parked or
    // through (run in place, the two Möller–Trumbore tests with their division were executed on about every second
    // step for the one or two lanes that had just reached a leaf).
    for (uint32_t guard = 0; guard < max_steps; guard++) {   // every node is tested at most once
#ifdef PT_WSTAT
        if (ws) {
            ws->steps++;
            ws->node_lanes += __popcll(__ballot(cur < PT_MESH_END));
            ws->idle_lanes += __popcll(__ballot(cur == PT_MESH_END));
        }
#endif
        if (cur < PT_MESH_END) {
            // THREE 16-byte loads per node (the walk is bound by the number of vector memory instructions as much as by
            // its ALU work): the box's half extent, sin alpha, the longest edge and q travel as binary16, rounded to the
            // safe side by the host (mesh_node_pack); cos alpha is derived from sin alpha.
            const float4 *nd = at32(sc.mbvh_nodes, cur * 48u);
            float4 a = nd[0], cn = nd[1], pk = nd[2];
            // (pinned: left to itself the compiler splits the loads and fetches some words only after a branch)
            asm("" : "+v"(a.w), "+v"(cn.w), "+v"(pk.x), "+v"(pk.w));
            const uint32_t B = __float_as_uint(cn.w);
            const uint32_t p0 = __float_as_uint(pk.x), p1 = __float_as_uint(pk.y), p2 = __float_as_uint(pk.z);
            const float sin_a = half_hi(p1);
            float4 b = make_float4(half_lo(p0), half_hi(p0), half_lo(p1), cn.w);
            float4 ex = make_float4(sin_a, pk.w, half_lo(p2), half_hi(p2));
            cn.w = __builtin_amdgcn_sqrtf(fmaxf(0.0f, __builtin_fmaf(-sin_a, sin_a, 1.0f)));
            if (mesh_node_miss<MODE>(r, k, a, b, cn, ex, best_face, dbg)) cur = __float_as_uint(a.w) & PT_MESH_END;
            else if (B & 0x80000000u) cur |= PT_MESH_PARKED;
            else cur = B;
        }
        // ---- leaf phase (wave-uniform decision)
        const bool parked = (cur & PT_MESH_PARKED) != 0;
        const bool flush = (guard & (PT_MESH_LEAF_EVERY - 1u)) == PT_MESH_LEAF_EVERY - 1u || guard + 1u == max_steps ||
                           __all(cur >= PT_MESH_END);
        if (flush && __any(parked)) {
#ifdef PT_WSTAT
            if (ws) { ws->leaf_runs++; ws->leaf_lanes += __popcll(__ballot(parked)); }
#endif
            if (parked) {
                const float4 *nd = at32(sc.mbvh_nodes, (cur & PT_MESH_END) * 48u);
                uint32_t A = __float_as_uint(nd[0].w), B = __float_as_uint(nd[1].w);
                uint32_t first = B & 0x0FFFFFFFu, cnt = (B >> 28) & 7u;
                for (uint32_t j = 0; j < cnt; j++) {
                    // (index and record fetched together — one trip to memory per face, not two)
                    uint32_t idx = *at32(sc.mbvh_face_idx, (first + j) << 2);
                    const float4 *fq = at32(sc.mbvh_faces, (first + j) * 48u);
                    float4 q0 = fq[0], q1 = fq[1], q2 = fq[2];
                    asm("" : "+v"(idx), "+v"(q0.x), "+v"(q1.x), "+v"(q2.x));
                    if (idx >= best_face) continue;
#ifndef PT_MESH_STAT
                    if (dbg) dbg->c[CN_DBG_BVH_TESTS]++;
#endif
                    float u, v;
                    float t = triangle_t(r, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x), &u, &v);
                    if (t < PT_MISS) {
                        if (MODE == 1) hits++;
                        else if (dot(mk(q2.y, q2.z, q2.w), r.d) < 0.0f) {
                            best_face = idx;
                            ft = t; fu = u; fv = v;
                        }
                    }
                }
                cur = A & PT_MESH_END;
            }
        }
        if (__all(cur == PT_MESH_END)) break;
    }
    w.cur = cur;
    return cur == PT_MESH_END;
}

// The first step of a search on its own (pt_samples_w tests the root while the lane is still in its cheap state: a ray
// that misses the whole mesh never joins a walk slice).  Same node test, same outcome as the first step of mesh_bvh_steps.
PT_DEV MeshWalk mesh_walk_first(const DeviceScene &sc, const Ray &r, uint32_t root, uint32_t best_face) {
    const MeshCull k = mesh_cull_setup(r);
    const float4 *nd = at32(sc.mbvh_nodes, root * 48u);
    float4 a = nd[0], cn = nd[1], pk = nd[2];
    const uint32_t B = __float_as_uint(cn.w);
    const uint32_t p0 = __float_as_uint(pk.x), p1 = __float_as_uint(pk.y), p2 = __float_as_uint(pk.z);
    const float sin_a = half_hi(p1);
    float4 b = make_float4(half_lo(p0), half_hi(p0), half_lo(p1), cn.w);
    float4 ex = make_float4(sin_a, pk.w, half_lo(p2), half_hi(p2));
    cn.w = __builtin_amdgcn_sqrtf(fmaxf(0.0f, __builtin_fmaf(-sin_a, sin_a, 1.0f)));
    if (mesh_node_miss<0>(r, k, a, b, cn, ex, best_face, nullptr)) return MeshWalk{__float_as_uint(a.w) & PT_MESH_END};
    return MeshWalk{(B & 0x80000000u) ? (root | PT_MESH_PARKED) : B};
}

template <int MODE>
PT_DEV uint32_t mesh_bvh_walk(const DeviceScene &sc, const Ray &r, uint32_t root, uint32_t &best_face, float &ft,
                              float &fu, float &fv, LaneCounters *dbg = nullptr) {
    MeshWalk w = mesh_walk_start(root);
    uint32_t hits = 0;
    if (!mesh_bvh_steps<MODE>(sc, r, w, best_face, ft, fu, fv, 0x7FFFFFFFu, hits, dbg))
        atomicOr(sc.walk_overflow, PT_OVF_MESH_WALK);   // cold: the tree has < 2^28 nodes, each tested at most once
    return hits;
}
