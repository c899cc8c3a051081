// mesh_bvh_build.hpp — host-side builder of the per-mesh BVHs that pt_mesh_bvh.hpp walks.
// Binned-SAH splits over face centroids, small leaves; children in adjacent pairs (left at
// an odd global index, every tree starts at an even index); per node: bounds, skip link (the node that
// follows the subtree in left-before-right order: the walk is threaded, pt_mesh_bvh.hpp), split axis,
// normal cone (axis, cos/sin of the half angle, widened by 1e-4 rad), smallest face index, longest
// edge and the smallest quality q = sin(angle between the two edges) · shape factor of the subtree.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#ifndef MESH_BVH_LEAF
#define MESH_BVH_LEAF 2  // faces per leaf (<= 7); A/B on C5: 1 → 249 ms, 2 → 227 ms, 4 → 280 ms
#endif

// A node's box is stored as centre and half extent (the slab tests then need no min / max: near = (c − o)·inv − (h + m)·|inv|):
// c = the rounded midpoint, h = the larger distance from c to the two bounds, rounded UP — [c − h, c + h] contains [lo, hi].
inline void bvh_centre_half(const float lo[3], const float hi[3], float c[3], float h[3]) {
    for (int k = 0; k < 3; k++) {
        c[k] = (float)(((double)lo[k] + (double)hi[k]) * 0.5);
        double hd = std::fmax((double)hi[k] - (double)c[k], (double)c[k] - (double)lo[k]);
        float hf = (float)hd;
        if ((double)hf < hd) hf = std::nextafter(hf, INFINITY);
        h[k] = std::nextafter(hf, INFINITY);   // one more ulp: c − h and c + h are themselves rounded on the device
    }
}

// binary16 bit pattern of a non-negative binary32, rounded up (to +inf above 65504; never a subnormal: at least 2^-14)
// or down (at most 65504; below 2^-14: 0)
inline uint32_t bvh_half_bits(float x, bool up) {
    if (!(x > 0.0f)) return up ? 0x0400u : 0u;          // (0, negative, NaN: the smallest normal, resp. 0)
    if (x >= 65504.0f) return (up && x > 65504.0f) ? 0x7C00u : 0x7BFFu;
    if (x < 6.103515625e-05f) return up ? 0x0400u : 0u;
    uint32_t u;
    memcpy(&u, &x, 4);
    uint32_t e = (u >> 23) - 127u + 15u, m = u & 0x7FFFFFu;
    uint32_t h = (e << 10) | (m >> 13);
    if (up && (m & 0x1FFFu)) h++;                       // (a carry runs into the exponent: the next binade, or inf)
    return h;
}
inline float bvh_half_value(uint32_t h) {
    uint32_t e = (h >> 10) & 31u, m = h & 1023u;
    if (e == 31u) return INFINITY;
    if (e == 0u) return std::ldexp((float)m, -24);
    return std::ldexp((float)(m | 1024u), (int)e - 25);
}
// builder's node (4 float4) → device node (3 float4): half extent, sin alpha and longest edge rounded up, q down
inline void mesh_node_pack(const float4 *nd, float4 *out) {
    out[0] = nd[0];
    out[1] = make_float4(nd[2].x, nd[2].y, nd[2].z, nd[1].w);
    const uint32_t p0 = bvh_half_bits(nd[1].x, true) | bvh_half_bits(nd[1].y, true) << 16;
    const uint32_t p1 = bvh_half_bits(nd[1].z, true) | bvh_half_bits(nd[3].x, true) << 16;
    const uint32_t p2 = bvh_half_bits(nd[3].z, true) | bvh_half_bits(nd[3].w, false) << 16;
    out[2] = nd[3];   // (.y = the smallest face index, kept as it is in .w below)
    out[2].w = nd[3].y;
    memcpy(&out[2].x, &p0, 4);
    memcpy(&out[2].y, &p1, 4);
    memcpy(&out[2].z, &p2, 4);
}

// Collapse of wide-cone inner nodes (device links only; the builder's tree stays as it is for the structural checks).
// A subtree whose normal cone is wide cannot be "steep" for any ray (pt_mesh_bvh.hpp): it can only be culled through
// the cap margin — about a unit of distance for C5's mesh — which the large boxes near the root next to never miss:
// 32 of the 97 nodes a C5 ray entered were such nodes, tested and entered by every ray.  A threaded tree lets them be
// SPLICED OUT: every link that points to a collapsed inner node points to its left child instead (whose skip link
// is the right child, whose skip link is the collapsed node's own) — the walk then tests the two children directly.
// Never a root: its box is the whole mesh's early-out.  Skipping a node test is always admissible (tests only cull).
#ifndef MESH_BVH_COLLAPSE_SIN
#define MESH_BVH_COLLAPSE_SIN 0.9f   // collapse inner nodes whose cone half angle has sin >= this (2 = never).  A/B on C5 at 1080p x 64 spp, bit-identical frames: never 21.69 ms, 0.9 → 20.83, 0.7 → 21.06, 0.5 → 25.89, 0.3 → 37.09 (the levels in between DO cull)
#endif
// nodes: 4 float4 per node as MeshBvhBuilder writes them; is_root[n] for the trees' roots.  Rewrites the A / B links in place.
inline void mesh_collapse_links(std::vector<float4> &nodes, const std::vector<uint8_t> &is_root) {
    const uint32_t n_nodes = (uint32_t)(nodes.size() / 4), END = 0x0FFFFFFFu;
    std::vector<uint8_t> gone(n_nodes, 0);
    std::vector<uint32_t> left(n_nodes, 0);
    for (uint32_t n = 0; n < n_nodes; n++) {
        uint32_t B;
        memcpy(&B, &nodes[4 * (size_t)n + 1].w, 4);
        const bool inner = !(B & 0x80000000u) && B != 0u;
        left[n] = B;
        gone[n] = inner && !is_root[n] && nodes[4 * (size_t)n + 3].x >= MESH_BVH_COLLAPSE_SIN;
    }
    auto resolve = [&](uint32_t n) {
        while (n != END && n < n_nodes && gone[n]) n = left[n];
        return n;
    };
    for (uint32_t n = 0; n < n_nodes; n++) {
        uint32_t A, B;
        memcpy(&A, &nodes[4 * (size_t)n].w, 4);
        memcpy(&B, &nodes[4 * (size_t)n + 1].w, 4);
        const uint32_t a2 = (A & ~END) | resolve(A & END);
        memcpy(&nodes[4 * (size_t)n].w, &a2, 4);
        if (!(B & 0x80000000u) && B != 0u) {
            const uint32_t b2 = resolve(B);
            memcpy(&nodes[4 * (size_t)n + 1].w, &b2, 4);
        }
    }
}

struct MeshBvhBuilder {
    // inputs: the mesh's face records (3 float4 per face: A, e1, e2, n as DeviceScene::faces)
    const float4 *rec = nullptr;
    uint32_t n_faces = 0;
    // outputs (appended to shared arrays)
    std::vector<float4> *nodes = nullptr;     // 4 per node
    std::vector<float4> *leaf_faces = nullptr;  // 3 per face
    std::vector<uint32_t> *leaf_idx = nullptr;

    std::vector<uint32_t> order;
    std::vector<float> lo, hi, cen;  // 3 per face
    std::vector<double> nrm;         // 3 per face (unit, or 0 for degenerate)
    std::vector<float> qual;         // sin(phi) · shape factor per face (0 = degenerate)
    std::vector<float> elen;         // longest of |e1|, |e2|, |e2 - e1| per face

    void prepare() {
        order.resize(n_faces);
        lo.resize(3 * n_faces); hi.resize(3 * n_faces); cen.resize(3 * n_faces);
        nrm.resize(3 * n_faces); qual.resize(n_faces); elen.resize(n_faces);
        for (uint32_t f = 0; f < n_faces; f++) {
            order[f] = f;
            const float4 &q0 = rec[3 * f], &q1 = rec[3 * f + 1], &q2 = rec[3 * f + 2];
            double A[3] = {q0.x, q0.y, q0.z}, e1[3] = {q0.w, q1.x, q1.y}, e2[3] = {q1.z, q1.w, q2.x};
            for (int k = 0; k < 3; k++) {
                double p0 = A[k], p1 = A[k] + e1[k], p2 = A[k] + e2[k];
                double l = std::min(p0, std::min(p1, p2)), h = std::max(p0, std::max(p1, p2));
                // the device forms B, C as A + e in exact terms only up to rounding: pad by 2 ulp of the magnitude
                double pad = 4e-7 * (std::fabs(l) + std::fabs(h)) + 1e-30;
                lo[3 * f + k] = (float)(l - pad);
                hi[3 * f + k] = (float)(h + pad);
                cen[3 * f + k] = (float)((p0 + p1 + p2) / 3.0);
            }
            double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            double cl = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
            double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
            double l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
            bool ok = std::isfinite(cl) && cl > 0 && l1 > 0 && l2 > 0;
            for (int k = 0; k < 3; k++) nrm[3 * f + k] = ok ? c[k] / cl : 0.0;
            double d3[3] = {e2[0] - e1[0], e2[1] - e1[1], e2[2] - e1[2]};
            double l3 = std::sqrt(d3[0] * d3[0] + d3[1] * d3[1] + d3[2] * d3[2]);
            // shape factor of the "steep" margin (pt_mesh_bvh.hpp): the corners of the region the reference
            // accepts lie within u·G·|s| / (sin(phi)·|cos(theta)|) of the face, G = 16.9 at A and
            // 8.25 + 8.66 (|e1|+|e3|)/|e2| resp. 8.66 + 8.25 (|e2|+|e3|)/|e1| at B and C (needles are worse);
            // q = sin(phi) · 25.6 / G  (25.6 = G of an equilateral face)
            double G = ok ? std::max(16.9, std::max(8.25 + 8.66 * (l1 + l3) / l2, 8.66 + 8.25 * (l2 + l3) / l1)) : 1.0;
            qual[f] = ok ? (float)(cl / (l1 * l2) * 25.6 / G) : 0.0f;
            elen[f] = (float)(std::max(l1, std::max(l2, l3)) * 1.0001);
        }
    }

    // skip: where the walk goes after this subtree.  depth: from level 32 on the split is the median, so the
    // recursion (and rt_debug_check_accel's) ends within 27 more levels for the < 2^26 faces the ABI admits
    void fill(uint32_t me, uint32_t skip, uint32_t b, uint32_t e, uint32_t depth = 0) {
        float nlo[3] = {INFINITY, INFINITY, INFINITY}, nhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        double ax[3] = {0, 0, 0};
        float q = INFINITY, emax = 0.0f;
        uint32_t min_face = 0xFFFFFFFFu;
        bool degenerate = false;
        for (uint32_t i = b; i < e; i++) {
            uint32_t f = order[i];
            for (int k = 0; k < 3; k++) {
                nlo[k] = std::fmin(nlo[k], lo[3 * f + k]); nhi[k] = std::fmax(nhi[k], hi[3 * f + k]);
                clo[k] = std::fmin(clo[k], cen[3 * f + k]); chi[k] = std::fmax(chi[k], cen[3 * f + k]);
                ax[k] += nrm[3 * f + k];
            }
            q = std::fmin(q, qual[f]);
            emax = std::fmax(emax, elen[f]);
            min_face = std::min(min_face, f);
            if (qual[f] <= 0.0f) degenerate = true;
        }
        // normal cone: axis = normalised sum, half angle = largest deviation (+1e-4 rad)
        double al = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        double cos_a = -1.0;
        if (al > 1e-12 && !degenerate) {
            for (int k = 0; k < 3; k++) ax[k] /= al;
            cos_a = 1.0;
            for (uint32_t i = b; i < e; i++) {
                uint32_t f = order[i];
                cos_a = std::min(cos_a, ax[0] * nrm[3 * f] + ax[1] * nrm[3 * f + 1] + ax[2] * nrm[3 * f + 2]);
            }
            double alpha = std::acos(std::max(-1.0, std::min(1.0, cos_a))) + 1e-4;
            cos_a = alpha >= M_PI ? -1.0 : std::cos(alpha);
        } else {
            ax[0] = 1; ax[1] = 0; ax[2] = 0;
        }
        double sin_a = std::sqrt(std::max(0.0, 1.0 - cos_a * cos_a));
        if (cos_a <= 0.0) { cos_a = 0.0; sin_a = 1.0; }  // cone wider than a hemisphere: always "grazing"
        if (degenerate) q = 0.0f;
        if (!std::isfinite(emax)) { emax = INFINITY; q = 0.0f; }

        uint32_t A = skip, B;
        if (e - b <= MESH_BVH_LEAF) {
            uint32_t first = (uint32_t)(leaf_faces->size() / 3);
            for (uint32_t i = b; i < e; i++) {
                uint32_t f = order[i];
                leaf_faces->push_back(rec[3 * f]);
                leaf_faces->push_back(rec[3 * f + 1]);
                leaf_faces->push_back(rec[3 * f + 2]);
                leaf_idx->push_back(f);
            }
            B = 0x80000000u | ((e - b) << 28) | first;
        } else {
            int axis = 0;
            for (int k = 1; k < 3; k++) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
            uint32_t mid = b + (e - b) / 2;
            constexpr int NB = 16;
            double best_cost = INFINITY;
            int best_ax = -1, best_bin = -1;
            for (int k = 0; k < 3 && depth < 32u; k++) {
                float ext = chi[k] - clo[k];
                if (!(ext > 0.0f)) continue;
                struct Bin { float lo[3], hi[3]; uint32_t n; } bins[NB];
                for (auto &bn : bins) { for (int t = 0; t < 3; t++) { bn.lo[t] = INFINITY; bn.hi[t] = -INFINITY; } bn.n = 0; }
                for (uint32_t i = b; i < e; i++) {
                    uint32_t f = order[i];
                    int bi = (int)((cen[3 * f + k] - clo[k]) / ext * NB);
                    bi = bi < 0 ? 0 : (bi >= NB ? NB - 1 : bi);
                    for (int t = 0; t < 3; t++) { bins[bi].lo[t] = std::fmin(bins[bi].lo[t], lo[3 * f + t]); bins[bi].hi[t] = std::fmax(bins[bi].hi[t], hi[3 * f + t]); }
                    bins[bi].n++;
                }
                auto area = [](const float *l, const float *h) {
                    double x = (double)h[0] - l[0], y = (double)h[1] - l[1], z = (double)h[2] - l[2];
                    return x < 0 ? 0.0 : 2.0 * (x * y + y * z + z * x);
                };
                double right_cost[NB];
                float rl[3] = {INFINITY, INFINITY, INFINITY}, rh[3] = {-INFINITY, -INFINITY, -INFINITY};
                uint32_t rn = 0;
                for (int j = NB - 1; j > 0; j--) {
                    for (int t = 0; t < 3; t++) { rl[t] = std::fmin(rl[t], bins[j].lo[t]); rh[t] = std::fmax(rh[t], bins[j].hi[t]); }
                    rn += bins[j].n;
                    right_cost[j] = rn ? area(rl, rh) * rn : INFINITY;
                }
                float ll[3] = {INFINITY, INFINITY, INFINITY}, lh[3] = {-INFINITY, -INFINITY, -INFINITY};
                uint32_t ln = 0;
                for (int j = 0; j < NB - 1; j++) {
                    for (int t = 0; t < 3; t++) { ll[t] = std::fmin(ll[t], bins[j].lo[t]); lh[t] = std::fmax(lh[t], bins[j].hi[t]); }
                    ln += bins[j].n;
                    if (ln == 0 || ln == e - b) continue;
                    double cost = area(ll, lh) * ln + right_cost[j + 1];
                    if (cost < best_cost) { best_cost = cost; best_ax = k; best_bin = j; }
                }
            }
            const float *cp = cen.data();
            if (best_ax >= 0) {
                axis = best_ax;
                float ext = chi[axis] - clo[axis], base = clo[axis];
                int bb = best_bin;
                auto it = std::partition(order.begin() + b, order.begin() + e, [cp, axis, ext, base, bb](uint32_t f) {
                    int bi = (int)((cp[3 * f + axis] - base) / ext * NB);
                    bi = bi < 0 ? 0 : (bi >= NB ? NB - 1 : bi);
                    return bi <= bb;
                });
                mid = (uint32_t)(it - order.begin());
            }
            if (best_ax < 0 || mid == b || mid == e) {
                mid = b + (e - b) / 2;
                std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e, [cp, axis](uint32_t i, uint32_t j) {
                    return cp[3 * i + axis] < cp[3 * j + axis] || (cp[3 * i + axis] == cp[3 * j + axis] && i < j);
                });
            }
            uint32_t left = (uint32_t)(nodes->size() / 4);  // the two children are adjacent
            nodes->resize(nodes->size() + 8);
            B = left;
            fill(left, left + 1u, b, mid, depth + 1u);
            fill(left + 1, skip, mid, e, depth + 1u);
        }
        float4 *nd = nodes->data() + 4 * (size_t)me;
        float bc[3], bh[3];
        bvh_centre_half(nlo, nhi, bc, bh);
        nd[0] = make_float4(bc[0], bc[1], bc[2], 0.0f);
        nd[1] = make_float4(bh[0], bh[1], bh[2], 0.0f);
        nd[2] = make_float4((float)ax[0], (float)ax[1], (float)ax[2], (float)cos_a);
        nd[3] = make_float4((float)sin_a, 0.0f, emax, q);
        memcpy(&nd[0].w, &A, 4);
        memcpy(&nd[1].w, &B, 4);
        memcpy(&nd[3].y, &min_face, 4);
    }

    // → global index of the tree's root
    uint32_t build() {
        prepare();
        if (!((nodes->size() / 4) & 1u)) nodes->resize(nodes->size() + 4, make_float4(0, 0, 0, 0));  // odd start
        uint32_t root = (uint32_t)(nodes->size() / 4);
        nodes->resize(nodes->size() + 4);  // the root is odd, so the child pairs that follow are (even, odd): one 128-byte line
        fill(root, 0x0FFFFFFFu, 0, n_faces);
        return root;
    }
};
