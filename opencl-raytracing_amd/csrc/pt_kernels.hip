// pt_kernels.hip — the path-tracing kernels and their launchers, compiled ONCE PER ARITHMETIC POLICY
// (-DPT_ARITH=0|1|2, see pt_arith.hpp) into namespaces pt_a0 / pt_a1 / pt_a2 of the same librt_amd.so.
// rt_amd.hip picks a policy's KernelSet at run time (rt_set_option(RT_OPT_ARITH, ...)).
//
// Build (see __graft_entry__.build_hip()), gfx950 only:
//   policy 0: hipcc -c -DPT_ARITH=0 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt ...
//   policy 1: hipcc -c -DPT_ARITH=1 -ffp-contract=off -fno-hip-fp32-correctly-rounded-divide-sqrt ...
//   policy 2: hipcc -c -DPT_ARITH=2 -ffp-contract=off -fno-hip-fp32-correctly-rounded-divide-sqrt ...
// (the contractions of policy 2 are written out as fma calls at the reference's sites; the compiler never contracts)
#include <hip/hip_runtime.h>

#include <algorithm>

#include "pt_device.hpp"
#include "rt_context.hpp"
#include "pt_kernels.hpp"

namespace PT_NS {
using namespace rtamd;

// =============================== device kernels ===============================

// wave (or pixel group) `unit` of a sample kernel → its first entry in the live list and how many of `want` exist.
// The list has two parts (pt_prefix): the HEAVY pixels, stored from the end of the capacity downwards, and the others,
// stored from 0 upwards in the order in which pt_prefix's workgroups finish.  The heavy ones are taken FIRST (longest
// processing time first — see pt_prefix), then the rest in list order.
// Safe by construction, whatever the grid: the counts are clamped to the list's capacity, a unit's start is formed
// in 64 bits and clamped INTO its part (start <= count), so `count - start` cannot wrap and `first + result` never
// leaves the part — a unit beyond the list gets 0 entries.  (Round 2's form `start < cnt ? min(want, cnt - start) : 0`
// is the same function, but an experiment that inlined it into a persistent loop faulted and the cause was never
// established beyond "the guard was optimised away"; this form has no guard to lose.  tests/test_gpu_properties.py
// ::test_live_list_far_shorter_than_the_grid renders an all-sky frame and a frame with ONE live pixel.)
PT_DEV uint32_t live_take(const FrameParams &fp, const uint32_t *__restrict__ live_count, uint32_t unit, uint32_t want,
                          uint32_t &first) {
    const uint32_t cap = fp.seg_cap;
    const uint32_t cnt_l = min(live_count[0], cap);
    const uint32_t cnt_h = min(live_count[LIVE_HEAVY_COUNTER], cap - cnt_l);
    const uint32_t units_h = want ? (cnt_h + want - 1u) / want : 0u;
    const bool heavy = unit < units_h;
    const uint32_t cnt = heavy ? cnt_h : cnt_l;
    const unsigned long long start64 = (unsigned long long)(heavy ? unit : unit - units_h) * want;
    const uint32_t start = (uint32_t)min(start64, (unsigned long long)cnt);
    first = (heavy ? cap - cnt_h : 0u) + start;
#ifdef PT_EXP_SKIP   // timing experiment (wrong image): leave out the last (1) / first (2) 3 % of the light list's chunks
    {
        const uint32_t nl = want ? (cnt_l + want - 1u) / want : 0u, c = unit - units_h;
        if (!heavy && PT_EXP_SKIP == 1 && c >= nl - nl / 32u) return 0u;
        if (!heavy && PT_EXP_SKIP == 2 && c < nl / 32u) return 0u;
        if (!heavy && PT_EXP_SKIP == 3 && c >= nl / 2u && c < nl / 2u + nl / 32u) return 0u;
    }
#endif
    return min(want, cnt - start);
}

template <bool COUNT>
PT_DEV void flush_counters(const LaneCounters &cn, unsigned long long *counters, uint32_t scale) {
    if (!COUNT) return;
    unsigned long long *row = counters + (size_t)(blockIdx.x % COUNTER_REPLICAS) * COUNTER_STRIDE;
#pragma unroll
    for (int i = 0; i < PT_N_COUNTERS; i++) {
        uint32_t v = cn.c[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&row[i], (unsigned long long)v * scale);
    }
}

PT_DEV void zero_counters(LaneCounters &cn) {
#pragma unroll
    for (int i = 0; i < PT_N_COUNTERS; i++) cn.c[i] = 0;
}

// how many set bits of a wave mask belong to lanes below this one (v_mbcnt: no per-lane mask to keep in registers)
PT_DEV uint32_t lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// xor-butterfly over the g lanes of a pixel group: a fixed summation tree
PT_DEV V3 group_sum(V3 sum, uint32_t g) {
    for (uint32_t off = g >> 1; off > 0; off >>= 1) {
        sum.x += __shfl_xor(sum.x, off);
        sum.y += __shfl_xor(sum.y, off);
        sum.z += __shfl_xor(sum.z, off);
    }
    return sum;
}

PT_DEV void accumulate(float4 *__restrict__ accum, size_t pix, V3 sum, uint32_t count) {
    float4 a = accum[pix];
    a.x += sum.x;
    a.y += sum.y;
    a.z += sum.z;
    a.w += (float)count;
    accum[pix] = a;
}

// Direct path: one work-item per (pixel, sample lane), every sample traced from
// the camera.  Lane l of a group of g = 2^group_log2 lanes traces samples
// first+l, first+l+g, ... of its pixel and sums them in that order; the g partial
// sums are combined by an xor butterfly, and the group's lane 0 updates the pixel:
//   MODE_ACCUM   accum += (sum, count)                      (rt_render_spp, prefix sharing off)
//   MODE_TRACE   image = sqrt(radiance(sample first))        (`trace`,  raytracer.cl:496-510)
//   MODE_RETRACE image = sqrt(mix(new, image², k/(k+1)))     (`retrace`, raytracer.cl:512-532)
template <int MODE, bool COUNT, bool ACCEL>
__global__ __launch_bounds__(256) void pt_render(DeviceScene sc, FrameParams fp, float4 *__restrict__ accum,
                                                 float4 *__restrict__ image, unsigned long long *counters) {
    __shared__ float4 s_mat[PT_LDS_STATIC_FLOAT4];
    LaneCounters cn;
    if (COUNT) zero_counters(cn);
    Ctx c{sc, stage_materials(sc, s_mat), &cn};
    c.lwin = staged_winners(sc, s_mat);
    c.lpln = staged_planes(sc, s_mat);

    uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    uint32_t g = 1u << fp.group_log2;
    uint32_t slot = fp.slot_begin + (tid >> fp.group_log2);
    uint32_t lane = tid & (g - 1u);
    uint32_t x = 0, y = 0;
    bool valid = slot < fp.slot_end && slot_to_pixel(fp, slot, x, y);

    V3 sum = mk(0.0f, 0.0f, 0.0f);
    if (valid) {
        Ray r0 = primary_ray(fp.cam, x, y, fp.w, fp.h);
        for (uint32_t s = fp.first + lane; s < fp.first + fp.count; s += g) {
            if (COUNT) cn.c[CN_SAMPLES]++;
            sum = sum + radiance<COUNT, ACCEL>(c, r0, s, x, y);
        }
    }
    sum = group_sum(sum, g);
    if (valid && lane == 0) {
        size_t pix = (size_t)y * fp.w + x;
        if (MODE == MODE_ACCUM) {
            accumulate(accum, pix, sum, fp.count);
        } else if (MODE == MODE_TRACE) {
            image[pix] = make_float4(sqrt1(sum.x), sqrt1(sum.y), sqrt1(sum.z), 1.0f);   // gamma_corr :488
        } else {
            if (COUNT) cn.c[CN_IMAGE_READS]++;
            float4 prev = image[pix];
            V3 lin = mk(prev.x * prev.x, prev.y * prev.y, prev.z * prev.z);
            float k = (float)fp.first / (float)(fp.first + 1u);
            V3 o = mk(mix1(sum.x, lin.x, k), mix1(sum.y, lin.y, k), mix1(sum.z, lin.z, k));   // mix(new, prev², k/(k+1)) :526
            image[pix] = make_float4(sqrt1(o.x), sqrt1(o.y), sqrt1(o.z), 1.0f);
        }
    }
    flush_counters<COUNT>(cn, counters, 1);
}

// Fused path, stage 1: one work-item per owned PIXEL traces the sample-invariant
// prefix of the pixel's paths (pt_device.hpp "shared deterministic prefix").
// A pixel whose paths never meet a random event (sky, direct light, mirror /
// glass chains) is finished here: all its samples are equal, and their sum in the
// order of stage 2 (k sequential adds per lane, then log2(g) doublings) is
// computed in closed form.  Other pixels are appended to the live list.
template <bool COUNT, bool ACCEL>
__global__ __launch_bounds__(256) void pt_prefix(DeviceScene sc, FrameParams fp, PixelRec *__restrict__ recs,
                                                 uint32_t *__restrict__ live, uint32_t *__restrict__ live_count,
                                                 float4 *__restrict__ accum, unsigned long long *counters) {
    __shared__ float4 s_mat[PT_LDS_STATIC_FLOAT4];
    LaneCounters cn;
    if (COUNT) zero_counters(cn);
    Ctx c{sc, stage_materials(sc, s_mat), &cn};
    c.lwin = staged_winners(sc, s_mat);
    c.lpln = staged_planes(sc, s_mat);
#if PT_LDS_SPHERES
    __shared__ float4 s_sph[PT_LDS_SPHERE_CAP];
    c.lsph = stage_spheres(sc, s_sph);
#endif

    uint32_t slot = fp.slot_begin + blockIdx.x * 256u + threadIdx.x;
    uint32_t x = 0, y = 0;
    bool valid = slot < fp.slot_end && slot_to_pixel(fp, slot, x, y);
    bool is_live = false;
    PixelRec rec;
    rec.p_kind = rec.n_extra = rec.d = rec.out = rec.col = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (valid) {
        Ray r0 = primary_ray(fp.cam, x, y, fp.w, fp.h);
        rec = trace_prefix<COUNT, ACCEL>(c, r0, x, y);
    }
    if (valid) {
        uint32_t g = 1u << fp.group_log2;
        bool final_px = (__float_as_uint(rec.p_kind.w) & 0xFFu) == REC_FINAL;
        if (final_px) {
            // `count` samples of one colour, summed in the kernels' order: lane l of the pixel's g lanes adds its samples
            // l, l + g, … one after the other (k or k + 1 of them: the first r = count mod g lanes have one more), then the
            // xor butterfly (offsets g/2 … 1).  Before a butterfly step over n lanes the first r lanes hold one value (X)
            // and the others another (Y); lane l takes T(l) + T(l + n/2), so the pattern survives with n/2 lanes:
            // r <= n/2 → (X + Y, Y + Y, r), else (X + X, X + Y, r − n/2).  Lane 0's value after the last step is the pixel's sum.
            // (Until round 3 only counts that are multiples of g took this shortcut; every other count sent its sky pixels
            // through the sample kernel: 56 samples per pixel took longer than 64.)
            const V3 col = xyz(rec.out);
            V3 Y = mk(0.0f, 0.0f, 0.0f);
            for (uint32_t k = 0; k < (fp.count >> fp.group_log2); k++) Y = Y + col;
            V3 X = Y + col;
            uint32_t r = fp.count & (g - 1u);
            for (uint32_t n = g; n > 1u; n >>= 1) {
                const uint32_t h = n >> 1;
                if (r <= h) { X = X + Y; Y = Y + Y; }
                else { Y = X + Y; X = X + X; r -= h; }
            }
            const V3 sum = r ? X : Y;
            accumulate(accum, (size_t)y * fp.w + x, sum, fp.count);
            if (COUNT) cn.c[CN_SAMPLES] += 1;  // scaled by count below
        } else {
            is_live = true;
        }
    }
    // Append the live pixels — slot index and record, both at the pixel's position in the live list, so the sample
    // kernels read records without an indirection.  Order within the list is irrelevant to the result but NOT to the
    // speed of the sample kernel, whose waves take the list chunk by chunk in launch order, a wave living as long as
    // its longest path:
    //  * the pixels of a HEAVY workgroup — one in which some pixel's path went through two or more mirror / glass
    //    bounces, or met glass as its first random event: the neighbourhood of mirrors and glass, where samples get
    //    trapped for many bounces whatever their own first vertex is — are stored from the END of the capacity
    //    downwards and taken FIRST (longest processing time first).  In completion order they sat at the end of the
    //    list (their workgroups finish last) and, started last, WERE the sample kernel's tail: leaving out the last
    //    3 % of the list's chunks made C2's frame 10.6 % shorter, the first 3 % 3.6 %, 3 % in the middle 1.8 %
    //    (profiles/r03_experiments.md);
    //  * the others from 0 upwards in the order in which the workgroups finish — consecutive chunks are neighbouring
    //    pixels: dealing the list out in strands costs 10–55 % (texel and table locality).
    // ONE atomic per WORKGROUP either way: the four waves' counts meet in LDS, thread 0 reserves the workgroup's run,
    // each wave takes its part of it (one atomic per wave made 32 400 waves of a 1080p frame queue on a single
    // address: 0.12 of the kernel's 0.20 ms).
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t kb = __float_as_uint(rec.p_kind.w);
    // (bits 8..15: the vertex's bounce index = mirror / glass bounces before it; a final colour's: hits on its way)
    const bool lane_heavy = valid && (((kb >> 8) & 0xFFu) >= 2u || is_glass_vertex(rec));
    const bool wg_heavy = __syncthreads_or(lane_heavy ? 1 : 0) != 0;
    __shared__ uint32_t s_wave_n[4], s_base;
    unsigned long long m = __ballot(is_live);
    if (lane == 0) s_wave_n[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = s_wave_n[0] + s_wave_n[1] + s_wave_n[2] + s_wave_n[3];
        s_base = total ? atomicAdd(&live_count[wg_heavy ? LIVE_HEAVY_COUNTER : 0u], total) : 0u;
    }
    __syncthreads();
    uint32_t pos = 0u;
    if (is_live) {
        uint32_t before = 0;
        for (uint32_t k = 0; k < wv; k++) before += s_wave_n[k];
        pos = s_base + before + lanes_below(m);
        if (wg_heavy) pos = fp.seg_cap - 1u - pos;
        live[pos] = slot;
        recs[pos] = rec;
    }
    const uint32_t glass_pos = pos;
    // Pixels whose first random event is a dielectric surface are listed for pt_tree_pass (one atomic per wave): their
    // two continuations through the glass are traced once per pixel there instead of once per sample.
    {
        const bool glass = !COUNT && is_live && fp.tree_cap != 0u && is_glass_vertex(rec);
        const unsigned long long gm = __ballot(glass);
        if (gm) {
            uint32_t base = 0u;
            const int leader = __builtin_ctzll(gm);
            if ((int)lane == leader) base = atomicAdd(fp.tree_count, (uint32_t)__popcll(gm));
            base = (uint32_t)__shfl((int)base, leader);
            const uint32_t gi = base + lanes_below(gm);
            if (glass && gi < fp.tree_cap) {
                fp.glass[gi] = glass_pos;
                fp.trees[gi].dec[0].hsh = 0u;   // the tree's leaf counter
            }
        }
    }
    flush_counters<COUNT>(cn, counters, fp.count);  // the prefix stands for `count` samples' worth of work
}

// Fused path, stage 1b: the shared decision trees (pt_types.hpp PixelTree) of the pixels pt_prefix listed, one LEVEL per
// launch.  Two work-items per waiting glass vertex — one per continuation (refracted / reflected ray) — so that every
// work-item traces ONE stretch of path (grid-stride: the work's length is only known on the device).  Level 0 takes
// the listed pixels' own records (the root, heap node 1) and rewrites them (REC_TREE + tree index; or the one record
// every sample continues from, when the glass reflects totally); a continuation that ends at another glass vertex
// waits in `out` for the next level, below the last level it becomes a leaf the samples continue from on their own.
template <bool ACCEL>
__global__ __launch_bounds__(256) void pt_tree_pass(DeviceScene sc, FrameParams fp, PixelRec *__restrict__ recs, uint32_t level,
                                                   const TreeWork *__restrict__ in, const uint32_t *__restrict__ in_count,
                                                   TreeWork *__restrict__ out, uint32_t *__restrict__ out_count, uint32_t q_cap) {
    __shared__ float4 s_mat[PT_LDS_STATIC_FLOAT4];
    Ctx c{sc, stage_materials(sc, s_mat), nullptr};
    c.lwin = staged_winners(sc, s_mat);
    c.lpln = staged_planes(sc, s_mat);
    const uint32_t n = level == 0u ? min(*fp.tree_count, fp.tree_cap) : min(*in_count, q_cap);
    const uint32_t lane = threadIdx.x & 63u;
    // (wave-uniform loop: the appends to the next level's queue are one atomic per WAVE — one per work-item made a
    // hundred thousand atomics queue on a single address, 0.2 ms of a 1.9 ms frame)
    for (uint32_t t0 = blockIdx.x * 256u + (threadIdx.x & ~63u); t0 < 2u * n; t0 += gridDim.x * 256u) {
        const uint32_t t = t0 + lane;
        const bool live = t < 2u * n;
        const uint32_t item = live ? t >> 1 : 0u, which = t & 1u;
        uint32_t tree = item, heap = 1u, pos = 0u;
        PixelRec rec;
        rec.p_kind = rec.n_extra = rec.d = rec.out = rec.col = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // (REC_FINAL: not a glass vertex)
        if (live) {
            if (level == 0u) {
                pos = fp.glass[item];
                rec = recs[pos];
            } else {
                rec = in[item].rec;
                tree = in[item].tree;
                heap = in[item].heap;
            }
        }
        PixelTree *T = fp.trees + tree;
        float prob = 0.0f;
        const bool decision = tree_settle<ACCEL>(c, rec, prob) && live;
        if (live && !decision && which == 0u) {
            // no decision here after all (total internal reflection led to another kind of vertex): a leaf
            if (heap == 1u) recs[pos] = rec;   // the pixel needs no tree: the prefix simply went on
            else {
                const uint32_t li = atomicAdd(&T->dec[0].hsh, 1u);
                T->leaf[li] = rec;
                T->dec[heap >> 1].child[heap & 1u] = (uint16_t)(0x8000u | li);
            }
        }
        if (decision && which == 0u) {
            T->dec[heap].hsh = dir_hash(xyz(rec.d));
            T->dec[heap].prob = prob;
            T->dec[heap].depth = (__float_as_uint(rec.p_kind.w) >> 8) & 0xFFu;
            if (heap == 1u) {
                recs[pos].p_kind.w = __uint_as_float((uint32_t)REC_TREE);
                recs[pos].col.w = __uint_as_float(tree);
            } else {
                T->dec[heap >> 1].child[heap & 1u] = (uint16_t)heap;
            }
        }
        PixelRec rb = rec;
        if (decision) rb = tree_branch<ACCEL>(c, rec, which);
        const bool wait = decision && is_glass_vertex(rb) && level + 1u < PT_TREE_LEVELS;
        const unsigned long long wm = __ballot(wait);
        bool queued = false;
        if (wm) {
            uint32_t base = 0u;
            const int leader = __builtin_ctzll(wm);
            if ((int)lane == leader) base = atomicAdd(out_count, (uint32_t)__popcll(wm));
            base = (uint32_t)__shfl((int)base, leader);
            const uint32_t qi = base + lanes_below(wm);
            if (wait && qi < q_cap) {
                out[qi].rec = rb;
                out[qi].tree = tree;
                out[qi].heap = 2u * heap + which;
                queued = true;   // (its parent's child link is written when the vertex is settled, next level)
            }
        }
        if (decision && !queued) {
            const uint32_t li = atomicAdd(&T->dec[0].hsh, 1u);
            T->leaf[li] = rb;
            T->dec[heap].child[which] = (uint16_t)(0x8000u | li);
        }
    }
}

// Fused path, stage 2: one group of g lanes per LIVE pixel; each lane continues
// its samples from the pixel's record.  Same summation order as pt_render.
template <bool COUNT, bool ACCEL>
__global__ __launch_bounds__(256) void pt_samples(DeviceScene sc, FrameParams fp, const PixelRec *__restrict__ recs,
                                                  const uint32_t *__restrict__ live,
                                                  const uint32_t *__restrict__ live_count,
                                                  float4 *__restrict__ accum, unsigned long long *counters) {
    __shared__ float4 s_mat[PT_LDS_STATIC_FLOAT4];
    LaneCounters cn;
    if (COUNT) zero_counters(cn);
    Ctx c{sc, stage_materials(sc, s_mat), &cn};
    c.lwin = staged_winners(sc, s_mat);
    c.lpln = staged_planes(sc, s_mat);

    uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    uint32_t g = 1u << fp.group_log2;
    uint32_t li = tid >> fp.group_log2;
    uint32_t lane = tid & (g - 1u);
    uint32_t entry = 0;
    bool valid = live_take(fp, live_count, li, 1u, entry) != 0u;
    uint32_t x = 0, y = 0;
    V3 sum = mk(0.0f, 0.0f, 0.0f);
    if (valid) {
        uint32_t slot = live[entry];
        (void)slot_to_pixel(fp, slot, x, y);
        PixelRec rec = recs[entry];
        bool final_px = (__float_as_uint(rec.p_kind.w) & 0xFFu) == REC_FINAL;
        for (uint32_t s = fp.first + lane; s < fp.first + fp.count; s += g) {
            if (COUNT) cn.c[CN_SAMPLES]++;
            sum = sum + radiance_from_rec<COUNT, ACCEL>(c, rec, s, x, y, fp.trees);
        }
        (void)final_px;
    }
    sum = group_sum(sum, g);
    if (valid && lane == 0) accumulate(accum, (size_t)y * fp.w + x, sum, fp.count);
    flush_counters<COUNT>(cn, counters, 1);
}

// Fused path, stage 2 with an in-wave SAMPLE QUEUE (default).  Path lengths differ
// wildly between samples (1 bounce into the sky … 30 inside glass), so with one fixed
// sample per lane most lanes of a wave idle while its longest path finishes.  Here a
// wave owns P live pixels = up to QUEUE_SLOTS samples and its 64 lanes pull the next
// sample whenever their path ends: every iteration is "scatter, then nearest hit" for
// all lanes, new samples joining at the scatter step straight from their pixel's
// record (staged in LDS).  A finished sample's radiance goes to its own LDS slot, and
// the slots are summed in exactly the order of pt_render (lane l: samples l, l+g, …;
// then the xor butterfly), so the result does not depend on which lane traced what.
#ifndef QUEUE_SLOTS
#define QUEUE_SLOTS 512   // upper bound of samples a wave owns; the launch picks pixels_per_wave
#endif
#ifndef PT_REFILL_MIN
#define PT_REFILL_MIN 1   // idle lanes that trigger a refill
#endif
#ifndef QUEUE_MAX_PIXELS
#define QUEUE_MAX_PIXELS 16
#endif
// dynamic LDS of pt_samples_q, per workgroup: materials, then per wave {records, coordinates, slots}
__host__ __device__ inline uint32_t queue_wave_lds_bytes(uint32_t pixels_per_wave, uint32_t count) {
    uint32_t b = pixels_per_wave * 5u * 16u + pixels_per_wave * 4u * 4u + pixels_per_wave * count * 3u * 4u;
    return (b + 15u) & ~15u;
}
#ifndef PT_UNIFORM_WAVE
#define PT_UNIFORM_WAVE 1
#endif
#ifndef PT_Q_WAVES
#define PT_Q_WAVES 6  // waves per SIMD the register allocator must leave room for: 6 = 80 VGPRs (A/B on C2: 5 → 2.62 ms, 6 → 2.48)
#endif
#ifndef PT_Q_WAVES_ACCEL
#define PT_Q_WAVES_ACCEL 5  // scenes that mix BVH meshes with small ones (every other mesh scene runs pt_samples_w), and scenes with a sphere BVH
                            // next to lenses / small meshes: 96 VGPRs (at 6 waves per SIMD = 80 VGPRs these instantiations spill 21 registers)
#endif
#ifndef PT_Q_WAVES_SPHERE_BVH
#define PT_Q_WAVES_SPHERE_BVH 6  // scenes whose only BVH is the sphere BVH (C4 at 8 spp, r02: 5 → 76.9 ms, 6 → 71.8 ms)
#endif
static_assert(QUEUE_SLOTS >= RT_SPP_PER_LAUNCH, "rt_render_spp's launches must fit a wave's sample queue");
#ifndef QUEUE_MIN_SAMPLES
#define QUEUE_MIN_SAMPLES 384u
#endif
// Pixels per wave: as many as the LDS of a CU allows with PT_Q_WAVES(_ACCEL) workgroups resident (6: 160 KB / 6 per
// workgroup); when that leaves a wave fewer than 384 samples (256 spp and up: the queue's tail grows) the
// budget of 5 resident workgroups is used instead — the kernel's 80 VGPRs fit either way.
#ifndef PT_LDS_GRANULE
#define PT_LDS_GRANULE 1024u
#endif
__host__ inline uint32_t queue_pixels_per_wave(uint32_t count, uint32_t waves, uint32_t static_float4, uint32_t block_waves = 4u) {
    auto fit = [&](uint32_t waves_per_simd) {
        uint32_t workgroups = waves_per_simd * 4u / block_waves;  // resident workgroups per CU
        // (LDS is handed out in blocks: a request of 6 584 bytes — 7 pixels of 64 samples — left fewer than 24 workgroups
        // resident although 24 × 6 584 < 160 KiB, and 6 pixels (5 728 bytes) are 4.5 % faster on C2; the budget is
        // therefore rounded DOWN to a multiple of PT_LDS_GRANULE)
        uint32_t budget = 163840u / workgroups / PT_LDS_GRANULE * PT_LDS_GRANULE;
        uint32_t per_wave = (budget - static_float4 * (uint32_t)sizeof(float4)) / block_waves - 15u;
        uint32_t p = per_wave / (5u * 16u + 4u * 4u + count * 3u * 4u);
        if (p * count > QUEUE_SLOTS) p = QUEUE_SLOTS / count;
        return p > QUEUE_MAX_PIXELS ? (uint32_t)QUEUE_MAX_PIXELS : p;
    };
    uint32_t p = fit(waves);
    if (p * count < QUEUE_MIN_SAMPLES) {
        uint32_t p5 = fit(5u);
        if (p5 > p) p = p5;
    }
    return p < 1u ? 1u : p;
}
#ifndef PT_UNITS_PER_WAVE_SLOT
#define PT_UNITS_PER_WAVE_SLOT 16u   // waves a sample-kernel launch should have per wave slot of the chip (launch_fused)
#endif
#ifndef PT_TREE_MIN_SAMPLES
#define PT_TREE_MIN_SAMPLES 24u   // samples per call from which the shared decision trees pay (launch_fused; RT_OPT_PREFIX_TREE 1)
#endif
#ifndef PT_LDS_FACE_CAP
#define PT_LDS_FACE_CAP 64u   // faces (48 bytes each) of a scene of small meshes that may be staged in LDS (launch_fused)
#endif
#ifndef PT_Q_BLOCK_WAVES
#define PT_Q_BLOCK_WAVES 1  // waves per workgroup of pt_samples_q (they share only the staged materials): a wave that is through frees
                            // its LDS and wave slot at once instead of waiting for three others (A/B on C2: 4 → 2.42 ms, 2 → 2.42, 1 → 2.34)
#endif
// ACCEL: the sphere BVH walk is compiled in.  GEOM: 0 = the scene holds spheres and planes only (C1, C2, C4: no
// lens, model or mesh code at all), 1 = everything by brute force or through the sphere BVH, 2 = the mesh BVH
// walk too.  A scene whose only BVH is the sphere BVH (C4) runs <true, 0>: without the mesh walk's registers the
// kernel keeps 6 waves per SIMD.
template <bool COUNT, bool ACCEL, int GEOM, int WAVES>
__global__ __launch_bounds__(64 * PT_Q_BLOCK_WAVES, WAVES) void pt_samples_q(DeviceScene sc, FrameParams fp, const PixelRec *__restrict__ recs,
                                                    const uint32_t *__restrict__ live,
                                                    const uint32_t *__restrict__ live_count,
                                                    float4 *__restrict__ accum, unsigned long long *counters,
                                                    uint32_t pixels_per_wave) {
    extern __shared__ float4 s_dyn[];  // 16-byte aligned: no static LDS in this kernel
    float4 *s_mat = s_dyn;
    LaneCounters cn;
    if (COUNT) zero_counters(cn);
    Ctx c{sc, stage_materials(sc, s_mat), &cn};
    c.lwin = staged_winners(sc, s_mat);
    c.lpln = staged_planes(sc, s_mat);
#if PT_LDS_SPHERES
    c.lsph = stage_spheres(sc, s_dyn + lds_static_used(sc.material_count, sc.sphere_count, sc.plane_count));
#endif
    if (GEOM != 0 && fp.lds_face_f4) {   // the face records of a scene of a few small meshes (hit_models' candidate loop)
        float4 *s_faces = s_dyn + lds_static_used(sc.material_count, sc.sphere_count, sc.plane_count);
        for (uint32_t i = threadIdx.x; i < fp.lds_face_f4; i += blockDim.x) s_faces[i] = sc.faces[i];
        __syncthreads();
        c.lfaces = lds_ptr(s_faces);
    }

#if PT_Q_BLOCK_WAVES == 1
    const uint32_t wave = 0u, lane = threadIdx.x;
#elif PT_UNIFORM_WAVE
    // the wave index is wave-uniform, which the compiler cannot see: this puts everything derived from it in SGPRs
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
#else
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
#endif
    char *wave_lds = reinterpret_cast<char *>(s_dyn + lds_static_used(sc.material_count, sc.sphere_count, sc.plane_count) +
                                              (PT_LDS_SPHERES ? PT_LDS_SPHERE_CAP : 0) + (GEOM != 0 ? fp.lds_face_f4 : 0u)) +
                     (size_t)wave * queue_wave_lds_bytes(pixels_per_wave, fp.count);
    float4 *s_rec = reinterpret_cast<float4 *>(wave_lds);
    uint32_t *s_xy = reinterpret_cast<uint32_t *>(s_rec + pixels_per_wave * 5u);
    float *slot = reinterpret_cast<float *>(s_xy + pixels_per_wave * 4u);
    uint32_t pix0 = 0;
    const uint32_t npix = live_take(fp, live_count, blockIdx.x * (uint32_t)PT_Q_BLOCK_WAVES + wave, pixels_per_wave, pix0);
    const uint32_t count = fp.count, total = npix * count;
    const float4 *rec = s_rec;
    const uint32_t *xy = s_xy;

    // stage this wave's pixel records and coordinates
    for (uint32_t i = lane; i < npix * 5u; i += 64u) {
        uint32_t p = i / 5u, part = i - p * 5u;
        s_rec[i] = reinterpret_cast<const float4 *>(recs + pix0 + p)[part];
    }
    if (lane < npix) {
        uint32_t x = 0, y = 0;
        (void)slot_to_pixel(fp, live[pix0 + lane], x, y);
        // (x, y, and the pixel's part of the two table index sums: rnd_base_v = (sample·2683 + x·3931 + y·2504)·3 and
        // rnd_base_u = sample·2683 + x·3931 + y are linear in uint32 arithmetic, so a refill needs two multiplies, not five)
        s_xy[4 * lane] = x;
        s_xy[4 * lane + 1] = y;
        s_xy[4 * lane + 2] = rnd_base_v(0u, x, y);
        s_xy[4 * lane + 3] = rnd_base_u(0u, x, y);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const float inv_count = fp.inv_count;
    const uint32_t count_log2 = (count & (count - 1u)) == 0u ? (uint32_t)__builtin_ctz(count) : 0xFFu;
    uint32_t next = 0;  // wave-uniform head of the queue
    bool active = false;
    // Per-lane state carried from one iteration to the next, kept small (the kernel sits on its VGPR budget):
    // the hit POINT is not carried — the ray's origin is moved there as soon as the hit is known — and the
    // per-sample part of the table index sums is precomputed (bv, bu) instead of carrying sample, x and y.
    uint32_t idx = 0, depth = 0, bv = 0, bu = 0;
    V3 col = mk(0.0f, 0.0f, 0.0f), out = mk(0.0f, 0.0f, 0.0f);   // col: material colour or texel of the hit
    Ray r;
    r.o = r.d = mk(0.0f, 0.0f, 0.0f);
    V3 hn = mk(0.0f, 0.0f, 0.0f);   // normal and material of the hit the next interaction happens at
    uint32_t hmat = 0;
    Rnd rnd;
    rnd.v = mk(0.0f, 0.0f, 0.0f);
    rnd.u = 0.0f;

#if PT_STAMPS
    c.st_last = __builtin_amdgcn_s_memtime();
#endif
    while (true) {
        PT_STAMP(c, 5);
        // ---- refill idle lanes from the queue
        bool need = !active;
        unsigned long long m = __ballot(need);
        // refill when enough lanes idle (or none is active): the refill step issues for the whole wave
        if (m && next < total && ((uint32_t)__popcll(m) >= PT_REFILL_MIN || m == ~0ull)) {
            uint32_t cand = next + lanes_below(m);
            if (need && cand < total) {
                idx = cand;
                // pixel of this queue entry: p = idx / count, exactly, without an integer divide:
                // (idx + 0.5)/count lies >= 0.5/count away from every integer, far more than the rounding
                // of the float product (idx < 8192, count <= 512)
                // (a power-of-two count — wave-uniform — needs a shift; otherwise one float multiply:)
                uint32_t p = count_log2 != 0xFFu ? idx >> count_log2 : (uint32_t)(((float)idx + 0.5f) * inv_count);
                const uint32_t sample = fp.first + (idx - p * count);
                float4 q0 = rec[5 * p], q1 = rec[5 * p + 1], q2 = rec[5 * p + 2], q3 = rec[5 * p + 3],
                       q4 = rec[5 * p + 4];
                bv = sample * 8049u + xy[4 * p + 2];   // = rnd_base_v(sample, x, y)
                bu = sample * 2683u + xy[4 * p + 3];   // = rnd_base_u(sample, x, y)
                uint32_t bits = __float_as_uint(q0.w);
                if ((bits & 0xFFu) == REC_TREE) {   // the pixel has a shared decision tree: this sample's leaf
                    const float4 *lf = tree_leaf(fp.trees + __float_as_uint(q4.w), sc.table, bu);
                    q0 = lf[0]; q1 = lf[1]; q2 = lf[2]; q3 = lf[3]; q4 = lf[4];
                    bits = __float_as_uint(q0.w);
                }
                if (COUNT) cn.c[CN_SAMPLES]++;
                if ((bits & 0xFFu) == REC_FINAL) {  // a leaf of a tree, or count is not a multiple of g
                    slot[3 * idx] = q3.x;
                    slot[3 * idx + 1] = q3.y;
                    slot[3 * idx + 2] = q3.z;
                } else {
                    depth = (bits >> 8) & 0xFFu;   // (type and extra_data of the record are the material's: re-read below)
                    hn = xyz(q1);
                    r.o = xyz(q0);
                    r.d = xyz(q2);
                    hmat = __float_as_uint(q2.w);
                    out = xyz(q3);
                    col = xyz(q4);
                    if (PT_RNG_PREFETCH) rnd = fetch_rnd_b(sc.table, r.d, depth, bv, bu);
                    active = true;
                }
            }
            next += (uint32_t)__popcll(m);
        }
        if (!__any(active)) {
            if (next >= total) break;
            continue;  // every candidate was a final-colour pixel: keep draining the queue
        }
        PT_STAMP(c, 0);
#ifdef PT_EXP_PAD  // timing experiment (tools/pad_experiment.sh): PT_EXP_PAD extra full-rate VALU instructions per iteration
                   // (v_or_b32 x, x, x on a live register: no new VGPR; PT_EXP_PAD_NOP: operand-free v_nop instead).  An
                   // issue-bound loop slows down by their issue time, a latency-bound one does not (profiles/r03_experiments.md)
#define PT_STR2(x) #x
#define PT_STR(x) PT_STR2(x)
#ifdef PT_EXP_PAD_NOP
        asm volatile(".rept " PT_STR(PT_EXP_PAD) "\n\tv_nop\n\t.endr");
#else
        asm volatile(".rept " PT_STR(PT_EXP_PAD) "\n\tv_or_b32 %0, %0, %0\n\t.endr" : "+v"(idx));
#endif
#endif
#ifdef PT_QSTAT  // diagnostic: lane-iterations used / offered (read through rt_get_debug_counters on a BVH-free scene)
        if (COUNT) {
            uint32_t na = (uint32_t)__popcll(__ballot(active));
            if (lane == 0) cn.c[CN_DBG_BVH_NODES] += na;
            if (lane == 0) cn.c[CN_DBG_BVH_TESTS] += 64u;
        }
#endif
        // ---- one material interaction for every active lane
        if (active) {
            if (!PT_RNG_PREFETCH) rnd = fetch_rnd_b(sc.table, r.d, depth, bv, bu);
            Hit at;   // the vertex this interaction happens at: the ray's origin already stands on it
            at.p = r.o;
            at.n = hn;
            at.u = at.v = 0.0f;
            at.tex = 0;
            at.mat = hmat;
            int type;
            float extra;
            V3 mcol;
            load_material(c, hmat, type, extra, mcol);   // type and extra_data are not carried: one LDS read each
            scatter<COUNT>(c, r, out, at, type, extra, col, rnd, false);
            depth++;
            if (depth >= RT_DEPTH) {  // survived DEPTH bounces: returns what it has (:447,485)
                slot[3 * idx] = out.x;
                slot[3 * idx + 1] = out.y;
                slot[3 * idx + 2] = out.z;
                active = false;
            }
        }
        PT_STAMP(c, 1);
        // ---- nearest hit for every lane still active; the table reads of the NEXT material
        // interaction are issued first (they depend on the ray direction only)
        if (active) {
            if (PT_RNG_PREFETCH == 1) rnd = fetch_rnd_b(sc.table, r.d, depth, bv, bu);
            V3 res;
            bool done = false;
            Hit h;
            h.p = h.n = mk(0.0f, 0.0f, 0.0f);
            h.u = h.v = 0.0f;
            h.tex = h.mat = 0;
            Nearest nb;
            hit_primitives<COUNT, ACCEL, GEOM != 0>(c, r, nb);
            if (GEOM != 0) hit_models<COUNT, GEOM == 2>(c, r, nb);
            if (!hit_finish<COUNT, GEOM == 0>(c, r, nb, h)) {
                res = mk(0.0f, 0.0f, 0.0f);
                done = true;
            } else {
                if (COUNT) cn.c[CN_H_BOUNCE]++;
                int type;
                float extra;
                load_material(c, h.mat, type, extra, col);
                if (type == RT_LIGHT) {
                    res = vmin(out, col);
                    done = true;
                } else if (type == RT_TEXTURED) {
                    if (COUNT) cn.c[CN_N_TEXFETCH]++;
                    col = texture_rgb(c.sc, h.u, h.v, h.tex);
                }
            }
            if (done) {
                slot[3 * idx] = res.x;
                slot[3 * idx + 1] = res.y;
                slot[3 * idx + 2] = res.z;
                active = false;
            } else if (PT_RNG_PREFETCH == 2) {
                // this lane WILL interact next iteration: its table reads fly during the refill step
                rnd = fetch_rnd_b(sc.table, r.d, depth, bv, bu);
            }
            if (!done) {   // the next interaction happens here
                r.o = h.p;
                hn = h.n;
                hmat = h.mat;
            }
        }
        PT_STAMP(c, 4);
    }
#if PT_STAMPS
    if (lane == 0 && npix)
        for (int k = 0; k < 6; k++) atomicAdd(&counters[(size_t)COUNTER_REPLICAS * COUNTER_STRIDE + k], c.st[k]);
#endif
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- per-pixel sums in pt_render's order
    const uint32_t g = 1u << fp.group_log2, ppp = 64u >> fp.group_log2;
    for (uint32_t pb = 0; pb < npix; pb += ppp) {
        uint32_t p = pb + (lane >> fp.group_log2), l = lane & (g - 1u);
        V3 sum = mk(0.0f, 0.0f, 0.0f);
        if (p < npix)
            for (uint32_t j = l; j < count; j += g) {
                const float *sl = slot + 3u * (p * count + j);
                sum = sum + mk(sl[0], sl[1], sl[2]);
            }
        sum = group_sum(sum, g);
        if (p < npix && l == 0) accumulate(accum, (size_t)xy[4 * p + 1] * fp.w + xy[4 * p], sum, count);
    }
    flush_counters<COUNT>(cn, counters, 1);
}

// pt_samples_w — the sample queue for scenes in which every mesh of every model has a BVH (C5: one mesh of
// 50 000 faces).  The mesh walk
// is the bulk of such a frame, and its length differs per lane from a handful of nodes (the root is missed)
// to hundreds: run in place, a wave executes the walk loop until its slowest lane is through — rocprofv3
// counted 9 of 64 lanes active per VALU instruction on C5.  Here every lane is a small state machine
//     0 material interaction + spheres/planes/lenses → 1 walking → 2 winner's record, material, next bounce
// and each loop iteration advances EVERY walking lane by at most PT_WALK_STEPS nodes (the threaded walk's
// whole position is one node index), while lanes in the cheap states 0 and 2 pass through them: lanes start
// and finish walks at different times, so the walk loop always has many lanes in it.  Same arithmetic per
// sample as pt_samples_q, same slots, same summation order: bit-identical.
#ifndef PT_WALK_STEPS
#define PT_WALK_STEPS 24u  // A/B: 8 → 122.9 ms, 16 → 118.3, 24 → 116.5, 48 → 118.4
#endif
#ifndef PT_W_BATCH
#define PT_W_BATCH 1u   // lanes that must be waiting for a cheap step before the cheap steps run (pt_samples_w).  A/B on C5 at 1080p x 64 spp:
                        // 1 → 20.87 ms, 8 → 21.14, 16 → 21.41, 24 → 21.79, 32 → 22.42 (16 with slices of 12 / 8 nodes: 20.79 / 21.01):
                        // waiting lanes cost more than sparsely filled cheap steps — batching stays off
#endif
// The root of the (single) mesh is tested while the lane is still in state 0: a ray that misses the whole mesh (a third of
// C5's walks) goes straight to state 2 instead of idling through a slice, and the cheap states repeat (at most
// PT_W_CHEAP_REPEATS times) while at least PT_W_CHEAP_AGAIN lanes came out of them with no walk to join.  C5 at 4K x 512 spp:
// off 561.8 ms, root test without repeats 571.6, repeats from 4 / 8 / 16 lanes 548.6 / 546.6 / 545.6.  (Ending a slice early once
// 8 / 16 / 24 of its lanes are through: 593.7 / 558.8 / 553.4 vs 548.2 — the fixed slice stays.)
#ifndef PT_W_ROOT_FIRST
#define PT_W_ROOT_FIRST 1
#endif
#ifndef PT_W_CHEAP_AGAIN
#define PT_W_CHEAP_AGAIN 8u
#endif
#ifndef PT_W_CHEAP_REPEATS
#define PT_W_CHEAP_REPEATS 4u
#endif
#ifndef PT_W_WAVES
#define PT_W_WAVES 6  // A/B on C5 at 16 spp (round 2, 92 VGPRs): 4 → 116.5 ms, 5 → 110.2, 6 → 116.1 (spills); round 3 (the loop reordered: 79 VGPRs) at 1080p x 64 spp: 5 → 20.87, 6 → 20.65
#endif
#ifndef PT_W_BLOCK_WAVES
#define PT_W_BLOCK_WAVES 1  // waves per workgroup (see PT_Q_BLOCK_WAVES)
#endif
#ifndef PT_W_WAVES_MULTI
#define PT_W_WAVES_MULTI 4  // several meshes: the running minimum over the jobs needs 13 more VGPRs — 109, no scratch at 4 waves per SIMD
#endif
template <bool MULTI>
__global__ __launch_bounds__(64 * PT_W_BLOCK_WAVES, MULTI ? PT_W_WAVES_MULTI : PT_W_WAVES) void pt_samples_w(DeviceScene sc, FrameParams fp, const PixelRec *__restrict__ recs,
                                                    const uint32_t *__restrict__ live,
                                                    const uint32_t *__restrict__ live_count,
                                                    float4 *__restrict__ accum, uint32_t pixels_per_wave,
                                                    const uint2 *__restrict__ jobs, uint32_t n_jobs
#ifdef PT_WSTAT
                                                    , unsigned long long *wstat
#endif
                                                    ) {
    extern __shared__ float4 s_dyn[];
    float4 *s_mat = s_dyn;
    Ctx c{sc, stage_materials(sc, s_mat), nullptr};
    c.lwin = staged_winners(sc, s_mat);
    c.lpln = staged_planes(sc, s_mat);
#if PT_W_BLOCK_WAVES == 1
    const uint32_t wave = 0u, lane = threadIdx.x;
#else
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
#endif
    char *wave_lds = reinterpret_cast<char *>(s_dyn + lds_static_used(sc.material_count, sc.sphere_count, sc.plane_count)) +
                     (size_t)wave * queue_wave_lds_bytes(pixels_per_wave, fp.count);
    float4 *s_rec = reinterpret_cast<float4 *>(wave_lds);
    uint32_t *s_xy = reinterpret_cast<uint32_t *>(s_rec + pixels_per_wave * 5u);
    float *slot = reinterpret_cast<float *>(s_xy + pixels_per_wave * 4u);
    uint32_t pix0 = 0;
    const uint32_t npix = live_take(fp, live_count, blockIdx.x * (uint32_t)PT_W_BLOCK_WAVES + wave, pixels_per_wave, pix0);
    const uint32_t count = fp.count, total = npix * count;
    const float4 *rec = s_rec;
    const uint32_t *xy = s_xy;
    for (uint32_t i = lane; i < npix * 5u; i += 64u) {
        uint32_t p = i / 5u, part = i - p * 5u;
        s_rec[i] = reinterpret_cast<const float4 *>(recs + pix0 + p)[part];
    }
    if (lane < npix) {
        uint32_t x = 0, y = 0;
        (void)slot_to_pixel(fp, live[pix0 + lane], x, y);
        // (x, y, and the pixel's part of the two table index sums: rnd_base_v = (sample·2683 + x·3931 + y·2504)·3 and
        // rnd_base_u = sample·2683 + x·3931 + y are linear in uint32 arithmetic, so a refill needs two multiplies, not five)
        s_xy[4 * lane] = x;
        s_xy[4 * lane + 1] = y;
        s_xy[4 * lane + 2] = rnd_base_v(0u, x, y);
        s_xy[4 * lane + 3] = rnd_base_u(0u, x, y);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // The walks of a bounce, in the reference's order: jobs[j] = (mesh index, material of its model), model by
    // model, mesh by mesh.  hitModel's "nearest of my meshes" followed by hitScene's "nearer than the best so
    // far" (:305-320, :349-356) equals ONE running strict-< minimum over this flat list, which is what state 1
    // keeps.  MULTI = false: a single job, its constants wave-uniform.
    const uint32_t mesh0 = jobs[0].x, mat0 = jobs[0].y;
    const uint32_t faces0 = sc.meshes[mesh0].face_count;
    const uint32_t root0 = sc.mesh_bvh_root[mesh0];

    const float inv_count = fp.inv_count;
    const uint32_t count_log2 = (count & (count - 1u)) == 0u ? (uint32_t)__builtin_ctz(count) : 0xFFu;
    uint32_t next = 0;  // wave-uniform head of the queue
    bool active = false;
    int phase = 0;
    uint32_t idx = 0, depth = 0, bv = 0, bu = 0;   // (see pt_samples_q: hit point and sample / pixel are not carried)
    V3 col = mk(0.0f, 0.0f, 0.0f), out = mk(0.0f, 0.0f, 0.0f);   // col: material colour or texel of the hit
    Ray r;
    r.o = r.d = mk(0.0f, 0.0f, 0.0f);
    V3 hn = mk(0.0f, 0.0f, 0.0f);
    uint32_t hmat = 0;
    float nb_t = RT_MAX_DISTANCE;       // nearest sphere / plane / lens of the current bounce
    uint32_t nb_id = PT_NO_HIT;
    MeshWalk wpos = mesh_walk_start(0);  // the walk's position and its best face so far
    uint32_t wbest = 0;
    float wt = 0.0f, wu = 0.0f, wv = 0.0f;
    uint32_t job = 0;                   // MULTI: the job being walked, and the winning mesh hit so far
    uint32_t nb_face = 0, nb_mat = 0;
    float nb_u = 0.0f, nb_v = 0.0f;

    // MULTI: move `job` on to the next mesh whose ROOT the ray does not miss — the root's node test runs here, in the cheap
    // code, so a bounce costs a walk slice per mesh the ray may touch, not one per mesh of the scene — and set the walk
    // up behind that root; false: no such mesh is left (the bounce's mesh search is over)
    auto enter_job = [&]() {
        while (job < n_jobs) {
            const uint32_t mesh_n = jobs[job].x;
            wbest = sc.meshes[mesh_n].face_count;
            wt = wu = wv = 0.0f;
            wpos = mesh_walk_first(sc, r, sc.mesh_bvh_root[mesh_n], wbest);
            if (wpos.cur != PT_MESH_END) return true;
            job++;
        }
        return false;
    };
#ifdef PT_WSTAT
    WalkStat ws = {0, 0, 0, 0, 0};
    unsigned long long it_n = 0, it_active = 0, it_p0 = 0, it_p1 = 0, it_p2 = 0, it_walk_calls = 0;
#endif
    // every iteration takes samples off the queue, or moves every active lane on (a bounce, or up to
    // PT_WALK_STEPS nodes of a walk that visits each of the < 2^28 nodes at most 3 times)
    for (unsigned long long guard = ((unsigned long long)total + 1ull) * (RT_DEPTH + 2ull) * (3ull * (1ull << 28) / PT_WALK_STEPS + 4ull); guard; guard--) {
        // Which lanes are inside a walk, and which want one of the cheap steps (the winner's record after a walk, a new
        // sample from the queue, a material interaction + the primitives that are not models)?  The cheap steps are
        // BATCHED: they run when at least PT_W_BATCH lanes want one (or nothing is walking) — run in every iteration
        // they executed for the handful of lanes whose walks had just ended, at the price of a wave's whole issue time.
        const bool walking = active && phase == 1;
        const bool wants_cheap = (active && phase != 1) || (!active && next < total);
        const uint32_t n_cheap = (uint32_t)__popcll(__ballot(wants_cheap));
        const bool any_walk = __any(walking);
        if (n_cheap == 0u && !any_walk) break;   // every path is through and the queue is empty
#ifdef PT_WSTAT
        it_n++;
        it_active += __popcll(__ballot(active));
        it_p0 += __popcll(__ballot(active && phase == 0));
        it_p1 += __popcll(__ballot(active && phase == 1));
        it_p2 += __popcll(__ballot(active && phase == 2));
#endif
        if (n_cheap >= PT_W_BATCH || !any_walk) {
          // (PT_W_ROOT_FIRST: a ray that misses the mesh's root goes from state 0 straight to state 2; the cheap states
          // repeat while at least PT_W_CHEAP_AGAIN lanes came out of them without a walk to join)
          for (uint32_t again = 0;; again++) {
            // ---- state 2: the winner's record, its material
            if (active && phase == 2) {
                Nearest nb;
                nb.t = nb_t;
                nb.id = nb_id;
                if (MULTI) {
                    nb.face = nb_face;
                    nb.mat = nb_mat;
                    nb.u = nb_u;
                    nb.v = nb_v;
                } else if (wbest < faces0 && wt < RT_MAX_DISTANCE && wt < nb.t) {
                    nb.t = wt;
                    nb.id = K_MESH | mesh0;
                    nb.face = wbest;
                    nb.mat = mat0;
                    nb.u = wu;
                    nb.v = wv;
                }
                V3 res;
                bool done = false;
                Hit h;
                h.p = h.n = mk(0.0f, 0.0f, 0.0f);
                h.u = h.v = 0.0f;
                h.tex = h.mat = 0;
                if (!hit_finish<false>(c, r, nb, h)) {
                    res = mk(0.0f, 0.0f, 0.0f);
                    done = true;
                } else {
                    int type;
                    float extra;
                    load_material(c, h.mat, type, extra, col);
                    if (type == RT_LIGHT) {
                        res = vmin(out, col);
                        done = true;
                    } else if (type == RT_TEXTURED) {
                        col = texture_rgb(c.sc, h.u, h.v, h.tex);
                    }
                }
                phase = 0;
                if (done) {
                    slot[3 * idx] = res.x;
                    slot[3 * idx + 1] = res.y;
                    slot[3 * idx + 2] = res.z;
                    active = false;
                } else {   // the next interaction happens here
                    r.o = h.p;
                    hn = h.n;
                    hmat = h.mat;
                }
            }
            // ---- refill idle lanes from the queue
            bool need = !active;
            unsigned long long m = __ballot(need);
            if (m && next < total) {
                uint32_t cand = next + lanes_below(m);
                if (need && cand < total) {
                    idx = cand;
                    uint32_t p = count_log2 != 0xFFu ? idx >> count_log2 : (uint32_t)(((float)idx + 0.5f) * inv_count);  // = idx / count exactly (pt_samples_q)
                    const uint32_t sample = fp.first + (idx - p * count);
                    float4 q0 = rec[5 * p], q1 = rec[5 * p + 1], q2 = rec[5 * p + 2], q3 = rec[5 * p + 3],
                           q4 = rec[5 * p + 4];
                    bv = sample * 8049u + xy[4 * p + 2];   // = rnd_base_v(sample, x, y)
                    bu = sample * 2683u + xy[4 * p + 3];   // = rnd_base_u(sample, x, y)
                    uint32_t bits = __float_as_uint(q0.w);
                    if ((bits & 0xFFu) == REC_TREE) {   // the pixel has a shared decision tree: this sample's leaf
                        const float4 *lf = tree_leaf(fp.trees + __float_as_uint(q4.w), sc.table, bu);
                        q0 = lf[0]; q1 = lf[1]; q2 = lf[2]; q3 = lf[3]; q4 = lf[4];
                        bits = __float_as_uint(q0.w);
                    }
                    if ((bits & 0xFFu) == REC_FINAL) {  // a leaf of a tree, or count is not a multiple of g
                        slot[3 * idx] = q3.x;
                        slot[3 * idx + 1] = q3.y;
                        slot[3 * idx + 2] = q3.z;
                    } else {
                        depth = (bits >> 8) & 0xFFu;   // (type and extra_data of the record are the material's: re-read below)
                        hn = xyz(q1);
                        r.o = xyz(q0);
                        r.d = xyz(q2);
                        hmat = __float_as_uint(q2.w);
                        out = xyz(q3);
                        col = xyz(q4);
                        active = true;
                        phase = 0;
                    }
                }
                next += (uint32_t)__popcll(m);
            }
            // ---- state 0: one material interaction, then the primitives that are not models
            if (active && phase == 0) {
                Rnd rnd = fetch_rnd_b(sc.table, r.d, depth, bv, bu);
                Hit at;
                at.p = r.o;
                at.n = hn;
                at.u = at.v = 0.0f;
                at.tex = 0;
                at.mat = hmat;
                int type;
                float extra;
                V3 mcol;
                load_material(c, hmat, type, extra, mcol);
                scatter<false>(c, r, out, at, type, extra, col, rnd, false);
                depth++;
                if (depth >= RT_DEPTH) {  // survived DEPTH bounces: returns what it has (:447,485)
                    slot[3 * idx] = out.x;
                    slot[3 * idx + 1] = out.y;
                    slot[3 * idx + 2] = out.z;
                    active = false;
                } else {
                    Nearest nb;
                    hit_primitives<false, true>(c, r, nb);
                    nb_t = nb.t;
                    nb_id = nb.id;
                    if (MULTI) {
                        nb_face = nb_mat = 0;
                        nb_u = nb_v = 0.0f;
                    }
                    wpos = mesh_walk_start(root0);
                    wbest = faces0;
                    wt = wu = wv = 0.0f;
                    job = 0;
                    phase = 1;
#if PT_W_ROOT_FIRST
                    if (!MULTI) {
                        wpos = mesh_walk_first(sc, r, root0, faces0);
                        if (wpos.cur == PT_MESH_END) phase = 2;
                    } else if (!enter_job()) {
                        phase = 2;
                    }
#endif
                }
            }
#if PT_W_ROOT_FIRST
            if (again >= PT_W_CHEAP_REPEATS) break;
            if ((uint32_t)__popcll(__ballot((active && phase != 1) || (!active && next < total))) < PT_W_CHEAP_AGAIN) break;
#else
            break;
#endif
          }
        }
        // ---- state 1: a slice of the current job's mesh walk
        if (active && phase == 1) {
            uint32_t hits = 0;
            uint32_t mesh_j = mesh0, mat_j = mat0, faces_j = faces0;
            if (MULTI) {
                uint2 jb = jobs[job];
                mesh_j = jb.x;
                mat_j = jb.y;
                faces_j = sc.meshes[mesh_j].face_count;
            }
#ifdef PT_WSTAT
            it_walk_calls++;
            if (mesh_bvh_steps<0>(sc, r, wpos, wbest, wt, wu, wv, PT_WALK_STEPS, hits, nullptr, &ws)) {
#else
            if (mesh_bvh_steps<0>(sc, r, wpos, wbest, wt, wu, wv, PT_WALK_STEPS, hits)) {
#endif
                if (!MULTI) {
                    phase = 2;  // (the one job's result is merged in state 2, straight from the walk's registers)
                } else {
                if (wbest < faces_j && wt < RT_MAX_DISTANCE && wt < nb_t) {  // the running strict-< minimum
                    nb_t = wt;
                    nb_id = K_MESH | mesh_j;
                    nb_face = wbest;
                    nb_mat = mat_j;
                    nb_u = wu;
                    nb_v = wv;
                }
                job++;
#if PT_W_ROOT_FIRST
                if (!enter_job()) phase = 2;   // (the next mesh whose root the ray does not miss: stay in state 1)
#else
                if (MULTI && job < n_jobs) {  // next mesh: stay in state 1
                    uint32_t mesh_n = jobs[job].x;
                    wpos = mesh_walk_start(sc.mesh_bvh_root[mesh_n]);
                    wbest = sc.meshes[mesh_n].face_count;
                    wt = wu = wv = 0.0f;
                } else {
                    phase = 2;
                }
#endif
                }
            }
        }
    }
    if (active) atomicOr(sc.walk_overflow, PT_OVF_WALK_SLICES);   // cold: the outer loop ended on its guard with a path unfinished
#ifdef PT_WSTAT
    if (lane == 0 && npix) {
        unsigned long long v[12] = {it_n, it_active, it_p0, it_p1, it_p2, it_walk_calls, ws.steps, ws.node_lanes, 0ull,
                                    ws.leaf_runs, ws.leaf_lanes, ws.idle_lanes};
        for (int k = 0; k < 12; k++) atomicAdd(&wstat[k], v[k]);
    }
#endif
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- per-pixel sums in pt_render's order
    const uint32_t g = 1u << fp.group_log2, ppp = 64u >> fp.group_log2;
    for (uint32_t pb = 0; pb < npix; pb += ppp) {
        uint32_t p = pb + (lane >> fp.group_log2), l = lane & (g - 1u);
        V3 sum = mk(0.0f, 0.0f, 0.0f);
        if (p < npix)
            for (uint32_t j = l; j < count; j += g) {
                const float *sl = slot + 3u * (p * count + j);
                sum = sum + mk(sl[0], sl[1], sl[2]);
            }
        sum = group_sum(sum, g);
        if (p < npix && l == 0) accumulate(accum, (size_t)xy[4 * p + 1] * fp.w + xy[4 * p], sum, count);
    }
}

// parity probe: one work-item per listed pixel-sample
template <bool ACCEL>
__global__ __launch_bounds__(256) void pt_probe(DeviceScene sc, FrameParams fp, const uint32_t *__restrict__ xs,
                                                const uint32_t *__restrict__ ys, const uint32_t *__restrict__ ss,
                                                uint32_t n, float *__restrict__ out) {
    __shared__ float4 s_mat[PT_LDS_STATIC_FLOAT4];
    Ctx c{sc, stage_materials(sc, s_mat), nullptr};
    c.lwin = staged_winners(sc, s_mat);
    c.lpln = staged_planes(sc, s_mat);
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    Ray r0 = primary_ray(fp.cam, xs[i], ys[i], fp.w, fp.h);
#ifdef PT_PROBE_NODES  // diagnostic build (tools/walk_hist.py): the probe returns (BVH nodes tested, mesh walks, bounces) of the sample
    LaneCounters cn;
    zero_counters(cn);
    c.cn = &cn;
    (void)radiance<true, ACCEL>(c, r0, ss[i], xs[i], ys[i]);
    out[3 * i] = (float)cn.c[CN_DBG_BVH_NODES];
    out[3 * i + 1] = (float)cn.c[CN_T_MESH];
    out[3 * i + 2] = (float)cn.c[CN_BOUNCES];
    return;
#endif
    V3 col = radiance<false, ACCEL>(c, r0, ss[i], xs[i], ys[i]);
    out[3 * i] = col.x;
    out[3 * i + 1] = col.y;
    out[3 * i + 2] = col.z;
}

// ---- unit probes of the device routines (tests only; rt_debug_hit / rt_debug_material / rt_debug_div3) -----------
// One work-item per record; the routines are the very ones the trace kernels inline (hit_primitives' sphere_t /
// plane_t / lens_t, triangle_t, hit_scene + hit_finish, scatter), so a unit vector that matches the oracle here
// pins the arithmetic of the hot loop piece by piece (SURVEY §8c "unit vectors").  Record layouts are those of
// oracle/ref_shim.cpp ref_hit / ref_material.
PT_DEV void put_hit(float *o, bool hit, float t, const Hit &h) {
    for (int k = 0; k < 12; k++) o[k] = 0.0f;
    if (!hit) return;
    o[0] = 1.0f; o[1] = t;
    o[2] = h.p.x; o[3] = h.p.y; o[4] = h.p.z;
    o[5] = h.n.x; o[6] = h.n.y; o[7] = h.n.z;
    o[8] = h.u; o[9] = h.v;
    o[10] = __uint_as_float(h.tex);
    o[11] = __uint_as_float(h.mat);
}
template <bool ACCEL>
__global__ __launch_bounds__(256) void pt_debug_hit(DeviceScene sc, int kind, const float *__restrict__ rays,
                                                    const uint32_t *__restrict__ prim, const uint32_t *__restrict__ face,
                                                    uint32_t n, float *__restrict__ out) {
    __shared__ float4 s_mat[PT_LDS_STATIC_FLOAT4];
    Ctx c{sc, stage_materials(sc, s_mat), nullptr};
    c.lwin = staged_winners(sc, s_mat);
    c.lpln = staged_planes(sc, s_mat);
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    Ray r;
    r.o = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    r.d = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    Hit h;
    h.p = h.n = mk(0.0f, 0.0f, 0.0f);
    h.u = h.v = 0.0f;
    h.tex = h.mat = 0;
    Nearest nb;
    bool hit = false;
    if (kind == 3) {                      // hitScene :322-360
        hit = hit_scene<false, ACCEL>(c, r, h);
        if (hit) { hit_primitives<false, ACCEL>(c, r, nb); hit_models<false, ACCEL>(c, r, nb); }
    } else {
        // a single primitive through the SAME (t, id) search + winner rebuild the trace kernels use
        uint32_t p = prim[i];
        float t = PT_MISS;
        if (kind == 0) { const rt_sphere &sp = sc.spheres[p]; t = sphere_t(r, make_float4(sp.pos.x, sp.pos.y, sp.pos.z, sp.r * sp.r)); nb.id = K_SPHERE | p; }
        else if (kind == 1) { const rt_plane &pl = sc.planes[p]; t = plane_t(r, ld3(pl.pos), ld3(pl.normal)); nb.id = K_PLANE | p; }
        else if (kind == 2) { int which; t = lens_t(r, sc.lenses[p], &which); nb.id = K_LENS | p; }
        else if (kind == 4) {             // hitTriangle :257-289 on face face[i] of mesh p
            const float4 *fr = sc.faces + 3u * ((size_t)sc.mesh_face_base[p] + face[i]);
            float4 q0 = fr[0], q1 = fr[1];
            float4 q2 = fr[2];
            float u, v;
            t = triangle_t(r, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x), &u, &v);
            nb.id = K_MESH | p; nb.face = face[i]; nb.u = u; nb.v = v; nb.mat = 0;
        }
        if (t < PT_MISS) {
            nb.t = t;
            hit = hit_finish<false>(c, r, nb, h);
            if (kind == 4) h.mat = h.tex = 0;  // hitTriangle sets neither mat_ID (hitModel does, :314) nor texture_ID (hitMeshOut, :299)
        }
    }
    put_hit(out + 12 * (size_t)i, hit, nb.t, h);
}

__global__ __launch_bounds__(256) void pt_debug_material(DeviceScene sc, int routine, const float *__restrict__ in,
                                                         uint32_t n, float *__restrict__ out) {
    __shared__ float4 s_mat[PT_LDS_STATIC_FLOAT4];
    Ctx c{sc, stage_materials(sc, s_mat), nullptr};
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float *v = in + 16 * (size_t)i;
    Ray r;
    r.o = mk(0.0f, 0.0f, 0.0f);
    r.d = mk(v[0], v[1], v[2]);
    Hit h;
    h.p = mk(v[3], v[4], v[5]);
    h.n = mk(v[6], v[7], v[8]);
    h.u = h.v = 0.0f;
    h.tex = 0;
    h.mat = __float_as_uint(v[12]);
    V3 out_col = mk(v[9], v[10], v[11]);
    uint32_t seed = __float_as_uint(v[13]), gx = __float_as_uint(v[14]), gy = __float_as_uint(v[15]);
    int type;
    float extra;
    V3 col;
    load_material(c, h.mat, type, extra, col);
    // the routine under test decides the branch of scatter(); the material supplies extra_data and — for
    // rayReflect's "*= extra only if t_reflective" (:366) — its own type
    int as_type = routine == 0 ? (type == RT_REFLECTIVE ? RT_REFLECTIVE : -1) : routine == 1 ? RT_REFRACTIVE
                  : routine == 2 ? RT_DIFFUSE : RT_DIELECTRIC;
    Rnd rnd = fetch_rnd(sc.table, r.d, seed, gx, gy);
    if (as_type == -1) {   // rayReflect on a material that is not t_reflective: the reflection tail of scatter()
        float k = 2.0f * dot(r.d, h.n);
        r.o = h.p;
        r.d = normalize(nmad(h.n, k, r.d));
    } else {
        scatter<false>(c, r, out_col, h, as_type, extra, mk(INFINITY, INFINITY, INFINITY), rnd);  // mixCol with +inf = identity
    }
    float *o = out + 9 * (size_t)i;
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z;
    o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
    o[6] = out_col.x; o[7] = out_col.y; o[8] = out_col.z;
}

// div3 (shared-reciprocal form of three IEEE divisions) against the compiler's divisions: in n × 4 {a.xyz, d} →
// out n × 6 {div3 result, a / d}; `force` = 1 runs the shared-reciprocal sequence even when PT_DIV3 is off
__global__ __launch_bounds__(256) void pt_debug_div3(const float *__restrict__ in, uint32_t n, float *__restrict__ out) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    V3 a = mk(in[4 * i], in[4 * i + 1], in[4 * i + 2]);
    float d = in[4 * i + 3];
    V3 q = a / d, s = q;
    if (div3_in_range(a, d)) {
        float nd = -d, rr = __builtin_amdgcn_rcpf(d);
        float e = __builtin_fmaf(nd, rr, 1.0f);
        rr = __builtin_fmaf(e, rr, rr);
        s = V3{div_shared(a.x, nd, rr), div_shared(a.y, nd, rr), div_shared(a.z, nd, rr)};
    }
    float *o = out + 6 * (size_t)i;
    o[0] = s.x; o[1] = s.y; o[2] = s.z; o[3] = q.x; o[4] = q.y; o[5] = q.z;
}

// ================================== launchers ==================================

#define PT_DISPATCH(count_on, accel_on, CALL)                           \
    do {                                                                \
        if (count_on) { if (accel_on) CALL(true, true); else CALL(true, false); }   \
        else { if (accel_on) CALL(false, true); else CALL(false, false); }          \
    } while (0)

// Direct path (every sample from the camera): trace / retrace compat modes, and the
// fused mode when prefix sharing is switched off.
template <int MODE>
int launch_render(rt_context *ctx, const float cam[12], uint32_t first, uint32_t count, uint32_t glog2) {
    FrameParams fp = frame_params(ctx, cam, first, count, glog2);
    DeviceScene sc = device_scene(ctx);
    uint32_t slots = fp.slot_end;
    if (slots == 0) return RT_OK;
    // split very long launches into slot ranges (keeps single kernels short on huge scenes)
    uint32_t slots_per_launch = ctx->max_threads_per_launch >> glog2;
    if (slots_per_launch == 0) slots_per_launch = 1;
    hipEvent_t *evp = ctx->ev[ctx->ev_count % rt_context::EV_RING];
    HIP_TRY(ctx, hipEventRecord(evp[0], ctx->stream));
    HIP_TRY(ctx, hipEventRecord(evp[2], ctx->stream));  // no first stage on the direct path
    for (uint32_t b = 0; b < slots; b += slots_per_launch) {
        fp.slot_begin = b;
        fp.slot_end = b + slots_per_launch < slots ? b + slots_per_launch : slots;
        uint64_t threads = (uint64_t)(fp.slot_end - fp.slot_begin) << glog2;
        dim3 grid((unsigned)((threads + 255) / 256)), block(256);
#define PT_CALL(C, A) \
    hipLaunchKernelGGL((pt_render<MODE, C, A>), grid, block, 0, ctx->stream, sc, fp, ctx->d_accum, ctx->d_image, ctx->d_counters)
        PT_DISPATCH(ctx->count_enabled, scene_has_accel(sc), PT_CALL);
#undef PT_CALL
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(evp[1], ctx->stream));
    ctx->ev_count++;
    return RT_OK;
}

// Fused path: pt_prefix (one work-item per pixel) + pt_samples (g lanes per live pixel).
int launch_fused(rt_context *ctx, const float cam[12], uint32_t first, uint32_t count, uint32_t glog2) {
    FrameParams fp = frame_params(ctx, cam, first, count, glog2);
    DeviceScene sc = device_scene(ctx);
    uint32_t slots = fp.slot_end;
    if (slots == 0) return RT_OK;
    int rc = ensure_slots(ctx, slots);
    if (rc) return rc;
    uint32_t *live_count = ctx->d_live + ctx->slot_capacity + (size_t)LIVE_SEGMENTS * 256u;
    uint32_t slots_per_launch = ctx->max_threads_per_launch >> glog2;
    if (slots_per_launch == 0) slots_per_launch = 1;
    hipEvent_t *evp = ctx->ev[ctx->ev_count % rt_context::EV_RING];
    HIP_TRY(ctx, hipEventRecord(evp[0], ctx->stream));
    for (uint32_t b = 0; b < slots; b += slots_per_launch) {
        fp.slot_begin = b;
        fp.slot_end = b + slots_per_launch < slots ? b + slots_per_launch : slots;
        uint32_t n = fp.slot_end - fp.slot_begin;
        HIP_TRY(ctx, hipMemsetAsync(live_count, 0, (size_t)LIVE_SEGMENTS * LIVE_COUNT_STRIDE * sizeof(uint32_t), ctx->stream));
        // shared decision trees (RT_OPT_PREFIX_TREE): not in counting builds — the counters price per-sample work
        fp.trees = ctx->d_trees;
        fp.glass = ctx->d_glass;
        fp.tree_count = live_count + LIVE_TREE_COUNTER;
        // (and not for a handful of samples per call: tracing both continuations of a pixel costs more than the few samples
        // that would share them — C2 at 1 / 8 / 16 / 32 samples per call: 0.321 / 0.457 / 0.609 / 1.031 ms with trees,
        // 0.243 / 0.393 / 0.582 / 1.088 without)
        const bool tree_on = ctx->prefix_tree == 2 || (ctx->prefix_tree == 1 && count >= PT_TREE_MIN_SAMPLES);
        fp.tree_cap = (tree_on && !ctx->count_enabled && ctx->d_trees && ctx->d_tree_work) ? (uint32_t)ctx->tree_capacity : 0u;
        // workgroup b of pt_prefix appends to segment b mod LIVE_SEGMENTS: a segment holds at most seg_cap entries
        const uint32_t prefix_blocks = (n + 255) / 256;
        fp.seg_cap = ((prefix_blocks + LIVE_SEGMENTS - 1) / LIVE_SEGMENTS) * 256u;
        dim3 block(256), grid1(prefix_blocks);
        // sample queue: a wave owns ppw live pixels (<= QUEUE_SLOTS samples); worst case all n pixels are live
        uint32_t static_f4 = lds_static_used(sc.material_count, sc.sphere_count, sc.plane_count) +
                             (PT_LDS_SPHERES ? PT_LDS_SPHERE_CAP : 0);
        const bool sphere_bvh_only = sc.bvh_node_count != 0 && sc.mesh_bvh_root == nullptr;
        const bool simple_geom = sc.lens_count == 0 && sc.model_count == 0;   // spheres and planes only
        const uint32_t q_waves = !scene_has_accel(sc) ? PT_Q_WAVES : ((sphere_bvh_only && simple_geom) ? PT_Q_WAVES_SPHERE_BVH : PT_Q_WAVES_ACCEL);
        uint32_t ppw = queue_pixels_per_wave(count, q_waves, static_f4, PT_Q_BLOCK_WAVES);
        // Small launches (small frames, a rank's share of a sharded frame): fewer pixels per wave, so that there are about
        // PT_UNITS_PER_WAVE_SLOT waves per wave slot of the chip — a wave works through its pixels' samples one batch of
        // 64 after the other, and 4 200 waves of 384 samples leave a third of the slots empty for the whole launch.
        // C2 at 64 spp, 6 pixels per wave vs this rule: 200 x 126 0.314 → 0.140 ms, 320 x 180 0.336 → 0.187, 480 x 270
        // 0.344 → 0.268, 640 x 360 0.419 → 0.369, 960 x 540 0.619 → 0.59; 1080p and up unchanged (6).
        const uint32_t want_units = (uint32_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 4u * 6u * PT_UNITS_PER_WAVE_SLOT;
        // (never fewer than the 64 samples that fill a wave's lanes once)
        const uint32_t ppw_full = (64u + count - 1u) / count;
        const uint32_t ppw_par = !ctx->wave_fill ? QUEUE_MAX_PIXELS : std::max(n / want_units, ppw_full);
        if (ppw > ppw_par) ppw = ppw_par;
        // Face records in LDS for hit_models' candidate loop: scenes whose meshes are all face-scanned (no mesh BVH) and
        // hold at most PT_LDS_FACE_CAP faces together, and only when the copy fits into what the 1 KiB allocation granule
        // leaves over anyway (C3: 576 bytes of a cube into 609 spare ones) — never at the price of a pixel per wave.
        fp.lds_face_f4 = 0u;
        if (PT_FACE_MASK && !simple_geom && sc.mesh_bvh_root == nullptr && !ctx->count_enabled) {
            const size_t nf = ctx->h_faces.size() / 3u - (ctx->h_faces.empty() ? 0u : 1u);   // (the array ends with one dummy record)
            if (nf > 0 && nf <= PT_LDS_FACE_CAP &&
                queue_pixels_per_wave(count, q_waves, static_f4 + 3u * (uint32_t)nf, PT_Q_BLOCK_WAVES) >= ppw) {
                fp.lds_face_f4 = 3u * (uint32_t)nf;
                static_f4 += fp.lds_face_f4;
            }
        }
        dim3 blockq(64 * PT_Q_BLOCK_WAVES);
        bool queue = ctx->sample_queue && count <= QUEUE_SLOTS;
        size_t lds_q = static_f4 * sizeof(float4) + PT_Q_BLOCK_WAVES * (size_t)queue_wave_lds_bytes(ppw, count);
#define PT_CALL_PREFIX(C, A) \
    hipLaunchKernelGGL((pt_prefix<C, A>), grid1, block, 0, ctx->stream, sc, fp, ctx->d_recs, ctx->d_live, live_count, ctx->d_accum, ctx->d_counters)
        bool accel_on = scene_has_accel(sc);
        PT_DISPATCH(ctx->count_enabled, accel_on, PT_CALL_PREFIX);
        if (fp.tree_cap) {
            // stage 1b: the decision trees of the glass-first pixels, level by level (grid-stride over device-side lists;
            // a modest grid: 4096 workgroups that stage the materials and find next to nothing to do cost 0.06 ms each).
            // These launches are latency-bound — each lasts as long as its longest stretch of path, 30-60 us with a few
            // thousand waves in flight (PT_TREE_STRETCH bounds it).  Tried and measured slower (profiles/r03_experiments.md):
            // the glass-first pixels in a list of their own, trees and a second launch of the sample kernel on a side
            // stream beside the main list's — a queue-kernel wave lives ~0.1 ms whatever the size of its launch.
            dim3 gridt(std::min<uint32_t>(768u, (2u * fp.tree_cap + 255u) / 256u));
            const uint32_t q_cap = (uint32_t)ctx->tree_capacity;
            TreeWork *q[2] = {ctx->d_tree_work, ctx->d_tree_work + ctx->tree_capacity};
            for (uint32_t level = 0; level < PT_TREE_LEVELS; level++) {
                const TreeWork *in = level ? q[(level - 1u) & 1u] : nullptr;
                const uint32_t *in_count = level ? fp.tree_count + level : nullptr;
                TreeWork *out = q[level & 1u];
                uint32_t *out_count = fp.tree_count + level + 1u;
                if (accel_on) hipLaunchKernelGGL(pt_tree_pass<true>, gridt, block, 0, ctx->stream, sc, fp, ctx->d_recs, level, in, in_count, out, out_count, q_cap);
                else hipLaunchKernelGGL(pt_tree_pass<false>, gridt, block, 0, ctx->stream, sc, fp, ctx->d_recs, level, in, in_count, out, out_count, q_cap);
            }
        }
        HIP_TRY(ctx, hipEventRecord(evp[2], ctx->stream));  // (the last slot range's; one range is the normal case)
        // the sample kernel over a list: (records, slots, its counter, its capacity)
        auto launch_samples = [&](hipStream_t st, const PixelRec *l_recs, const uint32_t *l_live, const uint32_t *l_count, uint32_t l_cap) {
            FrameParams fl = fp;
            fl.seg_cap = l_cap;
            // (the two parts of the list each end in a partial chunk: one unit more than capacity / chunk)
            auto units = [&](uint32_t per_unit) { return (l_cap + per_unit - 1) / per_unit + 1u; };
            dim3 gridq((units(ppw) + PT_Q_BLOCK_WAVES - 1) / PT_Q_BLOCK_WAVES);
            dim3 grid2((unsigned)((((uint64_t)units(1u) << glog2) + 255) / 256));
#define PT_CALL_QUEUE_W(C, A, G, W) \
    hipLaunchKernelGGL((pt_samples_q<C, A, G, W>), gridq, blockq, lds_q, st, sc, fl, l_recs, l_live, l_count, ctx->d_accum, ctx->d_counters, ppw)
#define PT_CALL_QUEUE(C, A)                                                                       \
    do {                                                                                          \
        if (!(A)) { if (simple_geom) PT_CALL_QUEUE_W(C, false, 0, PT_Q_WAVES); else PT_CALL_QUEUE_W(C, false, 1, PT_Q_WAVES); } \
        else if (sphere_bvh_only) { if (simple_geom) PT_CALL_QUEUE_W(C, true, 0, PT_Q_WAVES_SPHERE_BVH); else PT_CALL_QUEUE_W(C, true, 1, PT_Q_WAVES_ACCEL); } \
        else PT_CALL_QUEUE_W(C, true, 2, PT_Q_WAVES_ACCEL);                                       \
    } while (0)
#define PT_CALL_FIXED(C, A) \
    hipLaunchKernelGGL((pt_samples<C, A>), grid2, block, 0, st, sc, fl, l_recs, l_live, l_count, ctx->d_accum, ctx->d_counters)
            if (queue && sc.mesh_bvh_root && ctx->walk_jobs.n && ctx->walk_slices && !ctx->count_enabled && !PT_LDS_SPHERES) {
                // every mesh has a BVH: interleaved walk slices (pt_samples_w), sized for its own occupancy target
                uint32_t ppw_w = queue_pixels_per_wave(count, ctx->walk_jobs.n == 1 ? PT_W_WAVES : PT_W_WAVES_MULTI, static_f4, PT_W_BLOCK_WAVES);
                if (ppw_w > ppw_par) ppw_w = ppw_par;
                size_t lds_w = static_f4 * sizeof(float4) + PT_W_BLOCK_WAVES * (size_t)queue_wave_lds_bytes(ppw_w, count);
                dim3 gridw((units(ppw_w) + PT_W_BLOCK_WAVES - 1) / PT_W_BLOCK_WAVES), blockw(64 * PT_W_BLOCK_WAVES);
                if (ctx->walk_jobs.n == 1)
                    hipLaunchKernelGGL(pt_samples_w<false>, gridw, blockw, lds_w, st, sc, fl, l_recs, l_live, l_count,
                                       ctx->d_accum, ppw_w, ctx->walk_jobs.p, 1u
#ifdef PT_WSTAT
                                       , ctx->d_counters + COUNTER_REPLICAS * COUNTER_STRIDE + 8
#endif
                                       );
                else
                    hipLaunchKernelGGL(pt_samples_w<true>, gridw, blockw, lds_w, st, sc, fl, l_recs, l_live, l_count,
                                       ctx->d_accum, ppw_w, ctx->walk_jobs.p, (uint32_t)ctx->walk_jobs.n
#ifdef PT_WSTAT
                                       , ctx->d_counters + COUNTER_REPLICAS * COUNTER_STRIDE + 8
#endif
                                       );
            } else if (queue) PT_DISPATCH(ctx->count_enabled, accel_on, PT_CALL_QUEUE);
            else PT_DISPATCH(ctx->count_enabled, accel_on, PT_CALL_FIXED);
#undef PT_CALL_QUEUE
#undef PT_CALL_QUEUE_W
#undef PT_CALL_FIXED
        };
        launch_samples(ctx->stream, ctx->d_recs, ctx->d_live, live_count, fp.seg_cap);
#undef PT_CALL_PREFIX
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(evp[1], ctx->stream));
    ctx->ev_count++;
    return RT_OK;
}

// ---- policy-dependent precomputation and probes ---------------------------------------------------------------
// hitTriangle's unit normal, normalize(cross(edge1, edge2)) (raytracer.cl:285), depends on the face only: it is kept
// in the face records (A, e1, e2, n).  Policy 0 computes it on the host (rt_amd.hip build_face_records: plain IEEE
// operations); the ROCm-OpenCL policies need the library's cross and its v_rsq_f32-based normalize, which only
// the device can evaluate — one work-item per record, the same two calls the reference makes per test.
__global__ __launch_bounds__(256) void pt_face_normals(float4 *__restrict__ rec, uint32_t n) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 q0 = rec[3 * (size_t)i], q1 = rec[3 * (size_t)i + 1], q2 = rec[3 * (size_t)i + 2];
    V3 e1 = mk(q0.w, q1.x, q1.y), e2 = mk(q1.z, q1.w, q2.x);
    V3 nn = normalize(cross(e1, e2));
    rec[3 * (size_t)i + 2] = make_float4(q2.x, nn.x, nn.y, nn.z);
}

// One builtin of this policy per record (tests/test_gpu_ref950.py compares policies 1 / 2 with probe kernels that call
// ROCm's OpenCL builtins themselves, oracle/ref_gfx950_wrap.cl): in n × 8 floats, out n × 4 floats.
__global__ __launch_bounds__(256) void pt_debug_builtin(int op, const float *__restrict__ in, uint32_t n, float *__restrict__ out) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float *a = in + 8 * (size_t)i;
    V3 x = mk(a[0], a[1], a[2]), y = mk(a[3], a[4], a[5]);
    float t = a[6];
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (op == 0) o.x = dot(x, y);
    else if (op == 1) { V3 c = cross(x, y); o = make_float4(c.x, c.y, c.z, 0.0f); }
    else if (op == 2) { V3 c = normalize(x); o = make_float4(c.x, c.y, c.z, 0.0f); }
    else if (op == 3) { o.x = a[0] / a[1]; o.y = 1.0f / a[0]; V3 c = x / t; o.z = c.y; o.w = c.z; }
    else if (op == 4) o.x = sqrt1(a[0]);
    else if (op == 5) o = make_float4(mix1(x.x, y.x, t), mix1(x.y, y.y, t), mix1(x.z, y.z, t), 0.0f);
    else if (op == 6) { V3 c = vmin(x, y); o = make_float4(c.x, c.y, c.z, 0.0f); }
    else if (op == 7) o.x = sign1(a[0]);
    else if (op == 8) o.x = pow5(a[0]);
    else if (op == 9) o.x = __uint_as_float(dir_hash(x));
    reinterpret_cast<float4 *>(out)[i] = o;
}

namespace {

int ks_launch_render(rt_context *ctx, int mode, const float cam[12], uint32_t first, uint32_t count, uint32_t glog2) {
    switch (mode) {
        case MODE_ACCUM: return launch_render<MODE_ACCUM>(ctx, cam, first, count, glog2);
        case MODE_TRACE: return launch_render<MODE_TRACE>(ctx, cam, first, count, glog2);
        case MODE_RETRACE: return launch_render<MODE_RETRACE>(ctx, cam, first, count, glog2);
        default: return fail(ctx, RT_EINVAL, "unknown render mode %d", mode);
    }
}

int ks_launch_probe(rt_context *ctx, const FrameParams &fp, const DeviceScene &sc, const uint32_t *d_in, uint32_t n, float *d_out) {
    dim3 grid((n + 255u) / 256u), block(256);
    if (scene_has_accel(sc))
        hipLaunchKernelGGL(pt_probe<true>, grid, block, 0, ctx->stream, sc, fp, d_in, d_in + n, d_in + 2 * (size_t)n, n, d_out);
    else
        hipLaunchKernelGGL(pt_probe<false>, grid, block, 0, ctx->stream, sc, fp, d_in, d_in + n, d_in + 2 * (size_t)n, n, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

int ks_launch_debug_hit(rt_context *ctx, const DeviceScene &sc, int kind, const float *d_rays, const uint32_t *d_prim,
                        const uint32_t *d_face, uint32_t n, float *d_out) {
    dim3 grid((n + 255u) / 256u), block(256);
    if (scene_has_accel(sc))
        hipLaunchKernelGGL(pt_debug_hit<true>, grid, block, 0, ctx->stream, sc, kind, d_rays, d_prim, d_face, n, d_out);
    else
        hipLaunchKernelGGL(pt_debug_hit<false>, grid, block, 0, ctx->stream, sc, kind, d_rays, d_prim, d_face, n, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

int ks_launch_debug_material(rt_context *ctx, const DeviceScene &sc, int routine, const float *d_in, uint32_t n, float *d_out) {
    hipLaunchKernelGGL(pt_debug_material, dim3((n + 255u) / 256u), dim3(256), 0, ctx->stream, sc, routine, d_in, n, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

int ks_launch_debug_div3(rt_context *ctx, const float *d_in, uint32_t n, float *d_out) {
    hipLaunchKernelGGL(pt_debug_div3, dim3((n + 255u) / 256u), dim3(256), 0, ctx->stream, d_in, n, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

int ks_launch_face_normals(rt_context *ctx, float4 *d_records, uint32_t n_records) {
    if (n_records == 0) return RT_OK;
    hipLaunchKernelGGL(pt_face_normals, dim3((n_records + 255u) / 256u), dim3(256), 0, ctx->stream, d_records, n_records);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

int ks_launch_debug_builtin(rt_context *ctx, int op, const float *d_in, uint32_t n, float *d_out) {
    hipLaunchKernelGGL(pt_debug_builtin, dim3((n + 255u) / 256u), dim3(256), 0, ctx->stream, op, d_in, n, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

const pt::KernelSet g_kernel_set = {
    PT_ARITH,
#if PT_ARITH == 0
    "ieee",
#elif PT_ARITH == 1
    "rocm-opencl-nocontract",
#else
    "rocm-opencl",
#endif
    ks_launch_render, launch_fused, ks_launch_probe, ks_launch_debug_hit, ks_launch_debug_material, ks_launch_debug_div3,
    ks_launch_face_normals, ks_launch_debug_builtin};

}  // namespace

}  // namespace PT_NS

namespace pt {
#if PT_ARITH == 0
const KernelSet *kernel_set_a0() { return &pt_a0::g_kernel_set; }
#elif PT_ARITH == 1
const KernelSet *kernel_set_a1() { return &pt_a1::g_kernel_set; }
#else
const KernelSet *kernel_set_a2() { return &pt_a2::g_kernel_set; }
#endif
}  // namespace pt
