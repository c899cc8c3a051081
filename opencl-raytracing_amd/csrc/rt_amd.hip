// rt_amd.hip — kernels + C ABI of librt_amd.so (include/rt_amd.h), gfx950 only.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
// (see __graft_entry__.build()).  No CPU fallback exists: without a gfx950
// device rt_create() fails.
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

#include "pt_device.hpp"
#include "mesh_bvh_build.hpp"

using namespace pt;

// =============================== device kernels ===============================

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
// wave (or pixel group) `unit` of a sample kernel → its segment, its first entry and how many of `want` exist
PT_DEV uint32_t live_take(const FrameParams &fp, const uint32_t *__restrict__ live_count, uint32_t unit, uint32_t want,
                          uint32_t &first) {
    uint32_t seg = unit % LIVE_SEGMENTS, start = (unit / LIVE_SEGMENTS) * want;
    uint32_t cnt = live_count[seg * LIVE_COUNT_STRIDE];
    first = seg * fp.seg_cap + start;
    return start < cnt ? min(want, cnt - start) : 0u;
}

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

// Counters are spread over COUNTER_REPLICAS rows (one per workgroup residue) so
// that two million waves do not serialise on 14 addresses; the host sums the rows.
#define COUNTER_REPLICAS 512
#define COUNTER_STRIDE 16
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
            image[pix] = make_float4(sqrtf(sum.x), sqrtf(sum.y), sqrtf(sum.z), 1.0f);
        } else {
            if (COUNT) cn.c[CN_IMAGE_READS]++;
            float4 prev = image[pix];
            V3 lin = mk(prev.x * prev.x, prev.y * prev.y, prev.z * prev.z);
            float k = (float)fp.first / (float)(fp.first + 1u);
            V3 o = mk(sum.x + (lin.x - sum.x) * k, sum.y + (lin.y - sum.y) * k, sum.z + (lin.z - sum.z) * k);
            image[pix] = make_float4(sqrtf(o.x), sqrtf(o.y), sqrtf(o.z), 1.0f);
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
        uint32_t g = 1u << fp.group_log2;
        bool final_px = (__float_as_uint(rec.p_kind.w) & 0xFFu) == REC_FINAL;
        if (final_px && (fp.count & (g - 1u)) == 0) {
            V3 col = xyz(rec.out), sum = mk(0.0f, 0.0f, 0.0f);
            for (uint32_t k = 0; k < (fp.count >> fp.group_log2); k++) sum = sum + col;
            for (uint32_t off = g >> 1; off > 0; off >>= 1) sum = sum + sum;
            accumulate(accum, (size_t)y * fp.w + x, sum, fp.count);
            if (COUNT) cn.c[CN_SAMPLES] += 1;  // scaled by count below
        } else {
            is_live = true;
        }
    }
    // append live pixels — slot index and record, both at the pixel's position in the live list, so the
    // sample kernels read records without an indirection.  ONE atomic per WORKGROUP: the four waves' counts meet
    // in LDS, thread 0 reserves the workgroup's run, each wave takes its part of it.  (One atomic per wave made
    // 32 400 waves of a 1080p frame queue on a single address: 0.12 of the kernel's 0.20 ms.  Spreading the list
    // over LIVE_SEGMENTS > 1 independent counters removes the queue too, but costs pt_samples_q 13–26 %: the
    // list's ORDER matters to it — see LIVE_SEGMENTS.)  Order within the list is irrelevant to the result.
    __shared__ uint32_t s_wave_n[4], s_base;
    unsigned long long m = __ballot(is_live);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (lane == 0) s_wave_n[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = s_wave_n[0] + s_wave_n[1] + s_wave_n[2] + s_wave_n[3];
        s_base = total ? atomicAdd(&live_count[(blockIdx.x % LIVE_SEGMENTS) * LIVE_COUNT_STRIDE], total) : 0u;
    }
    __syncthreads();
    if (is_live) {
        uint32_t before = 0;
        for (uint32_t k = 0; k < wv; k++) before += s_wave_n[k];
        uint32_t pos = (blockIdx.x % LIVE_SEGMENTS) * fp.seg_cap + s_base + before +
                       lanes_below(m);
        live[pos] = slot;
        recs[pos] = rec;
    }
    flush_counters<COUNT>(cn, counters, fp.count);  // the prefix stands for `count` samples' worth of work
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
            sum = sum + radiance_from_rec<COUNT, ACCEL>(c, rec, s, x, y);
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
    uint32_t b = pixels_per_wave * 5u * 16u + pixels_per_wave * 2u * 4u + pixels_per_wave * count * 3u * 4u;
    return (b + 15u) & ~15u;
}
#ifndef PT_UNIFORM_WAVE
#define PT_UNIFORM_WAVE 1
#endif
#ifndef PT_Q_WAVES
#define PT_Q_WAVES 6  // waves per SIMD the register allocator must leave room for: 6 = 80 VGPRs (A/B on C2: 5 → 2.62 ms, 6 → 2.48)
#endif
#ifndef PT_Q_WAVES_ACCEL
#define PT_Q_WAVES_ACCEL 5  // scenes that mix BVH meshes with small ones (every other mesh scene runs pt_samples_w): 96 VGPRs, 2 spilled (6: 22 spilled)
#endif
#ifndef PT_Q_WAVES_SPHERE_BVH
#define PT_Q_WAVES_SPHERE_BVH 6  // scenes whose only BVH is the sphere BVH (C4 at 8 spp, r02: 5 → 76.9 ms, 6 → 71.8 ms)
#endif
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
        uint32_t p = per_wave / (5u * 16u + 2u * 4u + count * 3u * 4u);
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

#if PT_Q_BLOCK_WAVES == 1
    const uint32_t wave = 0u, lane = threadIdx.x;
#elif PT_UNIFORM_WAVE
    // the wave index is wave-uniform, which the compiler cannot see: this puts everything derived from it in SGPRs
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
#else
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
#endif
    char *wave_lds = reinterpret_cast<char *>(s_dyn + lds_static_used(sc.material_count, sc.sphere_count, sc.plane_count) +
                                              (PT_LDS_SPHERES ? PT_LDS_SPHERE_CAP : 0)) +
                     (size_t)wave * queue_wave_lds_bytes(pixels_per_wave, fp.count);
    float4 *s_rec = reinterpret_cast<float4 *>(wave_lds);
    uint32_t *s_xy = reinterpret_cast<uint32_t *>(s_rec + pixels_per_wave * 5u);
    float *slot = reinterpret_cast<float *>(s_xy + pixels_per_wave * 2u);
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
        s_xy[2 * lane] = x;
        s_xy[2 * lane + 1] = y;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const float inv_count = fp.inv_count;
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
                uint32_t p = (uint32_t)(((float)idx + 0.5f) * inv_count);
                const uint32_t sample = fp.first + (idx - p * count);
                float4 q0 = rec[5 * p], q1 = rec[5 * p + 1], q2 = rec[5 * p + 2], q3 = rec[5 * p + 3],
                       q4 = rec[5 * p + 4];
                const uint32_t gx = xy[2 * p], gy = xy[2 * p + 1];
                bv = rnd_base_v(sample, gx, gy);
                bu = rnd_base_u(sample, gx, gy);
                uint32_t bits = __float_as_uint(q0.w);
                if (COUNT) cn.c[CN_SAMPLES]++;
                if ((bits & 0xFFu) == REC_FINAL) {  // only when count is not a multiple of g
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
#ifdef PT_EXP_PAD  // timing experiment: PT_EXP_PAD extra independent full-rate VALU instructions per iteration — an
                   // issue-bound loop slows down in proportion, a latency-bound one does not
#pragma unroll
        for (int k = 0; k < PT_EXP_PAD / 4; k++)
            asm volatile("v_or_b32 %0, %0, %0\n\tv_or_b32 %0, %0, %0\n\tv_or_b32 %0, %0, %0\n\tv_or_b32 %0, %0, %0" : "+v"(idx));  // identity on a live register: no extra VGPR
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
        if (p < npix && l == 0) accumulate(accum, (size_t)xy[2 * p + 1] * fp.w + xy[2 * p], sum, count);
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
#ifndef PT_W_WAVES
#define PT_W_WAVES 5  // A/B on C5 at 16 spp: 4 → 116.5 ms, 5 → 110.2, 6 → 116.1
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
    float *slot = reinterpret_cast<float *>(s_xy + pixels_per_wave * 2u);
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
        s_xy[2 * lane] = x;
        s_xy[2 * lane + 1] = y;
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

#ifdef PT_WSTAT
    WalkStat ws = {0, 0, 0, 0, 0};
    unsigned long long it_n = 0, it_active = 0, it_p0 = 0, it_p1 = 0, it_p2 = 0, it_walk_calls = 0;
#endif
    // every iteration takes samples off the queue, or moves every active lane on (a bounce, or up to
    // PT_WALK_STEPS nodes of a walk that visits each of the < 2^28 nodes at most 3 times)
    for (unsigned long long guard = ((unsigned long long)total + 1ull) * (RT_DEPTH + 2ull) * (3ull * (1ull << 28) / PT_WALK_STEPS + 4ull); guard; guard--) {
        // ---- refill idle lanes from the queue
        bool need = !active;
        unsigned long long m = __ballot(need);
        if (m && next < total) {
            uint32_t cand = next + lanes_below(m);
            if (need && cand < total) {
                idx = cand;
                uint32_t p = (uint32_t)(((float)idx + 0.5f) * inv_count);  // = idx / count exactly (pt_samples_q)
                const uint32_t sample = fp.first + (idx - p * count);
                float4 q0 = rec[5 * p], q1 = rec[5 * p + 1], q2 = rec[5 * p + 2], q3 = rec[5 * p + 3],
                       q4 = rec[5 * p + 4];
                const uint32_t gx = xy[2 * p], gy = xy[2 * p + 1];
                bv = rnd_base_v(sample, gx, gy);
                bu = rnd_base_u(sample, gx, gy);
                uint32_t bits = __float_as_uint(q0.w);
                if ((bits & 0xFFu) == REC_FINAL) {  // only when count is not a multiple of g
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
        if (!__any(active)) {
            if (next >= total) break;
            continue;
        }
#ifdef PT_WSTAT
        it_n++;
        it_active += __popcll(__ballot(active));
        it_p0 += __popcll(__ballot(active && phase == 0));
        it_p1 += __popcll(__ballot(active && phase == 1));
        it_p2 += __popcll(__ballot(active && phase == 2));
#endif
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
                if (MULTI && job < n_jobs) {  // next mesh: stay in state 1
                    uint32_t mesh_n = jobs[job].x;
                    wpos = mesh_walk_start(sc.mesh_bvh_root[mesh_n]);
                    wbest = sc.meshes[mesh_n].face_count;
                    wt = wu = wv = 0.0f;
                } else {
                    phase = 2;
                }
                }
            }
        }
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
    }
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
        if (p < npix && l == 0) accumulate(accum, (size_t)xy[2 * p + 1] * fp.w + xy[2 * p], sum, count);
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
        r.d = normalize(r.d - h.n * k);
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

#define ACCEL_MIN_SPHERES 64
#define MESH_BVH_MIN_FACES 32

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

}  // namespace

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
    PixelRec *d_recs = nullptr;      // per owned pixel slot: shared path prefix (fused path)
    uint32_t *d_live = nullptr;      // slots that need per-sample work + [capacity] = their count
    size_t slot_capacity = 0;
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

namespace {

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

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(ctx, RT_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int alloc_frame(rt_context *ctx, int w, int h) {
    if (ctx->d_image) (void)hipFree(ctx->d_image);
    if (ctx->d_accum) (void)hipFree(ctx->d_accum);
    ctx->d_image = ctx->d_accum = nullptr;
    if (ctx->d_recs) (void)hipFree(ctx->d_recs);
    if (ctx->d_live) (void)hipFree(ctx->d_live);
    ctx->d_recs = nullptr;
    ctx->d_live = nullptr;
    ctx->slot_capacity = 0;
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

DeviceScene device_scene(const rt_context *ctx) {
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
Shard shard_of(const rt_context *ctx, int rank, int world) {
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
FrameParams frame_params(const rt_context *ctx, const float cam[12], uint32_t first, uint32_t count, uint32_t glog2,
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
    {
        volatile float c = (float)count;
        volatile float q = 1.0f / c;
        fp.inv_count = count ? q : 0.0f;
    }
    return fp;
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

// kernels come in (COUNT, ACCEL) instantiations; scenes without any BVH run the ACCEL = false ones
inline bool scene_has_accel(const DeviceScene &sc) { return sc.bvh_node_count != 0 || sc.mesh_bvh_root != nullptr; }
#define PT_DISPATCH(count_on, accel_on, CALL)                           \
    do {                                                                \
        if (count_on) { if (accel_on) CALL(true, true); else CALL(true, false); }   \
        else { if (accel_on) CALL(false, true); else CALL(false, false); }          \
    } while (0)

int ensure_slots(rt_context *ctx, size_t slots) {
    if (slots <= ctx->slot_capacity) return RT_OK;
    if (ctx->d_recs) (void)hipFree(ctx->d_recs);
    if (ctx->d_live) (void)hipFree(ctx->d_live);
    ctx->d_recs = nullptr;
    ctx->d_live = nullptr;
    ctx->slot_capacity = 0;
    // segmented live list: LIVE_SEGMENTS segments of whole workgroups' worth of entries, then the segment counters
    size_t entries = slots + (size_t)LIVE_SEGMENTS * 256u;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_recs, entries * sizeof(PixelRec)));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_live, (entries + (size_t)LIVE_SEGMENTS * LIVE_COUNT_STRIDE) * sizeof(uint32_t)));
    ctx->slot_capacity = slots;
    return RT_OK;
}

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
        // workgroup b of pt_prefix appends to segment b mod LIVE_SEGMENTS: a segment holds at most seg_cap entries
        const uint32_t prefix_blocks = (n + 255) / 256;
        fp.seg_cap = ((prefix_blocks + LIVE_SEGMENTS - 1) / LIVE_SEGMENTS) * 256u;
        // sample kernels deal their waves (pixel groups) over the segments: unit u → segment u mod LIVE_SEGMENTS
        auto units_for = [&](uint32_t per_unit) { return LIVE_SEGMENTS * ((fp.seg_cap + per_unit - 1) / per_unit); };
        dim3 block(256), grid1(prefix_blocks), grid2((unsigned)((((uint64_t)units_for(1u) << glog2) + 255) / 256));
        // sample queue: a wave owns ppw live pixels (<= QUEUE_SLOTS samples); worst case all n pixels are live
        uint32_t static_f4 = lds_static_used(sc.material_count, sc.sphere_count, sc.plane_count) +
                             (PT_LDS_SPHERES ? PT_LDS_SPHERE_CAP : 0);
        const bool sphere_bvh_only = sc.bvh_node_count != 0 && sc.mesh_bvh_root == nullptr;
        const bool simple_geom = sc.lens_count == 0 && sc.model_count == 0;   // spheres and planes only
        const uint32_t q_waves = !scene_has_accel(sc) ? PT_Q_WAVES : (sphere_bvh_only ? PT_Q_WAVES_SPHERE_BVH : PT_Q_WAVES_ACCEL);
        uint32_t ppw = queue_pixels_per_wave(count, q_waves, static_f4, PT_Q_BLOCK_WAVES);
        dim3 gridq((units_for(ppw) + PT_Q_BLOCK_WAVES - 1) / PT_Q_BLOCK_WAVES), blockq(64 * PT_Q_BLOCK_WAVES);
        bool queue = ctx->sample_queue && count <= QUEUE_SLOTS;
        size_t lds_q = static_f4 * sizeof(float4) + PT_Q_BLOCK_WAVES * (size_t)queue_wave_lds_bytes(ppw, count);
#define PT_CALL_PREFIX(C, A) \
    hipLaunchKernelGGL((pt_prefix<C, A>), grid1, block, 0, ctx->stream, sc, fp, ctx->d_recs, ctx->d_live, live_count, ctx->d_accum, ctx->d_counters)
#define PT_CALL_QUEUE_W(C, A, G, W) \
    hipLaunchKernelGGL((pt_samples_q<C, A, G, W>), gridq, blockq, lds_q, ctx->stream, sc, fp, ctx->d_recs, ctx->d_live, live_count, ctx->d_accum, ctx->d_counters, ppw)
#define PT_CALL_QUEUE(C, A)                                                                       \
    do {                                                                                          \
        if (!(A)) { if (simple_geom) PT_CALL_QUEUE_W(C, false, 0, PT_Q_WAVES); else PT_CALL_QUEUE_W(C, false, 1, PT_Q_WAVES); } \
        else if (sphere_bvh_only) { if (simple_geom) PT_CALL_QUEUE_W(C, true, 0, PT_Q_WAVES_SPHERE_BVH); else PT_CALL_QUEUE_W(C, true, 1, PT_Q_WAVES_SPHERE_BVH); } \
        else PT_CALL_QUEUE_W(C, true, 2, PT_Q_WAVES_ACCEL);                                       \
    } while (0)
#define PT_CALL_FIXED(C, A) \
    hipLaunchKernelGGL((pt_samples<C, A>), grid2, block, 0, ctx->stream, sc, fp, ctx->d_recs, ctx->d_live, live_count, ctx->d_accum, ctx->d_counters)
        bool accel_on = scene_has_accel(sc);
        PT_DISPATCH(ctx->count_enabled, accel_on, PT_CALL_PREFIX);
        HIP_TRY(ctx, hipEventRecord(evp[2], ctx->stream));  // (the last slot range's; one range is the normal case)
        if (queue && sc.mesh_bvh_root && ctx->walk_jobs.n && ctx->walk_slices && !ctx->count_enabled && !PT_LDS_SPHERES) {
            // every mesh has a BVH: interleaved walk slices (pt_samples_w), sized for its own occupancy target
            uint32_t ppw_w = queue_pixels_per_wave(count, ctx->walk_jobs.n == 1 ? PT_W_WAVES : PT_W_WAVES_MULTI, static_f4, PT_W_BLOCK_WAVES);
            size_t lds_w = static_f4 * sizeof(float4) + PT_W_BLOCK_WAVES * (size_t)queue_wave_lds_bytes(ppw_w, count);
            dim3 gridw((units_for(ppw_w) + PT_W_BLOCK_WAVES - 1) / PT_W_BLOCK_WAVES), blockw(64 * PT_W_BLOCK_WAVES);
            if (ctx->walk_jobs.n == 1)
                hipLaunchKernelGGL(pt_samples_w<false>, gridw, blockw, lds_w, ctx->stream, sc, fp, ctx->d_recs, ctx->d_live, live_count,
                                   ctx->d_accum, ppw_w, ctx->walk_jobs.p, 1u
#ifdef PT_WSTAT
                                   , ctx->d_counters + COUNTER_REPLICAS * COUNTER_STRIDE + 8
#endif
                                   );
            else
                hipLaunchKernelGGL(pt_samples_w<true>, gridw, blockw, lds_w, ctx->stream, sc, fp, ctx->d_recs, ctx->d_live, live_count,
                                   ctx->d_accum, ppw_w, ctx->walk_jobs.p, (uint32_t)ctx->walk_jobs.n
#ifdef PT_WSTAT
                                   , ctx->d_counters + COUNTER_REPLICAS * COUNTER_STRIDE + 8
#endif
                                   );
        } else if (queue) PT_DISPATCH(ctx->count_enabled, accel_on, PT_CALL_QUEUE);
        else PT_DISPATCH(ctx->count_enabled, accel_on, PT_CALL_FIXED);
#undef PT_CALL_PREFIX
#undef PT_CALL_QUEUE
#undef PT_CALL_QUEUE_W
#undef PT_CALL_FIXED
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(evp[1], ctx->stream));
    ctx->ev_count++;
    return RT_OK;
}

uint32_t group_log2_for(uint32_t count) {
    uint32_t g = 0;
    while ((1u << g) < count && g < 6) g++;
    return g;
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
    {
        // test layout of the spheres: (cx, cy, cz, r*r), whole batches, then one dummy batch that the
        // prefetch of the last iteration reads.  Dummies have r*r = -inf: their discriminant is -inf.
        uint32_t batches = (d->sphere_count + PT_SPHERE_BATCH - 1) / PT_SPHERE_BATCH;
        std::vector<float4> v((size_t)(batches + 1) * PT_SPHERE_BATCH, make_float4(0.0f, 0.0f, 0.0f, -INFINITY));
        for (uint32_t i = 0; i < d->sphere_count; i++) {
            volatile float r2 = d->spheres[i].r * d->spheres[i].r;  // one rounded binary32 product, as in hitSphere
            v[i] = make_float4(d->spheres[i].pos.x, d->spheres[i].pos.y, d->spheres[i].pos.z, r2);
        }
        HIP_TRY(ctx, ctx->sph4.upload(v.data(), v.size()));
        ctx->sphere_batches = batches;
    }
    {
        std::vector<uint32_t> base;
        std::vector<float4> fr;
        if (!build_face_records(d, base, fr)) return fail(ctx, RT_EINVAL, "too many faces");
        HIP_TRY(ctx, ctx->faces.upload(fr.data(), fr.size()));
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
        {   // device layout: 48 bytes per node (mesh_node_pack)
            std::vector<float4> packed(nodes.size() / 4 * 3);
            for (size_t n = 0; n < nodes.size() / 4; n++) mesh_node_pack(&nodes[4 * n], &packed[3 * n]);
            HIP_TRY(ctx, ctx->mbvh_nodes.upload(packed.data(), packed.size()));
        }
        HIP_TRY(ctx, ctx->mbvh_faces.upload(lfaces.data(), lfaces.size()));
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
            HIP_TRY(ctx, ctx->bvh_sph.upload(bb.leaf_sph.data(), bb.leaf_sph.size()));
            HIP_TRY(ctx, ctx->bvh_idx.upload(bb.leaf_idx.data(), bb.leaf_idx.size()));
            ctx->bvh_node_count = (uint32_t)(bb.nodes.size() / 4);
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
    if ((rc = launch_render<MODE_TRACE>(ctx, camera, 0, 1, 0)) != RT_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // queue.finish(), src/raytracer.cpp:140
    return RT_OK;
}

int rt_render_again(rt_context *ctx, const float camera[12]) {
    int rc = check_ready(ctx, camera);
    if (rc) return rc;
    if (ctx->sample_counter >= RT_MAX_SAMPLE) return fail(ctx, RT_EINVAL, "sample counter limit %u reached", RT_MAX_SAMPLE);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->sample_counter++;  // src/raytracer.cpp:147
    if ((rc = launch_render<MODE_RETRACE>(ctx, camera, ctx->sample_counter, 1, 0)) != RT_OK) return rc;
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
    rc = ctx->prefix_sharing ? launch_fused(ctx, camera, first_sample, n_samples, group_log2_for(n_samples))
                             : launch_render<MODE_ACCUM>(ctx, camera, first_sample, n_samples, group_log2_for(n_samples));
    if (rc != RT_OK) return rc;
    ctx->accum_count += n_samples;
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
        if (scene_has_accel(sc))
            hipLaunchKernelGGL(pt_probe<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, sc, fp, d_in,
                               d_in + n, d_in + 2 * n, (uint32_t)n, d_out);
        else
            hipLaunchKernelGGL(pt_probe<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, sc, fp, d_in,
                               d_in + n, d_in + 2 * n, (uint32_t)n, d_out);
        e = hipGetLastError();
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
        dim3 grid((unsigned)((n + 255) / 256)), block(256);
        if (scene_has_accel(sc))
            hipLaunchKernelGGL(pt_debug_hit<true>, grid, block, 0, ctx->stream, sc, kind, d_rays, d_prim, d_face, (uint32_t)n, (float *)d_out);
        else
            hipLaunchKernelGGL(pt_debug_hit<false>, grid, block, 0, ctx->stream, sc, kind, d_rays, d_prim, d_face, (uint32_t)n, (float *)d_out);
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
        hipLaunchKernelGGL(pt_debug_material, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, sc, routine,
                           (const float *)d_in, (uint32_t)n, (float *)d_out);
    });
}

int rt_debug_div3(rt_context *ctx, const float *in4, size_t n, float *out6) {
    if (!ctx || !in4 || !out6 || n > (1u << 28)) return RT_EINVAL;
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return debug_roundtrip(ctx, in4, 4 * n * sizeof(float), out6, 6 * n * sizeof(float), [&](void *d_in, void *d_out) {
        hipLaunchKernelGGL(pt_debug_div3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const float *)d_in,
                           (uint32_t)n, (float *)d_out);
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
        case RT_OPT_ACCEL:
            if (value < 0 || value > 2) return fail(ctx, RT_EINVAL, "RT_OPT_ACCEL takes 0, 1 or 2");
            ctx->accel = value;
            return RT_OK;
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
