/*
 * rt_amd.h — C ABI of the MI355X (gfx950) path tracer: librt_amd.so.
 *
 * This is the drop-in boundary for ONE hot path of antoni-wojcik/OpenCL-Raytracing:
 * the per-pixel Monte-Carlo trace loop (kernels `trace` / `retrace`,
 * kernels/raytracer.cl:496-532 and everything they reach, :93-494).  The
 * reference has no FFI layer; its boundary is the C++ class API
 * (include/raytracer.h:17-47, include/scene.h:83-153, include/camera.h:20-53)
 * plus the OpenCL kernel argument lists (src/raytracer.cpp:108-121,
 * src/scene.cpp:89-108).  Every entry point below names the reference
 * interface it replaces.  Plain pointers and sizes only: no C++ types, no
 * torch types.  All file:line citations are relative to the reference repo.
 *
 * Conventions
 *   - every function returns 0 on success, a negative RT_E* code on failure;
 *     rt_last_error() gives the message (reference: print + exit(-1),
 *     src/kernelgl.cpp:47-56, src/scene.cpp:29-32);
 *   - calls on one context must be serialised by the caller (reference: single
 *     GL thread, src/raytracer.cpp:134-140);
 *   - host arrays are copied at the call, the caller keeps ownership;
 *   - "device" pointers are HIP device addresses of the context's GPU;
 *   - the library never falls back to a CPU path: without a usable gfx950
 *     device rt_create() fails with RT_ENODEVICE.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 3

/* error codes */
#define RT_OK 0
#define RT_EINVAL (-1)    /* bad argument                                    */
#define RT_ENODEVICE (-2) /* no HIP device / wrong architecture              */
#define RT_EHIP (-3)      /* a HIP runtime call failed                       */
#define RT_ESTATE (-4)    /* call order violated (e.g. render before scene)  */
#define RT_ERANGE (-5)    /* index inside the scene points outside an array  */

/* kernel constants, kernels/raytracer.cl:1-7 (duplicated src/raytracer.cpp:21) */
#define RT_TRIANGLE_EPSILON 0.0000001f
#define RT_MIN_DISTANCE 0.001f
#define RT_MAX_DISTANCE 1000.0f
#define RT_DEPTH 30
#define RT_RANDOM_BUFFER_SIZE 100000
#define RT_RANDOM_TABLE_FLOATS (4 * RT_RANDOM_BUFFER_SIZE)

/* API limits: keep the 64-bit table index of raytracer.cl:115,122 below 2^32 */
#define RT_MAX_DIM 16384
#define RT_MAX_SAMPLE 65535u

/* ---- device data layouts (byte-identical to the reference's structs) ---- */

/* cl_float3 / OpenCL float3: 16 bytes, align 16 (include/scene.h:34,42,...) */
typedef struct rt_float3 { float x, y, z, w; } rt_float3;
typedef struct rt_float2 { float x, y; } rt_float2;

/* enum MatType, kernels/raytracer.cl:23, include/scene.h:30 */
enum rt_mat_type {
    RT_REFRACTIVE = 0,
    RT_REFLECTIVE = 1,
    RT_DIELECTRIC = 2,
    RT_DIFFUSE = 3,
    RT_TEXTURED = 4,
    RT_LIGHT = 5
};

/* Material, raytracer.cl:25-29 / scene.h:32-39 — 48 bytes */
typedef struct rt_material {
    int32_t type;
    int32_t _pad0[3];
    rt_float3 color;
    float extra_data;
    int32_t _pad1[3];
} rt_material;

/* Sphere, raytracer.cl:40-44 / scene.h:41-47 — 32 bytes */
typedef struct rt_sphere {
    rt_float3 pos;
    float r;
    uint32_t mat_ID;
    uint32_t _pad[2];
} rt_sphere;

/* Plane, raytracer.cl:46-50 / scene.h:49-55 — 48 bytes */
typedef struct rt_plane {
    rt_float3 pos;
    rt_float3 normal;
    uint32_t mat_ID;
    uint32_t _pad[3];
} rt_plane;

/* Lens, raytracer.cl:52-59 / scene.h:57-64 — 64 bytes */
typedef struct rt_lens {
    rt_float3 pos;
    rt_float3 p1;
    rt_float3 p2;
    float r1;
    float r2;
    uint32_t mat_ID;
    uint32_t _pad;
} rt_lens;

/* Mesh, raytracer.cl:61-66 / scene.h:66-73 — 16 bytes */
typedef struct rt_mesh {
    uint32_t vertex_anchor;
    uint32_t index_anchor;
    uint32_t face_count;
    uint32_t texture_ID;
} rt_mesh;

/* Model, raytracer.cl:68-72 / scene.h:75-81 — 12 bytes */
typedef struct rt_model {
    uint32_t mesh_anchor;
    uint32_t mesh_count;
    uint32_t mat_ID;
} rt_model;

/*
 * The nine arrays + counts the kernel reads: Scene, raytracer.cl:74-91, filled
 * by createScene (:541-558) from SceneCreator::setKernelArgs (src/scene.cpp:89-108).
 * The uv array is indexed with the vertex index (raytracer.cl:97-99): uv_count
 * must be 0 or equal to vertex_count (a short uv array is zero-filled on upload).
 */
typedef struct rt_scene_desc {
    const rt_material *materials;
    const rt_sphere *spheres;
    const rt_plane *planes;
    const rt_lens *lenses;
    const rt_float3 *vertices;
    const rt_float2 *uvs;
    const uint32_t *indices;
    const rt_mesh *meshes;
    const rt_model *models;
    uint32_t material_count;
    uint32_t sphere_count;
    uint32_t plane_count;
    uint32_t lens_count;
    uint32_t vertex_count;
    uint32_t uv_count;
    uint32_t index_count;
    uint32_t mesh_count;
    uint32_t model_count;
    uint32_t _pad;
} rt_scene_desc;

/*
 * Work counters of one render call (data dependent, exact, deterministic).
 * They price the reference kernel's logical global-memory traffic
 * (SURVEY §8d "algorithmic bytes"); see rt_counters_bytes().
 */
typedef struct rt_counters {
    uint64_t samples;       /* pixel-samples traced (camera block 48 B, image write 16 B) */
    uint64_t bounces;       /* hitScene calls, raytracer.cl:449                          */
    uint64_t t_sphere;      /* hitSphere calls  (32 B each)                              */
    uint64_t t_plane;       /* hitPlane calls   (48 B)                                   */
    uint64_t t_lens;        /* hitLens calls    (64 B)                                   */
    uint64_t t_model;       /* hitModel calls   (12 B)                                   */
    uint64_t t_mesh;        /* hitMeshOut calls (16 B)                                   */
    uint64_t t_tri;         /* hitTriangle calls (3 idx + 3 vtx = 60 B)                  */
    uint64_t h_tri;         /* triangle hits reaching the uv fetch :281 (36 B)           */
    uint64_t h_bounce;      /* bounces that hit something (material 48 B)                */
    uint64_t n_scatter;     /* randomVec table reads :113 (12 B)                         */
    uint64_t n_dielectric;  /* random() table reads :120 (4 B)                           */
    uint64_t n_texfetch;    /* bilinear texture fetches :105 (4 texels x 16 B)           */
    uint64_t image_reads;   /* retrace read of the previous pixel :524 (16 B)            */
} rt_counters;

typedef struct rt_context rt_context;

/* ---- lifetime ------------------------------------------------------------ */

/* ABI version of the loaded library (== RT_ABI_VERSION it was built with). */
int rt_abi_version(void);

/* Message of the last failure on ctx (or of the last failed rt_create when
 * ctx is NULL).  Replaces KernelGL::processError, src/kernelgl.cpp:47-56. */
const char *rt_last_error(const rt_context *ctx);

/* Replaces RayTracer::RayTracer(w,h,kernel_path) minus scene loading
 * (src/raytracer.cpp:24-36) and KernelGL::initialiseOpenCL
 * (src/kernelgl.cpp:58-93).  `device` is a HIP ordinal (the reference
 * hard-wires OpenCL GPU index 1, kernelgl.cpp:76).  Allocates the W×H RGBA32F
 * image (raytracer.cpp:54,60), the 12-float camera block (:64-65) and the
 * 400 000-float random table (:69-93, seeded as rt_set_seed(ctx, 0xC0FFEE)). */
int rt_create(int device, int width, int height, rt_context **out);

/* Replaces RayTracer::~RayTracer (src/raytracer.cpp:38-40). */
void rt_destroy(rt_context *ctx);

/* RayTracer::resize — declared, never defined (include/raytracer.h:46).
 * Reallocates the image; resets the sample counter. */
int rt_resize(rt_context *ctx, int width, int height);

/* Run on this hipStream_t (0 = the context's own stream).  The reference
 * creates a fresh cl::CommandQueue per call (src/raytracer.cpp:134). */
int rt_set_stream(rt_context *ctx, void *hip_stream);

/* ---- inputs -------------------------------------------------------------- */

/* Replaces SceneCreator::setupBuffers/createScene/setKernelArgs
 * (src/scene.cpp:46-108) and the createScene kernel (raytracer.cl:541-558).
 * Validates every index the kernel would dereference (RT_ERANGE). */
int rt_set_scene(rt_context *ctx, const rt_scene_desc *scene);

/* Replaces SceneCreator::loadTextures' upload (src/scene.cpp:164,173): `layers`
 * RGBA32F images of w×h texels, layer-major.  layers==0 installs the 1×1×1
 * dummy the reference creates for model-free scenes (scene.cpp:187-189). */
int rt_set_textures(rt_context *ctx, const float *rgba, int w, int h, int layers);

/* Replaces the random-table fill of RayTracer::createCLBuffers
 * (src/raytracer.cpp:69-93), which is unseeded (std::random_device).  The
 * table has the same layout and distribution (100 000 points uniform in the
 * unit ball, then 100 000 U[0,1)) but is a pure function of `seed`
 * (Philox-4x32-10; DESIGN.md "random table"). */
int rt_set_seed(rt_context *ctx, uint64_t seed);

/* Inject a caller-made table (n must be RT_RANDOM_TABLE_FLOATS). */
int rt_set_random_table(rt_context *ctx, const float *table, size_t n);

/* Copy the table currently on the device to `out` (n floats). */
int rt_get_random_table(rt_context *ctx, float *out, size_t n);

/* Host-only: fill `out[400000]` with the table rt_set_seed(seed) would
 * upload.  Needs no device and no context. */
int rt_make_random_table(uint64_t seed, float *out, size_t n);

/*
 * Frame sharding for multi-GPU rendering (new; the reference is single
 * device, src/kernelgl.cpp:76).  The frame is cut into tile_w×tile_h tiles,
 * numbered row-major; this context renders tiles t with t % world == rank
 * and leaves every other pixel of its buffers at zero, so an element-wise
 * sum over ranks (RCCL reduce) is the full frame.  Pixels keep their
 * whole-frame coordinates (they enter the table index, raytracer.cl:115,122),
 * so every pixel is bit-identical to the unsharded render.
 * Default: rank 0 of 1, 64×4 tiles.
 */
int rt_set_shard(rt_context *ctx, int rank, int world, int tile_w, int tile_h);

/* Compact exchange of a sharded accumulator (new).  rt_shard_slots: number of pixel
 * slots every rank of `world` packs (whole tiles, the same for all ranks).
 * rt_pack_accum: this rank's owned accumulator pixels → d_packed (device, slots × 16 B,
 * slot order).  rt_unpack_accum: scatter the packed pixels of rank `src_rank` of
 * `world` into this context's accumulator.  gather(packed) + unpack on rank 0 moves
 * 1/world of the frame per rank over xGMI instead of a full-frame reduce. */
int rt_shard_slots(rt_context *ctx, int world, uint32_t *slots_out);
int rt_pack_accum(rt_context *ctx, void *d_packed, size_t bytes);
int rt_unpack_accum(rt_context *ctx, const void *d_packed, size_t bytes, int src_rank, int world);

/* ---- rendering ----------------------------------------------------------- */

/* Replaces RayTracer::render (src/raytracer.cpp:127-144) + kernel `trace`
 * (raytracer.cl:496-510): sample_counter = 0; image = sqrt(getCol(sample 0)). */
int rt_render(rt_context *ctx, const float camera[12]);

/* Replaces RayTracer::renderAgain (src/raytracer.cpp:146-165) + kernel
 * `retrace` (raytracer.cl:512-532): ++sample_counter; running mean kept in
 * gamma space, bit-identical to the reference's arithmetic. */
int rt_render_again(rt_context *ctx, const float camera[12]);

/* Value of RayTracer::sample_counter (include/raytracer.h:21). */
int rt_sample_counter(const rt_context *ctx, uint32_t *out);

/*
 * Native fused path (new): samples first_sample .. first_sample+n-1 of every
 * owned pixel in ONE launch, each sample bit-identical to what `trace` /
 * `retrace` would have traced, summed per pixel in linear space in a fixed
 * order, added to the linear accumulator (RGB sum, and the sample count in the
 * 4th channel, so accumulators of several ranks can simply be summed).
 * rt_clear() zeroes it; rt_resolve() writes image = sqrt(sum / count), alpha 1.
 * 64 spp: rt_clear; rt_render_spp(cam, 0, 64); rt_resolve.
 * A call of more than 512 samples per pixel is executed as consecutive launches of 512 (the accumulator is the sum of
 * their sums; results within a launch are summed in a fixed order).
 */
int rt_clear(rt_context *ctx);
int rt_render_spp(rt_context *ctx, const float camera[12], uint32_t first_sample, uint32_t n_samples);
int rt_resolve(rt_context *ctx);

/* Wait for everything queued on the context's stream (reference:
 * queue.finish(), src/raytracer.cpp:140).  rt_render/rt_render_again already
 * return synchronously; rt_render_spp/rt_resolve/rt_clear are asynchronous. */
int rt_sync(rt_context *ctx);

/*
 * Parity probe (new): linear radiance getCol(...) (raytracer.cl:444-486) of
 * `n` individual pixel-samples (x[i], y[i], sample[i]) → out_rgb[3*i..].
 * Bit-exact against the reference per sample.  Host pointers.
 */
int rt_trace_samples(rt_context *ctx, const float camera[12], const uint32_t *x, const uint32_t *y,
                     const uint32_t *sample, size_t n, float *out_rgb);

/* ---- outputs ------------------------------------------------------------- */

/* Replaces RayTracer::transferImage (src/raytracer.cpp:167-174; there the
 * pixels stay in a GL texture): copy the gamma-space RGBA32F image, row 0 =
 * first row the kernel wrote (y = 0), to host memory.  bytes = w*h*16. */
int rt_read_image(rt_context *ctx, float *rgba, size_t bytes);

/* Linear accumulator divided by the sample count (RGBA32F, alpha 1). */
int rt_read_linear(rt_context *ctx, float *rgba, size_t bytes);

/* Device addresses of the W×H×4-float buffers, for zero-copy consumers
 * (RCCL reduce of the radiance buffer, display interop). */
int rt_device_image(rt_context *ctx, void **d_rgba);
int rt_device_accum(rt_context *ctx, void **d_rgba);

/* Tuning / diagnostics switches (new).  Results never depend on them. */
#define RT_OPT_PREFIX_SHARING 1         /* 1 (default): rt_render_spp traces the sample-invariant path
                                           prefix once per pixel; 0: every sample from the camera      */
#define RT_OPT_MAX_THREADS_PER_LAUNCH 2 /* split one render call into several kernel launches           */
#define RT_OPT_SAMPLE_QUEUE 3           /* 1 (default): lanes pull samples from an in-wave queue as their
                                           paths end; 0: one fixed sample set per lane                  */
#define RT_OPT_ACCEL 4                  /* sphere / mesh search: 0 brute force (the reference's loops), 1
                                           (default) conservative BVHs for >= 64 spheres and for meshes of
                                           >= 32 faces, 2 sphere BVH always.  Same winner in every mode
                                           (a set of >= 2^24 spheres, or meshes of >= 2^26 BVH nodes in all,
                                           take the brute-force loops: the walks address 32-bit offsets)      */
#define RT_OPT_WALK_SLICES 5            /* 1 (default): in scenes where every mesh of every model has a BVH, the
                                           lanes' mesh walks advance in interleaved slices; 0: every walk runs in place */
#define RT_OPT_PREFIX_TREE 7            /* a pixel whose first random event is a dielectric surface gets a shared DECISION TREE —
                                           glass has only two outcomes (refract / reflect, raytracer.cl:407-435), so both
                                           continuations are traced once per pixel and every sample only picks its branch with
                                           its own table entry.  1 (default): in calls of >= 24 samples per pixel (below that
                                           the trees cost more than they share); 2: in every call; 0: never — every sample
                                           traces its own way through the glass.  Same result bit for bit in every mode */
#define RT_OPT_WAVE_FILL 8              /* 1 (default): a sample-kernel wave owns fewer pixels when the launch is small (a small
                                           frame, a rank's share of a sharded one), so that the chip's wave slots are filled;
                                           0: always as many pixels per wave as its LDS share holds.  Same result bit for bit */
#define RT_OPT_ARITH 6                  /* the ARITHMETIC POLICY of the trace kernels (csrc/pt_arith.hpp).  The reference's
                                           random numbers are table entries indexed by a hash of the ray direction
                                           (raytracer.cl:113-125): one ulp re-routes a path, so "the reference's
                                           result" exists only relative to one definition of the OpenCL builtins, of
                                           `/`, sqrt and of the contraction of a*b+c.  Unlike the other options this
                                           one CHANGES THE RESULT — it selects which build of the reference is matched
                                           bit for bit per pixel-sample:                                          */
#define RT_ARITH_IEEE 0                 /*   (default) plain IEEE-754 builtins, correctly rounded / and sqrt, no fused
                                             multiply-add: the reference compiled -ffp-contract=off for a CPU with the
                                             builtin formulas of oracle/ref_shim.cpp — what the CPU oracle restates   */
#define RT_ARITH_ROCM_OCL_NOCONTRACT 1  /*   the reference built by ROCm's OpenCL tool chain for gfx950 with
                                             -ffp-contract=off: ROCm's builtin library (fma-chain dot / cross / mix,
                                             v_rsq_f32 normalize, ocml pow), 2.5-ulp `/`, 3-ulp sqrt               */
#define RT_ARITH_ROCM_OCL 2             /*   the same with OpenCL's DEFAULT flags: clang also contracts the kernel's
                                             own a*b+c expressions — what an unmodified KernelGL::buildProgram
                                             (src/kernelgl.cpp:95-106, no build options) gets from ROCm's OpenCL    */
int rt_set_option(rt_context *ctx, int option, int value);

/* ---- measurement --------------------------------------------------------- */

/* When enabled, render calls run the counting build of the kernel (slower)
 * and add to the context's counters.  Off by default. */
int rt_enable_counters(rt_context *ctx, int enable);
int rt_reset_counters(rt_context *ctx);
int rt_get_counters(rt_context *ctx, rt_counters *out);

/* Host-only self-check (no device, no context): builds the sphere BVH and the per-mesh BVHs
 * exactly as rt_set_scene does and verifies their invariants — every primitive in exactly one
 * leaf and inside its boxes, child boxes inside parent boxes, split axes and skip links (both
 * walks are threaded), the walk visiting every leaf exactly once in all eight direction octants,
 * smallest-face indices, normal cones, edge bounds.  stats = {sphere nodes, sphere leaves,
 * sphere depth, mesh nodes, mesh leaves, mesh depth, meshes with a BVH, 0}. */
int rt_debug_check_accel(const rt_scene_desc *scene, uint64_t stats[8], char *err, size_t err_len);

/* Diagnostics of the sphere search since the last reset (counting build only):
 * out[0] = BVH nodes entered, out[1] = sphere tests actually executed. */
int rt_get_debug_counters(rt_context *ctx, uint64_t out[2]);

/* Sticky flags "a BVH walk left its loop on its iteration bound instead of at the end of the tree" since the last
 * rt_reset_counters(): bit 0 sphere walk, bit 1 mesh walk, bit 2 the walk-slice kernel's outer loop.  Such a walk
 * may return a wrong nearest hit; the bounds are sized so that it cannot happen, and every GPU test and bench.py's
 * parity leg assert 0 here. */
int rt_walk_overflow(rt_context *ctx, uint32_t *flags_out);

/* Algorithmic bytes of the reference kernel for these counters
 * (SURVEY §8d): 32·t_sphere + 48·t_plane + 64·t_lens + 12·t_model +
 * 16·t_mesh + 60·t_tri + 36·h_tri + 48·h_bounce + 12·n_scatter +
 * 4·n_dielectric + 64·n_texfetch + (48+16)·samples + 16·image_reads. */
uint64_t rt_counters_bytes(const rt_counters *c);

/* Milliseconds the last render call's kernel(s) took on the device
 * (hipEvent pair recorded on the launch stream around the launch). */
int rt_last_kernel_ms(rt_context *ctx, float *ms);

/* The same for the last *n_out <= min(cap, 64) render calls, oldest first.  The
 * events are only read here, so a timed loop can run without per-step syncs. */
int rt_kernel_ms_history(rt_context *ctx, float *ms, size_t cap, size_t *n_out);

/* ---- unit probes of the device routines (test instrumentation; no reference counterpart) ----
 * rt_debug_hit: one intersection routine per record, through the same search + winner rebuild the
 * trace kernels inline.  kind 0 hitSphere (raytracer.cl:149), 1 hitPlane (:176), 2 hitLens (:196),
 * 3 hitScene (:322), 4 hitTriangle (:257) on face[i] of mesh prim[i].  rays: n × 6 floats (origin,
 * direction); out12: n × 12 floats {hit, t, p.xyz, normal.xyz, uv.xy, texture_ID bits, mat_ID bits},
 * all zero on a miss — the record layout of oracle/ref_shim.cpp ref_hit.
 * rt_debug_material: routine 0 rayReflect (:362), 1 rayRefract (:369), 2 rayScatter (:393),
 * 3 rayRefractDielectric (:407).  in16: n × 16 floats {ray dir.xyz, hit p.xyz, hit normal.xyz,
 * colour so far.xyz, mat_ID bits, s_seed bits, pixel x bits, pixel y bits}; out9: n × 9 floats
 * {new origin.xyz, new direction.xyz, colour.xyz} (getCol's mixCol is not part of the routines).
 * rt_debug_div3: in4 n × {a.xyz, d} → out6 n × {shared-reciprocal a/d, compiler's a/d}. */
int rt_debug_hit(rt_context *ctx, int kind, const float *rays, const uint32_t *prim, const uint32_t *face, size_t n,
                 float *out12);
int rt_debug_material(rt_context *ctx, int routine, const float *in16, size_t n, float *out9);
int rt_debug_div3(rt_context *ctx, const float *in4, size_t n, float *out6);
/* rt_debug_builtin: ONE builtin of the selected arithmetic policy per record — op 0 dot, 1 cross, 2 normalize,
 * 3 {a0/a1, 1/a0, (a.yz)/a6}, 4 sqrt, 5 mix(a, b, a6), 6 min, 7 sign, 8 pow(a0, 5), 9 the table hash of a.xyz (uint
 * bits).  in8: n × 8 floats {a.xyz, b.xyz, t, -}; out4: n × 4 floats.  tests/test_gpu_ref950.py compares policies
 * 1 / 2 with probe kernels that call ROCm's OpenCL builtins themselves. */
int rt_debug_builtin(rt_context *ctx, int op, const float *in8, size_t n, float *out4);

/* The two stages of the same calls separately: a fused rt_render_spp call is pt_prefix
 * (first_ms: one work-item per pixel, the sample-invariant path prefix) followed by the
 * per-sample kernel (second_ms: pt_samples_q / pt_samples_w, the dominant kernel that
 * bench.py prices against the roofline); a third event is recorded between them.  Calls
 * on the direct path (rt_render, rt_render_again) report first_ms = 0. */
int rt_stage_ms_history(rt_context *ctx, float *first_ms, float *second_ms, size_t cap, size_t *n_out);

/* Name, CU count and arch of the context's device, e.g. "gfx950". */
int rt_device_info(rt_context *ctx, char *name, size_t name_len, int *cu_count, char *arch, size_t arch_len);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
