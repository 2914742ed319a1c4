/* mirt.h -- C ABI of the MI355X (gfx950) render path.
 *
 * Drop-in boundary for the hot path of anthonyebiner/raytracer: everything `PathTracer::raytrace()` does after its
 * guards (reference include/raytracer/renderer.cuh:136-188: BVH build + flatten, uploads, the `raytrace_cuda` launch,
 * film read-back) and `PathTracer::save_to_file` (renderer.cuh:103-113).  Plain C: pointers and sizes only, no C++ or
 * torch types.  The library is libmirt.so (raytracer_amd/csrc); the C++ mirror of the reference's host API that calls
 * it is include/raytracer_amd/pathtracer.hpp; the binding a reference maintainer would add is in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success or a negative MIRT_E_* code; mirt_last_error() gives the message
 *     (the reference prints and returns / exit(1)s: renderer.cuh:118-133, util/misc.cuh:33-43);
 *   - host pointers are caller-owned and only touched during the call; device memory lives behind mirt_scene;
 *   - the library context (streams, events, pinned memory) belongs to the host thread that called mirt_init; one render in
 *     flight per host thread (the reference is single-threaded and blocking too, renderer.cuh:178).  A scene belongs to the
 *     DEVICE it was created on: using it from a thread whose context is on another device (or after mirt_init moved this
 *     thread to another device) fails with MIRT_E_STATE; another thread on the same device may use it (one call at a time).
 *   - the GPU is mandatory: there is no CPU fallback; without a usable HIP device every compute entry point fails with
 *     MIRT_E_DEVICE.
 */
#ifndef MIRT_H
#define MIRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRT_OK 0
#define MIRT_E_ARG (-1)       /* bad argument (null pointer, zero size, batch == 0, ...) */
#define MIRT_E_DEVICE (-2)    /* HIP error (message has the HIP error string) */
#define MIRT_E_STATE (-3)     /* scene / camera / lights not set, library not initialised */
#define MIRT_E_LIMIT (-4)     /* size beyond a documented limit */
#define MIRT_E_IO (-5)        /* file could not be written / read */

/* BSDF record = the reference's `BSDF` (scene/bsdf.cuh:31-67), 36 bytes: type tag + the union's floats in
 * declaration order.  type: 0 DIFFUSE {reflectance[3]}  1 EMISSION {radiance[3]}  2 REFLECTION {roughness, reflectance[3]}
 * 3 REFRACTION {ior, roughness, transmittance[3]}  4 GLASS {ior, roughness, reflectance[3], transmittance[3]}
 * 5 MICROFACET {alpha, eta[3], k[3]}. */
typedef struct mirt_bsdf {
  int32_t type;
  float p[8];
} mirt_bsdf;

/* Light record = the reference's `SceneLight` (scene/lights.cuh:8-35), 68 bytes, as its factories store it
 * (lights.cuh:123-146).  type: 0 AREA {radiance[3], position[3], direction[3] (unit), dim_x[3], dim_y[3], area}
 * 1 POINT {radiance[3], position[3]}  2 DIRECTIONAL {radiance[3], dir_to_light[3] (unit, = -travel direction)}. */
typedef struct mirt_light {
  int32_t type;
  float p[16];
} mirt_light;

/* Camera record = the reference's `Camera` (scene/camera.cuh:8-17), 72 bytes; c2w is column major (Matrix3f.cuh:165). */
typedef struct mirt_camera {
  float origin[3];
  float c2w[9];
  float hFov, vFov, nClip, fClip, aperture, focal_distance;
} mirt_camera;

/* = the reference's `Parameters` (raytracer.cuh:10-17), 24 bytes, same field order. */
typedef struct mirt_params {
  uint32_t samples_per_pixel;
  uint32_t samples_per_light;
  uint32_t samples_per_batch;   /* adaptive-stop test every this many samples (raytracer.cuh:115); must be > 0 */
  float max_tolerance;
  uint32_t max_ray_depth;
  uint32_t bvh_node_size;       /* informational here: the leaf size is fixed when the scene is created */
} mirt_params;

/* What the reference has no knob for.  Zero-initialise for reference behaviour on one GPU. */
typedef struct mirt_options {
  uint32_t rng_mode;        /* 0 = reference xorshift32 stream per pixel (util/sampler.cuh:8-15, renderer.cuh:50-51);
                               1 = counter-based Philox2x32-10 keyed by (pixel, sample, seed): needed for spp slices */
  uint32_t seed;            /* rng_mode 0: xor-ed into the pixel seed (0 = the reference); rng_mode 1: global seed */
  uint32_t traversal;       /* 0 = ordered traversal of the library's own 8-wide SAH tree (default, fastest); rays whose
                                   outcome could hinge on the reference's own box tests are re-traced by the reference walk.
                                   Equal to the reference on every fixture and on every soak of arbitrary rays (0 of 600 M),
                                   EXCEPT one known class ("what remains open"): a ray that runs almost in the
                                   plane of a triangle, for which the reference accepts an ill-conditioned Moeller-Trumbore
                                   "hit" lying in front of that triangle's own box and of the true nearest hit; an ordered
                                   traversal has pruned that box by distance.  Measured: 1 of 36 M rays aimed at vertices and
                                   edges; 1 - 6 per 10 000 rays CONSTRUCTED to run 1e-6 ... 1e-2 rad off a triangle's plane
                                   (profiles/r3_grazing_soak.txt).  Pinned: tests/test_gpu_configs.py
                                   ::test_known_counter_example_of_traversal_0, ::test_grazing_rays_exact_modes_and_rate_...
                               1 = the reference's tree folded to 4-wide nodes, visited in the reference's DFS order:
                                   THE PROVABLY EXACT MODE (bit-exact with bvh.cuh:78-107 by construction); its frame cost
                                   next to mode 0 is tabulated in DESIGN.md 5
                               2 = the reference's BVH2 walked as the reference does (cross-check) */
  uint32_t tile_rank;       /* tile partition: this process renders 8x8 tiles t with t % tile_world == tile_rank */
  uint32_t tile_world;      /* 0 or 1 = whole image */
  uint32_t sample_begin;    /* rng_mode 1 only: first sample index of this process's spp slice */
  uint32_t sample_count;    /* rng_mode 1 only: samples in the slice; 0 = samples_per_pixel */
  uint32_t glass_int_abs;   /* 1 = reproduce the g++ build's integer abs() in GLASS (bsdf.cuh:213) */
  uint32_t flags;           /* bit 0 = per-launch event timing into mirt_stats (bit 3: of the closest-hit launches only),
                               bit 1 = count node visits / triangle tests,
                               bit 2 = progressive pass (mirt_render_accum, rng_mode 1): add this call's samples
                               [sample_begin, sample_begin + sample_count) to what accum_dev already holds instead of
                               clearing it; mirt_tonemap of the buffer after any pass is the image so far,
                               bit 4 = do not trace shadow rays whose contribution is exactly zero (a light behind the
                               surface; the reference traces them, raytracer.cuh:38-47, and adds 0): same film, fewer rays
                               in mirt_stats.shadow_rays (at most 32 shadow rays per hit, else ignored) */
  uint32_t streams;         /* rng_mode 1 only, 0 or 1 = off: S sample streams per pixel in flight at once (stream k renders
                               the k-th contiguous share of the pixel's samples into its own partial sum; the partial sums are
                               added in stream order at the end).  Same samples, S times fewer and S times larger wavefront
                               iterations; like an spp slice it switches the adaptive stop off.  At most 16. */
  uint32_t shares;          /* rng_mode 1 only, 0 = off: the share pool.  A pixel's samples are cut into W = shares contiguous shares,
                               each summed in sample order into a partial sum of its own, the partial sums added in share
                               order - exactly the image of streams = W - but only `streams` paths per pixel are in flight:
                               a path slot that has finished a share takes the next unrendered (share, pixel) of the frame,
                               so every slot stays busy until the frame runs out of work instead of idling while the
                               pixels with long paths finish (no reference counterpart).  Needs
                               shares > streams (else it is ignored), shares <= samples and samples % shares == 0; memory:
                               one float4 plane of the image per share. */
  uint32_t pool_group;      /* share pool: a slot takes its next (share, pixel) from the pool of its home group, the
                               pool_group neighbouring pixel slots (whole 8x8 tiles: a multiple of 64) around its own pixel,
                               so that it keeps rendering its neighbourhood; 0 = the library's default (64, MIRT_POOL_GROUP).
                               Any value gives the same image; it trades balance (large) against locality (small). */
  uint32_t reserved[4];
} mirt_options;

typedef struct mirt_stats {
  uint64_t closest_rays;    /* one per iteration of the bounce loop (raytracer.cuh:60-62) */
  uint64_t shadow_rays;     /* one per light sample per non-delta hit (raytracer.cuh:38-43) */
  uint64_t paths;           /* camera samples started */
  uint64_t exact_fallback_rays;   /* rays re-traced by the reference-order kernel (non-finite slab / hit values) */
  uint32_t iterations;      /* wavefront iterations */
  uint32_t tonemap_table_exact;
  double render_ms;         /* first launch -> accumulation buffer complete, device events */
  double trace_closest_ms, trace_shadow_ms, logic_ms;   /* per-kernel sums, device events */
  double total_ms;          /* wall: entry -> outputs in host memory */
  /* filled when opts->reserved[0] has bit 0 (time every launch with stream events) / bit 1 (count traversal work) */
  uint32_t closest_launches, shadow_launches;         /* launches of the wide-BVH traversal kernel that were timed */
  uint64_t closest_node_visits, closest_tri_tests;    /* wide nodes fetched / triangles tested by closest-hit rays */
  uint64_t shadow_node_visits, shadow_tri_tests;
  /* wave-level loop steps behind those counts: visits / (64 * steps) = SIMD lane utilisation of that loop */
  uint64_t closest_node_steps, closest_tri_steps, shadow_node_steps, shadow_tri_steps;
  /* the traversal that ran: opts->traversal, except that 0 becomes 1 for a scene whose own tree could not be built
   * (non-finite vertex coordinates) */
  uint32_t traversal_used;
  uint32_t logic_launches;  /* launches of k_logic behind logic_ms (flags bit 0) */
  /* k_logic's work over the frame, from the queue counters the host reads anyway: slots that ran the slot function (still
   * rendering at the launch) and slots a launch looked at (all of a pipe's slots, or its live-slot list in the frame's tail).
   * What bench.py prices k_logic's algorithmic bytes with. */
  uint64_t logic_live_slots, logic_scanned_slots;
} mirt_stats;

typedef struct mirt_scene mirt_scene;

/* Select the HIP device for this process (one process per GPU) and create streams/tables.  Idempotent. */
int mirt_init(int device);
void mirt_shutdown(void);
const char *mirt_last_error(void);
const char *mirt_version(void);

/* Replaces renderer.cuh:136-146 (BVHAccel ctor bvh.cuh:177-225, to_optimized_bvh bvh.cuh:231-290, to_cuda bvh.cuh:112-145).
 * tri_pos / tri_nrm: n_tris x 9 floats (p1 p2 p3 / n1 n2 n3, scene/primitives.cuh:23-26); tri_mat: material index per
 * triangle.  The reference's BVH2 is rebuilt bit for bit on the host (its flat primitive order defines hit indices), then
 * folded into the wide nodes of traversal 1; the ordered traversal's own tree (binned SAH, 8-wide quantised nodes) is built
 * next to it over the same primitives. */
int mirt_scene_create(const float *tri_pos, const float *tri_nrm, const uint32_t *tri_mat, uint32_t n_tris,
                      const mirt_bsdf *mats, uint32_t n_mats, const mirt_light *lights, uint32_t n_lights,
                      uint32_t leaf_size, mirt_scene **out);

/* The same with sphere primitives (scene/primitives.cuh:18-22, PrimitiveFactory::createSphere :184-190): prim_type holds
 * the reference's enum per primitive (1 SPHERE, 2 TRIANGLE; NULL = all triangles).  A sphere's 9-float record in tri_pos is
 * (origin.x, origin.y, origin.z, r, 5 unused); its tri_nrm record is ignored.  Replaces the SPHERE arms of
 * Primitive::intersect / has_intersection / get_bbox (primitives.cuh:64-85, :111-136, :166-169). */
int mirt_scene_create_prims(const float *tri_pos, const float *tri_nrm, const uint32_t *tri_mat, const uint8_t *prim_type,
                            uint32_t n_prims, const mirt_bsdf *mats, uint32_t n_mats, const mirt_light *lights,
                            uint32_t n_lights, uint32_t leaf_size, mirt_scene **out);

/* Scene build options.  No counterpart in the reference, whose builder runs on the host only (bvh.cuh:177-225): the
 * reference's BVH2 (flat primitive order, the tree of traversals 1 and 2 and of the near-tie replay) is always rebuilt on the
 * host bit for bit; the option selects who builds the ordered traversal's OWN tree, which is free in its topology. */
typedef struct mirt_build_options {
  uint32_t own_tree_builder;    /* 0: host, binned SAH (default)
                                 * 1: device, Morton order + binary radix tree (fastest build)
                                 * 2: device, PLOC: nearest-neighbour clustering in Morton order (traversal within a few
                                 *    per cent of the host tree's, built in milliseconds; falls back to 1 if clustering
                                 *    stalls on degenerate input)
                                 * 1 and 2 (raytracer_amd/csrc/build.hip) are for scenes that are rebuilt every frame,
                                 * the reference app's 80-frame loop (apps/app.cu:24-30) */
  uint32_t reserved[7];
} mirt_build_options;

/* mirt_scene_create_prims with build options (NULL: defaults, or MIRT_BUILDER=device|ploc from the environment). */
int mirt_scene_create_opts(const float *tri_pos, const float *tri_nrm, const uint32_t *tri_mat, const uint8_t *prim_type,
                           uint32_t n_prims, const mirt_bsdf *mats, uint32_t n_mats, const mirt_light *lights,
                           uint32_t n_lights, uint32_t leaf_size, const mirt_build_options *build, mirt_scene **out);
void mirt_scene_destroy(mirt_scene *scene);

/* ms[8]: reference BVH2 on the host (replaces the timing print of bvh.cuh:177-185), its fold to 4-wide, the own tree in
 * total, then the device builder's stages (upload + keys + sort | radix tree + boxes | fold | gather; 0 for the host
 * builder), uploads. */
int mirt_scene_build_times(const mirt_scene *scene, float ms[8]);

/* The ordered traversal's own tree as it sits on the device, for structure tests.  counts[4] = 8-wide nodes (0: the scene
 * has no such tree), levels below the root, primitives, builder used; nodes8q_out: 24 dwords per node; prims_out: 12 dwords
 * per primitive in tree order (word 9 = flat index).  Either may be NULL.  No reference counterpart. */
int mirt_scene_own_tree(const mirt_scene *scene, uint32_t counts[4], uint32_t *nodes8q_out, uint32_t *prims_out);

/* info[0..7]: bvh2 nodes, bvh2 depth (BVHAccel::get_max_depth, bvh.cuh:297), wide nodes, wide depth, n_tris, n_mats,
 * n_lights, device bytes / 1024. */
int mirt_scene_info(const mirt_scene *scene, uint32_t info[8]);

/* The reference-layout BVH2 (BVHNodeOpt, bvh.cuh:12-30: 8 dwords per node) and, per flat primitive slot, the index the
 * primitive had in the caller's arrays.  For structure parity tests. */
int mirt_scene_bvh2(const mirt_scene *scene, uint32_t *nodes_out, uint32_t *flat_to_orig);

/* Builds the camera record exactly as the reference's Camera ctor does (scene/camera.cuh:19-36) incl. the
 * focal_distance == 0 rule of PathTracer::set_camera (renderer.cuh:95-101). */
int mirt_camera_look_at(const float from[3], const float at[3], const float up[3], float hFov, float vFov, float nClip,
                        float fClip, float aperture, float focal_distance, mirt_camera *out);

/* Replaces renderer.cuh:148-188 (uploads, raytrace_cuda, from_cuda).  rgb_out: w*h*3 int32 in host memory, row major,
 * y down, exactly `SampleBuffer::data` (util/image.cuh:9-48).  opts and stats may be NULL. */
int mirt_render(mirt_scene *scene, const mirt_camera *camera, const mirt_params *params, const mirt_options *opts,
                uint32_t width, uint32_t height, int32_t *rgb_out, mirt_stats *stats);

/* Same render, but leaves the float accumulation buffer on the device: accum_dev = width*height float4
 * (sum r, sum g, sum b, divisor) in DEVICE memory owned by the caller (e.g. a torch tensor), pixels this process does not
 * own are written as zeros, so a plain sum over processes (RCCL reduce) composes tile and spp partitions.
 * Stream contract (also mirt_tonemap): the library runs on its own non-blocking HIP streams.  On entry it orders itself
 * after everything queued so far on the device's NULL stream (torch's default stream), and it returns only when accum_dev is
 * complete, so NULL-stream users need no synchronisation of their own.  Work the caller has in flight on OTHER streams
 * that touches accum_dev must be synchronised by the caller before the call. */
int mirt_render_accum(mirt_scene *scene, const mirt_camera *camera, const mirt_params *params,
                      const mirt_options *opts, uint32_t width, uint32_t height, void *accum_dev, mirt_stats *stats);

/* Film: `color /= divisor; SampleBuffer::update_pixel` (raytracer.cuh:124-125, util/image.cuh:36-48) over a device
 * accumulation buffer; rgb_out is host memory (w*h*3 int32). */
int mirt_tonemap(const void *accum_dev, uint32_t width, uint32_t height, int32_t *rgb_out);

/* Parity probe: sample-0 camera ray of every pixel, generated with the reference seed procedure (renderer.cuh:50-51,
 * raytracer.cuh:101-104) and traced with the selected traversal.  idx_out: flat primitive index (reference order) or -1;
 * tbits_out: bit pattern of ray.max_t after the traversal.  Host buffers, w*h each. */
int mirt_primary_hits(mirt_scene *scene, const mirt_camera *camera, const mirt_options *opts, uint32_t width,
                      uint32_t height, int32_t *idx_out, uint32_t *tbits_out);

/* Parity probe: trace caller-supplied rays.  rays: n x 8 floats (origin[3], direction as given to the reference's Ray
 * ctor [3], min_t, max_t) (scene/ray.cuh:19-25).  any_hit 0: BVHAccelOpt::intersect (bvh.cuh:78-107) -> idx_out flat
 * index or -1, tbits_out = bits of max_t, bary_out (n x 2, may be NULL) = the two barycentrics b.y b.z.
 * any_hit 1: BVHAccelOpt::has_intersection (bvh.cuh:49-76) -> idx_out 1 / -1. */
int mirt_trace(mirt_scene *scene, const float *rays, uint32_t n, int any_hit, uint32_t traversal, int32_t *idx_out,
               uint32_t *tbits_out, float *bary_out);

/* mirt_trace without any allocation.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot call hipMalloc --
 * code under HIP graph capture, tight serving loops, harnesses that forbid it.  rays_dev (n x 8 floats, as mirt_trace),
 * idx_dev and tbits_dev (n each) are DEVICE memory of the caller; results as mirt_trace (no barycentrics).  rays_dev must be
 * 16-byte aligned (the records are read as float4), idx_dev / tbits_dev 4-byte aligned: MIRT_E_ARG otherwise.  Every
 * intermediate (ray and hit records, queues, counters, the traversal's stack overflow region) lives in one static device
 * array of ~10 MB that each GPU the library runs on holds until the process ends; its address is looked up once per
 * device in mirt_init.  Limit, rejected with MIRT_E_LIMIT: n <= MIRT_TRACE_STATIC_MAX_RAYS (a ray whose traversal stack
 * outgrows the array's 24 entries per lane beyond the LDS part is finished by the reference walk, as in mirt_trace: any tree
 * works).  The call makes no hipMalloc / hipFree / hipHostMalloc and creates no stream or event; it is ordered after
 * the device's NULL stream like mirt_render_accum and returns when idx_dev / tbits_dev are complete.  All calls on one GPU
 * share that GPU's array: calls that overlap on one GPU (two host threads) race and must be serialised by the caller. */
#define MIRT_TRACE_STATIC_MAX_RAYS 65536u
int mirt_trace_static(mirt_scene *scene, const float *rays_dev, uint32_t n, int any_hit, uint32_t traversal, int32_t *idx_dev,
                      uint32_t *tbits_dev);

/* ---- several GPUs from one host process.  NO COUNTERPART IN THE REFERENCE, which is single-device; this is how
 * a caller of the kept host API (PathTracer::raytrace, renderer.cuh:117-189) splits a frame over the GPUs of a node without
 * Python.  A group owns one persistent host thread per member, each with its own library context (the context of
 * mirt_init is per host thread) and its own replica of the scene.  mirt_group_render: every member renders its partition
 * of the frame - MIRT_PARTITION_SPP: member r takes samples [r * S / n, (r + 1) * S / n) of every pixel (counter RNG is
 * switched on; needs S >= n); MIRT_PARTITION_TILE: interleaved 8x8 tiles, bit-identical to one GPU - then ONE sum of the
 * float4 accumulation buffers onto member 0 (RCCL ncclReduce over xGMI when every member has a GPU of its own - librccl is
 * loaded at run time, only then; members sharing a GPU are added by a kernel, in member order) and the film as mirt_render
 * returns it (opts->shares, if set, is reduced per member to a divisor of the samples that member renders).  devices[]: HIP device index per member; an index may repeat (logical partitions of one GPU).  stats: ray
 * counts summed over the members, times = the slowest member's.  opts: as mirt_render, tile_* / sample_* must be zero. */
typedef struct mirt_group mirt_group;
#define MIRT_PARTITION_SPP 0
#define MIRT_PARTITION_TILE 1
int mirt_group_create(const int *devices, int n, mirt_group **out);
void mirt_group_destroy(mirt_group *group);
int mirt_group_size(const mirt_group *group);
/* the scene of mirt_scene_create_opts, replicated on every member (replaces the group's previous scene) */
int mirt_group_set_scene(mirt_group *group, const float *tri_pos, const float *tri_nrm, const uint32_t *tri_mat,
                         const uint8_t *prim_type, uint32_t n_prims, const mirt_bsdf *mats, uint32_t n_mats,
                         const mirt_light *lights, uint32_t n_lights, uint32_t leaf_size, const mirt_build_options *build);
int mirt_group_render(mirt_group *group, const mirt_camera *camera, const mirt_params *params, const mirt_options *opts,
                      int partition, uint32_t width, uint32_t height, int32_t *rgb_out, mirt_stats *stats);

/* Replaces PathTracer::save_to_file (renderer.cuh:103-113) + bitmap_image::save_image (util/bitmap_image.cuh:385-430):
 * 24-bit BMP, bottom-up rows padded to 4 bytes. */
int mirt_save_bmp(const char *path, const int32_t *rgb, uint32_t width, uint32_t height);

/* Host-side film function for single values (same table the device kernel uses); for tests. */
int mirt_tonemap_host(const float *rgb_in, uint32_t n_pixels, int32_t *rgb_out);

/* Host-only structure probe (needs no GPU): runs the same builder mirt_scene_create uses.  counts[4] = BVH2 nodes, BVH2
 * depth, wide nodes, wide depth.  Any of nodes2_out (8 dwords per node, BVHNodeOpt bvh.cuh:12-30), flat_to_orig (n_tris)
 * and nodes4_out (32 dwords per 128-byte wide node) may be NULL; call once with NULLs to size them. */
int mirt_bvh_build_host(const float *tri_pos, uint32_t n_tris, uint32_t leaf_size, uint32_t counts[4],
                        uint32_t *nodes2_out, uint32_t *flat_to_orig, uint32_t *nodes4_out);

/* The same for scenes with spheres (prim_type as in mirt_scene_create_prims), plus the ordered traversal's own tree:
 * counts[8] = BVH2 nodes, BVH2 depth, wide nodes, wide depth, quantised 8-wide nodes, their depth, 0, 0; nodes8q_out: 24 dwords per 96-byte node (raytracer_amd/csrc/host_bvh.hpp Node8Q); fast_to_flat_out: that tree's
 * primitive order -> flat index of the reference order.  Parity probe; no reference counterpart. */
int mirt_bvh_build_host_prims(const float *prim_pos, const uint8_t *prim_type, uint32_t n_prims, uint32_t leaf_size,
                              uint32_t counts[8], uint32_t *nodes2_out, uint32_t *flat_to_orig, uint32_t *nodes4_out,
                              uint32_t *nodes8q_out, uint32_t *fast_to_flat_out);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_H */
