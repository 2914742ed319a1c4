// Argument blocks and launch wrappers shared by kernels.hip (device) and capi.cpp (host orchestration).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fastdiv.hpp"
#include "host_bvh.hpp"
#include "shade.hpp"

#ifndef MIRT_REC12
#define MIRT_REC12 1          // path masks and shadow contributions as 12-byte records (0: 16-byte records with a pad word)
#endif
#ifndef MIRT_SHARED_ORIGIN
#define MIRT_SHARED_ORIGIN 1  // the shadow rays of a hit share one origin record per slot (0: one origin per shadow ray)
#endif

namespace mirt {

#if MIRT_REC12
typedef float3 rec3;
__host__ __device__ inline rec3 mk_rec3(float x, float y, float z) { return make_float3(x, y, z); }
#else
typedef float4 rec3;
__host__ __device__ inline rec3 mk_rec3(float x, float y, float z) { return make_float4(x, y, z, 0.f); }
#endif

constexpr int TRACE_BLOCK = 256;
#ifndef MIRT_LDS_STACK
#define MIRT_LDS_STACK 10
#endif
constexpr int LDS_STACK = MIRT_LDS_STACK;        // traversal-stack entries per lane kept in LDS (8 B each): 20 KiB per workgroup: 7 workgroups per CU (12 entries, 6 workgroups: 1.3 % slower)
constexpr int BVH2_STACK = 64;       // k_trace2 (the reference uses 32 unchecked, util/misc.cuh:19)

// per-slot flag bits (RenderArgs::st[].z low byte)
enum : uint32_t { F_DONE = 1, F_HAS_RAY = 2, F_PENDING = 4, F_AHEAD = 8, F_NEWITEM = 16 };
// Queue counters (word offsets into RenderArgs::qcount).  Every counter that kernels update with atomics lives on a
// 128-byte line of its own: same-address atomics retire at ~11 ns each on MI355X (tools/micro/atomic_rate.hip) and
// counters sharing a line queue up behind each other.
constexpr int CTR_STRIDE = 32;
enum : int { QC_EXT_FRONT = 0, QC_EXT_BACK = CTR_STRIDE, QC_SHADOW = 2 * CTR_STRIDE, QC_LIVE = 3 * CTR_STRIDE,
             QC_CULLED = 4 * CTR_STRIDE, QC_WORDS = 5 * CTR_STRIDE };
constexpr uint32_t QUEUE_SKIP = 0xFFFFFFFFu;   // a queue entry no ray stands behind (a culled shadow ray)
constexpr uint32_t NO_CONST_K = 0xFFFFFFFFu;
constexpr int WORK_PARTS = 8;        // a trace launch splits its queue into this many ranges, one fetch counter each

struct RenderArgs {
  uint32_t n_slots, width, height, tiles_x, tiles_y, tile_rank, tile_world;
  uint32_t rng_mode, seed, sample_begin;
  uint32_t stream_first;                  // the stream of this launch's slot 0 (a pipe renders a contiguous range of streams)
  uint32_t n_pixel_slots, streams;        // streams > 1: slot = stream * n_pixel_slots + pixel slot (mirt_options.streams)
  uint32_t spp, spl, batch, max_depth, adaptive, glass_int_abs, n_lights;
  float tol;
  DCamera cam;
  // per-slot path state (SoA, 16-byte records)
  uint4 *st;            // rng state | sample index | flags + shadow count << 8 | bounce index
  rec3 *mask;           // color_mask (12-byte records: k_logic is bound by the bytes it moves, most of them writes)
  float4 *total;        // total_color of the sample in flight, .w = s1 (raytracer.cuh:98)
  float4 *accum;        // sum of finished samples
  rec3 *dmask;          // color_mask at the hit whose shadow rays are in flight
  float4 *ray_o, *ray_d, *hit;            // extension ray (origin|min_t, ctor direction|max_t) and its hit (t, b.y, b.z, idx)
  float4 *sh_o, *sh_d;                    // shadow rays: ONE origin per slot (the shadow rays of a hit all start at hit + bump,
  rec3 *sh_c;                             //   raytracer.cuh:36-41), direction|max_t and contribution per entry e = k * n_slots + slot
  uint32_t *sh_occ;
  uint32_t *ext_queue, *sh_queue, *qcount;
  unsigned long long *ray_stats;          // closest, shadow, paths
  float4 *out_accum;                      // width * height, (sum rgb, divisor); streams > 1: `streams` such planes
  const float4 *tri_nrm;                  // 3 float4 per triangle: n1 n2 n3, material, original index
  const float4 *mats;                     // 3 float4 per material: mirt_bsdf (9 dwords) padded to 48 B
  uint32_t *mf_queue, *mf_count;          // slots whose hit material is MICROFACET: shaded by k_logic<true>
  // Slots that are still rendering after this launch are appended to live_out (count: qcount[QC_LIVE]); the next launch
  // walks the previous list instead of all slots once it covers less than half of them (the tail of a frame, where a
  // wave would otherwise run the whole slot function for one or two lanes).  live_in == NULL: first iteration.
  const uint32_t *live_in, *live_in_count;
  uint32_t *live_out;
  uint32_t live_16th;                     // the list is used when it holds at most live_16th / 16 of the slots
  uint32_t cull_zero;                     // mirt_options.flags bit 4: shadow rays whose contribution is exactly 0 are not traced
  const float *lights;                    // 17 dwords per light (mirt_light)
  // Housekeeping folded into k_logic so that an iteration costs no extra dispatches (host loop):
  uint32_t *clear_next;                   // k_logic<false>: workgroup 0 zeroes the control block of the NEXT iteration (three
  uint32_t clear_words;                   //   blocks rotate, so that block is idle now), clear_words words of it
  uint32_t first;                         // k_logic<false>: the frame's first iteration - slots are initialised in place of k_init
  // Shadow rays towards a DIRECTIONAL light all have the light's direction and no far limit.  For ONE entry of a hit - the
  // first that goes to such a light, const_k (NO_CONST_K: none) - the host fills the sh_d records once (capi.hip,
  // fill_constant_directions) and k_logic never writes them again: 16 bytes less to write per hit, and the traversal reads
  // its ray records as before.  (Handing the direction to the traversal as an argument instead was tried first: a table
  // in memory, four entries or one in the argument block - each made the occlusion kernel 25 - 30 % slower or cost it
  // 40 VGPRs: any second source for the record changes the code around its refill, which runs every other trip.)
  uint32_t const_k;
  // divisions k_logic does per slot and iteration, by multiplication (fastdiv.hpp; every dividend is below 2^31)
  FastDiv fd_pixel_slots, fd_tiles_x, fd_pool_group;
  uint32_t share_len;                     // spp / shares
  // Share pool (mirt_options.shares = W > streams, counter RNG): a pixel's samples are cut into W contiguous shares, each summed
  // in sample order into a plane of its own (out_accum holds W planes, added in share order at the end), but only `streams`
  // slots per pixel slot exist.  A work item is (share, pixel slot) = share * n_pixel_slots + pixel slot; slot s starts with
  // item stream_first * n_pixel_slots + s - the static assignment - and a slot that has finished its item takes the next one
  // from the pool of its home group (kernels.hip, pool_item): the slots of a neighbourhood stay busy until the neighbourhood
  // runs out of items, instead of the slots with short paths idling while the ones with long paths finish their fixed share
  // (DESIGN.md 5, round 3).
  // total[].w (s1, only read by the adaptive stop, which is off here) carries the item of the path in flight, accum[].w the
  // item the accumulator belongs to - they differ while a finished share waits for its last shadow rays (F_NEWITEM).
  uint32_t shares;                        // planes per pixel: W, or `streams` when the pool is off (1: a single accumulator)
  uint32_t pool;                          // 1: items beyond the static ones come from pool_next
  uint32_t pool_group;                    // pixel slots per home group (a multiple of 64): a slot takes items of its own group only
  uint32_t *pool_next;                    // one counter per group, shared by the pipes: items (streams + j / size, j % size) handed out
};

struct TraceArgs {
  const float4 *nodes4;                   // traversal 1: 8 float4 per 4-wide node of the folded reference tree
  const void *nodes8;                     // traversal 0: 96-byte quantised 8-wide nodes of the SAH tree (k_trace_fast)
  uint32_t refill_min;                    // k_trace_fast: idle lanes a wave collects before it fetches new rays
  uint32_t static_16th;                   // k_trace_fast: sixteenths of the queue handed out as per-workgroup slices
  uint32_t spheres;                       // the scene has sphere primitives (selects the kernel instantiation)
  float sph_rmin;                         // smallest sphere radius of the scene (0: unknown): bounds the box dilation of sphere scenes
  const float4 *nodes2;                   // 2 float4 per reference node
  const float4 *tris;                     // triangles in the order of the tree k_trace4 walks: p1, e1, e2, flat index
  const float4 *tris2;                    // triangles in the reference's flat order (k_trace2)
  const float *verts;                     // 9 floats per primitive, flat order: the exact p1 p2 p3 (own-box test of k_trace_fast)
  const uint2 *big;                       // (start, count) of leaves with more than 15 primitives
  const float4 *ray_o, *ray_d;
  uint32_t o_slots;                       // != 0: ray id's origin is ray_o[id % o_slots] (the shadow rays of a render: one origin per slot)
  const uint32_t *queue, *qcount;
  int qc_front, qc_back;                  // indices into qcount; qc_back < 0: no back range
  uint32_t qcap;                          // the back range ends at queue[qcap - 1]
  float4 *hit;                            // closest-hit output
  uint32_t *occ;                          // any-hit output
  float *any_t;                           // any-hit: max_t after the walk (probe only, may be null)
  uint32_t *work_head;                    // WORK_PARTS counters, CTR_STRIDE words apart
  uint2 *ovf;
  uint32_t ovf_depth;
  uint32_t *redo_queue, *redo_count;
  uint32_t redo;                          // 1: k_trace2 walks redo_queue[0 .. *redo_count)
  uint32_t *error_flag;
  unsigned long long *redo_stat;          // counts rays re-traced by the reference walk (may be null)
  unsigned long long *count_stat;         // [0] node visits, [1] triangle tests (counting launches only)
  uint32_t count;
  uint32_t require_unit;                  // traversal 0: rays whose ctor direction is not unit length go to k_trace2
  uint32_t *host_slot;                    // non-null (closest-hit launch of a render iteration): the first wave copies the queue
  uint32_t stamp;                         //   counters + `stamp` into this host-mapped ring slot (publish_counters)
};

hipError_t launch_init(const RenderArgs &a, hipStream_t st);
hipError_t launch_logic(const RenderArgs &a, bool deferred_microfacet, hipStream_t st);
hipError_t launch_probe_primary(const RenderArgs &a, hipStream_t st);
hipError_t launch_gather_primary(const RenderArgs &a, int32_t *idx, uint32_t *tb, hipStream_t st);
hipError_t launch_iota(uint32_t *q, uint32_t n, hipStream_t st);
hipError_t launch_fill4(float4 *p, uint32_t n, float4 v, hipStream_t st);
// mirt_trace_static: caller's n x 8 ray records (device) -> ray_o / ray_d / identity queue / queue counter, and back
hipError_t launch_stage_rays(const float *rays8, uint32_t n, float4 *ray_o, float4 *ray_d, uint32_t *queue, uint32_t *qcount_slot,
                             hipStream_t st);
hipError_t launch_unstage_hits(uint32_t n, int any, const float4 *hit, const uint32_t *occ, const float *any_t, int32_t *idx_out,
                               uint32_t *tbits_out, hipStream_t st);
hipError_t launch_sum_planes(const float4 *planes, uint32_t n_planes, uint32_t n, float4 *out, int add_to_out, hipStream_t st);
hipError_t launch_tonemap(const void *accum, uint32_t n, const float *table, uint32_t *rgb /* r | g << 8 | b << 16 */, hipStream_t st);
hipError_t launch_trace(const TraceArgs &a, bool any, uint32_t traversal, uint32_t grid_blocks, hipStream_t st);
size_t trace4_lds_bytes();
int trace4_blocks_per_cu();

}  // namespace mirt
