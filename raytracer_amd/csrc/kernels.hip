// gfx950 kernels of the wavefront path tracer (see DESIGN.md for the data layout and the roofline of each).
//
//   k_init            per-slot path state, reference seed procedure (renderer.cuh:50-51)
//   k_logic           fold shadow results -> finish sample -> shade the hit (raytracer.cuh:30-126) -> emit the next
//                     extension ray and the shadow rays into compacted queues (wave64 ballot / mbcnt)
//   k_trace_fast<..>  traversal 0: persistent-thread ordered traversal of the 8-wide quantised SAH tree (LDS stack)
//   k_redo<ANY>       the reference walk, one wave per ray, for the rays k_trace_fast sets aside
//   k_trace4<ANY,..>  traversal 1: the reference tree folded to 4-wide nodes, reference visiting order (exact)
//   k_trace2<ANY>     traversal 2: the reference's own BVH2 walk (bvh.cuh:49-107), one ray per lane (cross-check)
//   k_sum_planes      sample streams / progressive passes: partial sums added in a fixed order
//   k_tonemap         film: divide + threshold-table tonemap (util/image.cuh:36-48)
//
// Built with -ffp-contract=off: the reference's g++ build never fuses a*b+c.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "shade.hpp"

// ============================================================================================ build-time knobs
// -DNAME=value on the hipcc line (tools/build_variant.sh builds a variant library, tools/ab.sh compares them on one box).
// The defaults are the measured best; what the other settings cost is in DESIGN.md 5.
#ifndef MIRT_LATE_ARGS
#define MIRT_LATE_ARGS 1   // k_logic's late phases read the argument block next to their use: 444 -> 22 SGPR spill instructions, 127 VGPR
#endif
#ifndef MIRT_LOGIC_WAVES
#define MIRT_LOGIC_WAVES 0   // k_logic: amdgpu_waves_per_eu(min); 0 = the compiler's choice.  4..6 measured: nothing / slower (spills)
#endif
#ifndef MIRT_LOGIC_EAGER
#define MIRT_LOGIC_EAGER 1   // k_logic: 1 = a slot's hit and running total are requested together with its state record (one dependent
#endif                       // memory round trip less, 32 B wasted per finished slot a launch looks at); 2 = its shadow results too
#ifndef MIRT_ADMIT_K
#define MIRT_ADMIT_K 1.0e-5f   // own-box admission pre-test: barycentric margin factor
#endif
#ifndef MIRT_ADMIT_LEVEL
#define MIRT_ADMIT_LEVEL 3   // own-box admission: 3 = the full test; 0 = off (timing experiments only: not exact)
#endif
#ifndef MIRT_TRACE_WAVES
#define MIRT_TRACE_WAVES 0   // k_trace_fast: amdgpu_waves_per_eu(n, n); 0 = the compiler's choice (72 VGPR, 7 waves).  8: spills, +13 %
#endif
#ifndef MIRT_SLACK
#define MIRT_SLACK 2.5e-5f   // relative slack of the distance pruning (triangle scenes).  1e-3: +2 %
#endif
#ifndef MIRT_SLACK_SPH
#define MIRT_SLACK_SPH 2.5e-5f   // the same for sphere scenes (their boxes are dilated instead, k_trace_fast); 1.2e-3 until round 2
#endif
#ifndef MIRT_SPH_LATERAL
#define MIRT_SPH_LATERAL 9.5e-4f   // sphere scenes: sqrt of the relative error bound of c = a.a - b*b, 15 x 2^-24 (k_trace_fast)
#endif
#ifndef MIRT_THIN
#define MIRT_THIN 1   // launches with very few rays spread thin over the grid's waves (tail launches 124 -> 90 us)
#endif
#ifndef MIRT_THIN_MIN
#define MIRT_THIN_MIN 2u   // fewest rays per wave of a thin launch
#endif
#ifndef MIRT_XCD_SLICES
#define MIRT_XCD_SLICES 0   // XCD x takes the x-th eighth of the per-workgroup slices.  Measured: nothing
#endif
#ifndef MIRT_CHUNK_MAX
#define MIRT_CHUNK_MAX 256u   // largest pool fetch of a wave, in rays.  64 ... 1024: within noise
#endif
#ifndef MIRT_CHUNK_DIV
#define MIRT_CHUNK_DIV 2u   // pool fetches per wave the pool is sized for
#endif
#ifndef MIRT_FAR_FIRST
#define MIRT_FAR_FIRST 0   // closest-hit rays push the farthest waiting child first: -2 % node visits, no time gain
#endif
#ifndef MIRT_LEAF_BATCH
#define MIRT_LEAF_BATCH 1   // primitives of a leaf requested and tested per trip.  2: +4 % (74-78 VGPR), 3: +15 % (90 VGPR)
#endif
#ifndef MIRT_LEAF_BATCH_ANY
#define MIRT_LEAF_BATCH_ANY MIRT_LEAF_BATCH   // the same for the occlusion kernel
#endif

namespace mirt {

// ============================================================================================ helpers
__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {       // set bits of `mask` below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}

// Wave-aggregated reservation of `n_each` consecutive queue entries for every lane that calls it (the calling lanes are
// the active lanes of a divergent branch): one atomic per wave.
__device__ __forceinline__ uint32_t wave_reserve(uint32_t *counter, uint32_t n_each) {
  const uint64_t m = __ballot(1);
  const uint32_t rank = lane_rank(m);
  uint32_t base = 0;
  if (rank == 0) base = atomicAdd(counter, (uint32_t) __popcll(m) * n_each);
  base = __builtin_amdgcn_readfirstlane(base);
  return base + rank * n_each;
}

__device__ __forceinline__ void pslot_to_pixel(const RenderArgs &a, uint32_t pslot, uint32_t &px, uint32_t &py, bool &valid) {
  // pixel slots are 8x8 pixel tiles (one wave = one tile); tiles are dealt round-robin over the processes of a tile partition
  const uint32_t tile_local = pslot >> 6, lane = pslot & 63;
  const uint32_t tile = tile_local * a.tile_world + a.tile_rank;
  const uint32_t ty = fdiv(tile, a.fd_tiles_x), tx = tile - ty * a.tiles_x;
  px = tx * 8 + (lane & 7);
  py = ty * 8 + (lane >> 3);
  valid = (ty < a.tiles_y) && px < a.width && py < a.height;
}
__device__ __forceinline__ void slot_to_pixel(const RenderArgs &a, uint32_t slot, uint32_t &px, uint32_t &py, bool &valid) {
  // with sample streams the slot array repeats once per stream
  pslot_to_pixel(a, a.streams > 1 ? slot - fdiv(slot, a.fd_pixel_slots) * a.n_pixel_slots : slot, px, py, valid);
}

// ============================================================================================ k_init
// A slot's state before its first sample: the reference's seed procedure (renderer.cuh:50-51) for its pixel, sample index 1,
// no ray; slots that cover no pixel (ragged tiles) are finished from the start.
__device__ __forceinline__ uint4 initial_state(const RenderArgs &a, uint32_t s) {
  uint32_t px, py;
  bool valid;
  slot_to_pixel(a, s, px, py, valid);
  uint4 st = make_uint4(0, 1, 0, 0);
  if (!valid) {
    st.z = F_DONE;
  } else if (a.rng_mode == 0) {
    Rng r;
    r.mode = 0; r.key = 0; r.sample = 0;
    r.state = (0x42ffffffU & (py * a.width + px)) ^ a.seed;          // renderer.cuh:50
    for (int k = 0; k < 10; k++) rng_next(r);                        // renderer.cuh:51
    st.x = r.state;
  }
  return st;
}

// (mirt_primary_hits only: a frame's first k_logic initialises its slots itself, RenderArgs::first)
__global__ void k_init(RenderArgs a) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n_slots) return;
  const uint4 st = initial_state(a, s);
  a.st[s] = st;
  a.mask[s] = mk_rec3(1, 1, 1);
  a.total[s] = make_float4(0, 0, 0, 0);
  a.accum[s] = make_float4(0, 0, 0, 0);
}

// ============================================================================================ k_logic
// A work item = (share of a pixel's samples, pixel slot) = share * n_pixel_slots + pixel slot.  Slot s of a launch starts with
// item stream_first * n_pixel_slots + s (with sample streams that is all it ever renders: share = its stream); with the
// share pool (RenderArgs::pool) it goes on with items from the pool.  The share's samples: [first, first + count).
__device__ __forceinline__ void item_share(const RenderArgs &a, uint32_t item, uint32_t &share, uint32_t &pslot, uint32_t &first,
                                           uint32_t &count) {
  share = 0; pslot = item; first = 0; count = a.spp;
  if (a.shares > 1) {
    share = fdiv(item, a.fd_pixel_slots);
    pslot = item - share * a.n_pixel_slots;
    const uint32_t per = a.share_len;
    first = share * per;
    count = (share == a.shares - 1) ? a.spp - per * (a.shares - 1) : per;     // (the pool needs spp % shares == 0: count == per)
  }
}

// Share pool: a slot takes its next items from the counter of its HOME GROUP, the `pool_group` neighbouring pixel slots
// (whole 8x8 tiles) its own pixel slot lies in, so a slot only ever renders pixels of its neighbourhood.  (One pool for
// the whole frame - any slot renders any pixel - was measured first, conference* 1080p 64 spp: the traversals got 5 % faster,
// k_logic 2.5 x slower, the frame 120 -> 176 ms: 116 M sixteen-byte partial sums scattered over 2 GB of planes at 2.4 G
// writes/s, and one counter taking an atomic per workgroup step at its 11 ns each.  Per group the counters are thousands,
// the plane writes stay inside a few KB per plane, and neighbouring lanes keep tracing neighbouring rays.)
// Counter value j of group g -> item: share `streams + j / size`, pixel slot `g * pool_group + j % size`.
__device__ __forceinline__ uint32_t pool_group_size(const RenderArgs &a, uint32_t g) {
  const uint32_t lo = g * a.pool_group, left = a.n_pixel_slots - lo;
  return left < a.pool_group ? left : a.pool_group;
}
__device__ __forceinline__ uint32_t pool_item(const RenderArgs &a, uint32_t g, uint32_t size, uint32_t j) {
  const uint32_t sh = size == a.pool_group ? fdiv(j, a.fd_pool_group) : j / size;     // (a ragged last group: plain division)
  return (a.streams + sh) * a.n_pixel_slots + g * a.pool_group + (j - sh * size);
}

struct PathCtx {
  Rng rng;
  uint32_t sample_base; // index of this slot's first sample in the pixel's sample sequence (counter mode)
  uint32_t i;          // 1-based index of the sample being accumulated (raytracer.cuh:97)
  uint32_t flags;
  uint32_t depth;
};

// What a slot appended this iteration; queue positions are handed out per workgroup at the end of k_logic.
struct Emit {
  uint32_t ext;        // 0 none, 1 camera ray (front of the extension queue), 2 bounced ray (back of the same array)
  uint32_t nsh;        // shadow rays written to entries k * n_slots + s, k < nsh (0 or n_lights * spl)
  uint32_t cull;       // bit k: entry k carries no ray (RenderArgs::cull_zero)
};

__device__ __forceinline__ void emit_extension(const RenderArgs &a, uint32_t s, const DRay &r, bool primary, Emit &em) {
  a.ray_o[s] = make_float4(r.o.x, r.o.y, r.o.z, r.min_t);
  a.ray_d[s] = make_float4(r.d_arg.x, r.d_arg.y, r.d_arg.z, r.max_t);
  em.ext = primary ? 1u : 2u;
}

// raytracer.cuh:100-107: jittered film point -> camera ray.  `sample_index` only matters in counter mode.
__device__ __forceinline__ void generate_camera_ray(const RenderArgs &a, uint32_t s, uint32_t px, uint32_t py,
                                                    PathCtx &c, uint32_t sample_number, Emit &em) {
  if (c.rng.mode == 1) {
    c.rng.state = 0;
    c.rng.sample = c.sample_base + sample_number - 1;
  }
  float gx = rng_next(c.rng);
  float gy = rng_next(c.rng);
  float fx = (float) px + gx, fy = (float) py + gy;
  DRay r = camera_generate_ray(a.cam, fx / (float) a.width, fy / (float) a.height, c.rng);
  emit_extension(a, s, r, true, em);
  c.depth = 0;
  c.flags |= F_HAS_RAY;
}

// Tables every lane reads at the same address (lights) are read through the constant address space: scalar loads into
// SGPRs instead of one VGPR per value and lane.
typedef const float __attribute__((address_space(4))) *uniform_floats;

// One slot, one iteration.  MF = false is the main kernel: it is compiled without the microfacet BSDF (whose double
// precision pow/exp/tan/erf would set the register budget of every slot) and hands slots that hit such a material,
// untouched, to the MF = true kernel through mf_queue.
// The kernel's argument block again, read where it is used.  k_logic's RenderArgs is ~60 scalars (25 array pointers, the
// camera, counts): read once at kernel entry they stay live through the whole slot function, far more than the 102 SGPRs
// hold, and the compiler parks them in VGPR lanes (v_writelane / v_readlane + hazard nops: ~450 such instructions in the
// kernel).  The late phases of logic_slot therefore read their fields through this reference: the same kernarg segment,
// behind an opaque barrier, so that those scalar loads (scalar-cache hits) are issued next to their use.  Only valid in
// a kernel whose first argument is the RenderArgs (k_logic).
__device__ __forceinline__ const RenderArgs &late_args(const RenderArgs &entry) {
#if MIRT_LATE_ARGS
  auto p = __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const RenderArgs *) p;
#else
  return entry;
#endif
}

// What logic_slot leaves in registers for slot_commit: the camera ray a slot wants is generated there, after the
// workgroup has fetched items from the share pool for the slots that asked (k_logic), and the slot's state is stored there.
struct SlotTail {
  PathCtx c;
  V3 total;
  float s1;
  float4 acc4;
  uint32_t item;        // the item of the path in flight (its pixel, its share of the samples)
  uint32_t n_shadow;
  uint32_t want;        // camera ray wanted: 0 none, 1 = sample cam_n of `item`, 2 = the first sample of an item from the pool
  uint32_t cam_n;
  float divisor;
  bool acc_dirty, finished, skip;
};
constexpr uint32_t NO_ITEM = 0xFFFFFFFFu;

// What k_logic requests together with a slot's state record, before it knows what the slot needs (MIRT_LOGIC_EAGER).
struct SlotEager {
  float4 h, tot4;
  rec3 dm, sc0, sc1;
  uint32_t occ0, occ1;
};
template<bool MF>
__device__ __forceinline__ void logic_slot(const RenderArgs &a0, const uint32_t s, const uint4 st, Emit &em, SlotTail &T, const bool first = false,
                                           const SlotEager *eager = nullptr) {
  const uint32_t flags0 = st.z & 0xFF;
  uint32_t n_shadow = (st.z >> 8) & 0xFFFF;
  const bool has_ray = (flags0 & F_HAS_RAY) != 0;

  // ---- 0. every load this slot can need, issued in dependency order before anything waits on one of them
  float4 h = make_float4(0, 0, 0, u2f(0xFFFFFFFFu));
  if (has_ray) h = (MIRT_LOGIC_EAGER >= 1 && eager) ? eager->h : a0.hit[s];
  const float4 tot4 = first ? make_float4(0, 0, 0, 0)                    // (a frame's first iteration: nothing to read yet)
                            : ((MIRT_LOGIC_EAGER >= 1 && eager) ? eager->tot4 : a0.total[s]);
  float4 acc4 = make_float4(0, 0, 0, 0);              // loaded when a sample is finished (once in a few iterations)
  float4 ro = make_float4(0, 0, 0, 0), rd = ro;
  rec3 m4 = mk_rec3(0, 0, 0), dm = m4, sc0 = m4, sc1 = m4;
  uint32_t occ0 = 1, occ1 = 1;
  if (n_shadow) {
    if (MIRT_LOGIC_EAGER >= 2 && eager) {
      dm = eager->dm; occ0 = eager->occ0; sc0 = eager->sc0;
      if (n_shadow > 1) { occ1 = eager->occ1; sc1 = eager->sc1; }
    } else {
      dm = a0.dmask[s];
      occ0 = a0.sh_occ[s];
      sc0 = a0.sh_c[s];
      if (n_shadow > 1) {
        occ1 = a0.sh_occ[(size_t) a0.n_slots + s];
        sc1 = a0.sh_c[(size_t) a0.n_slots + s];
      }
    }
  }
  const int32_t prim = (int32_t) f2u(h.w);
  const bool shade = has_ray && prim >= 0;
  float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0;
  if (shade) {                                        // (a miss needs neither the ray nor the path's mask: 48 B less per miss)
    m4 = a0.mask[s];
    ro = a0.ray_o[s];
    rd = a0.ray_d[s];
    n0 = a0.tri_nrm[3 * (size_t) prim];
    n1 = a0.tri_nrm[3 * (size_t) prim + 1];
    n2 = a0.tri_nrm[3 * (size_t) prim + 2];
  }
  DBsdf bsdf;
  bsdf.type = BSDF_DIFFUSE;
#pragma unroll
  for (int k = 0; k < 8; k++) bsdf.p[k] = 0.f;
  if (shade) {
    const float4 *mp = a0.mats + 3 * (size_t) f2u(n2.y);
    const float4 b0 = mp[0], b1 = mp[1], b2 = mp[2];
    bsdf.type = (int32_t) f2u(b0.x);
    bsdf.p[0] = b0.y; bsdf.p[1] = b0.z; bsdf.p[2] = b0.w;
    bsdf.p[3] = b1.x; bsdf.p[4] = b1.y; bsdf.p[5] = b1.z; bsdf.p[6] = b1.w;
    bsdf.p[7] = b2.x;
  }
  T.want = 0;
  T.skip = false;
  if (!MF && shade && bsdf.type == BSDF_MICROFACET) {
    const uint32_t pos = wave_reserve(a0.mf_count, 1);
    a0.mf_queue[pos] = s;
    T.skip = true;
    return;
  }

  const RenderArgs &a = late_args(a0);                 // from here on: fields read next to their use
  const bool pool = a.pool != 0;
  // the item of the path in flight: the slot's static one, or (share pool) whatever it took last, kept in total[].w
  const uint32_t item = (pool && !first) ? f2u(tot4.w) : a.stream_first * a.n_pixel_slots + s;
  uint32_t share, pslot, share_first, share_count;
  item_share(a, item, share, pslot, share_first, share_count);
  uint32_t px, py;
  bool valid;
  pslot_to_pixel(a, pslot, px, py, valid);

  PathCtx c;
  c.rng.mode = a.rng_mode;
  c.rng.state = st.x;
  c.rng.key = (a.rng_mode == 1) ? ctr_key(py * a.width + px, a.seed) : 0;
  c.i = st.y;
  c.flags = flags0;
  c.depth = st.w;
  c.sample_base = a.sample_begin + share_first;
  // counter mode: the path in flight belongs to sample i, or - started ahead of the finalisation - to sample i + 1 of the same
  // item, or to the first sample of the item the slot took from the pool when its share ran out (F_NEWITEM)
  c.rng.sample = c.sample_base + ((c.flags & F_NEWITEM) ? 1u : c.i + ((c.flags & F_AHEAD) ? 1u : 0u)) - 1u;

  V3 total = mk(tot4.x, tot4.y, tot4.z);
  float s1 = tot4.w;

  // ---- 1. fold the shadow rays emitted at the previous hit (raytracer.cuh:43-47, :68)
  if (n_shadow) {
    V3 color = mk(0, 0, 0);
    if (occ0 == 0) color = vadd(color, mk(sc0.x, sc0.y, sc0.z));
    if (n_shadow > 1 && occ1 == 0) color = vadd(color, mk(sc1.x, sc1.y, sc1.z));
    for (uint32_t k = 2; k < n_shadow; k++) {
      size_t e = (size_t) k * a.n_slots + s;
      if (a.sh_occ[e] == 0) {
        rec3 cc = a.sh_c[e];
        color = vadd(color, mk(cc.x, cc.y, cc.z));
      }
    }
    total = vadd(total, vmul(mk(dm.x, dm.y, dm.z), color));
    n_shadow = 0;
  }

  bool finished = false, acc_dirty = false;
  bool item_done = false;                              // share pool: the accumulator's item is complete and no path is in flight
  float divisor = 0.f;
  uint32_t want = 0, cam_n = 0;

  // finish sample c.i: raytracer.cuh:111-122
  auto finalize = [&]() {
    if (!acc_dirty) acc4 = a.accum[s];
    acc4.x += total.x; acc4.y += total.y; acc4.z += total.z;
    acc_dirty = true;
    float illum = villum(total);
    if (!pool) s1 += illum;                            // (pool: the word carries the item; the adaptive stop is off)
    total = mk(0, 0, 0);
    bool stop = false;
    if (a.adaptive && (c.i % a.batch) == 0) {
      float variance = (float) (((double) s1 - ((double) s1 * (double) s1) / (double) (int) c.i) / (double) ((int) c.i - 1));
      float mean = s1 / (float) (int) c.i;
      stop = (1.96 * (double) sqrtf(variance / (float) (int) c.i)) <= (double) (a.tol * mean);
    }
    if (stop) {
      finished = true;
      divisor = (float) ((int) c.i - 1);          // raytracer.cuh:124 after `break`: i was not incremented
    } else {
      c.i++;
      if (c.i > share_count) {
        if (!pool) {
          finished = true;
          divisor = (float) ((int) c.i - 1);
        } else {
          // the share is complete: its sum goes to the plane of the ACCUMULATOR's item (accum[].w), which is not the item of
          // the path in flight if that one was started ahead from the pool
          const uint32_t item_acc = f2u(acc4.w);
          const uint32_t sh_acc = fdiv(item_acc, a.fd_pixel_slots);
          uint32_t qx, qy;
          bool qv;
          pslot_to_pixel(a, item_acc - sh_acc * a.n_pixel_slots, qx, qy, qv);
          a.out_accum[(size_t) sh_acc * a.width * a.height + (size_t) qy * a.width + qx] =
              make_float4(acc4.x, acc4.y, acc4.z, (float) ((int) c.i - 1));
          if (c.flags & F_NEWITEM) {                   // the path in flight already belongs to the next item
            acc4 = make_float4(0, 0, 0, u2f(item));
            c.i = 1;
            c.flags &= ~F_NEWITEM;
          } else {
            item_done = true;
          }
        }
      }
    }
  };

  // ---- 2. a sample that ended at the previous hit but still waited for its shadow rays
  if (c.flags & F_PENDING) {
    c.flags &= ~F_PENDING;
    finalize();
    c.flags &= ~F_AHEAD;
  }

  if (item_done) {
    want = 2;
  } else if (!finished) {
    if (!(c.flags & F_HAS_RAY)) {
      // ---- first sample of the pixel
      want = 1;
      cam_n = c.i;
    } else {
      // ---- 3. the extension ray of this path was traced: shade (estimate_global_lighting body, raytracer.cuh:60-88)
      V3 mask = mk(m4.x, m4.y, m4.z);
      bool path_ends = false;
      DRay next;
      if (prim < 0) {
        path_ends = true;                                   // miss: break (raytracer.cuh:62)
      } else {
        const V3 origin = mk(ro.x, ro.y, ro.z);
        const V3 direction = vunit(mk(rd.x, rd.y, rd.z));   // Ray::direction (ray.cuh:20)
        const float by = h.y, bz = h.z;
        const float bx = 1 - by - bz;                       // primitives.cuh:147
        DIsect is;
        if (f2u(n2.w) != 0u)                                           // sphere: (o + max_t * d) - origin, primitives.cuh:134
          is.normal = vsub(vadd(origin, vlscale(h.x, direction)), mk(n0.x, n0.y, n0.z));
        else
          is.normal = vadd(vadd(vlscale(bx, mk(n0.x, n0.y, n0.z)), vlscale(by, mk(n0.w, n1.x, n1.y))),
                           vlscale(bz, mk(n1.z, n1.w, n2.x)));         // primitives.cuh:156
        isect_compute(is, origin, direction, h.x);
        if (bsdf.type == BSDF_EMISSION) total = vadd(total, vmul(mask, mk(bsdf.p[0], bsdf.p[1], bsdf.p[2])));
        else total = vadd(total, vmul(mask, mk(0, 0, 0)));  // raytracer.cuh:65 (keeps inf/NaN * 0 semantics)

        if (!bsdf_is_delta(bsdf.type)) {
          // estimate_direct_lighting (raytracer.cuh:30-51): one shadow ray per light sample, traced by k_trace*<ANY>
          const V3 bump = vscale(is.normal, 0.001f);
          const V3 so = vadd(is.hit_point, bump);
          uint32_t k = 0;
          for (uint32_t li = 0; li < a.n_lights; li++) {
            DLight L;
            {
              uniform_floats lp = (uniform_floats) a.lights + 17 * (size_t) li;
              L.type = (int32_t) f2u(lp[0]);
#pragma unroll
              for (int q = 0; q < 16; q++) L.p[q] = lp[1 + q];
            }
            for (uint32_t j = 0; j < a.spl; j++, k++) {
              V3 to_light;
              float dist, pdf;
              V3 Lr = light_sample(L, so, to_light, dist, pdf, c.rng);
              V3 f = bsdf_f<MF>(bsdf, is.o_out, mmulv(is.w2o, to_light));
              V3 sdir = vunit(to_light);                                // shadow_ray.direction
              float cosv = smax(vdot(is.normal, vunit(sdir)), 0.f);     // raytracer.cuh:45
              V3 contrib = vdiv(vdiv(vscale(vmul(f, Lr), cosv), pdf), (float) a.spl);
              size_t e = (size_t) k * a.n_slots + s;
              if (a.cull_zero && contrib.x == 0.f && contrib.y == 0.f && contrib.z == 0.f) {
                // a light behind the surface: whatever the shadow ray finds, the sum does not change (x + 0 = x)
                a.sh_occ[e] = 1u;
                em.cull |= 1u << k;
                continue;
              }
#if !MIRT_SHARED_ORIGIN
              a.sh_o[e] = make_float4(so.x, so.y, so.z, MIRT_EPS);
#endif
              if (k != a.const_k)                                  // (a directional light's rays: their records hold the light's direction already)
                a.sh_d[e] = make_float4(to_light.x, to_light.y, to_light.z, dist - MIRT_EPS);
              a.sh_c[e] = mk_rec3(contrib.x, contrib.y, contrib.z);
            }
          }
          n_shadow = k;
          em.nsh = k;
#if MIRT_SHARED_ORIGIN
          a.sh_o[s] = make_float4(so.x, so.y, so.z, MIRT_EPS);         // the one origin of this hit's shadow rays
#endif
          a.dmask[s] = mk_rec3(mask.x, mask.y, mask.z);
        }

        float cpdf = smax(mask.x, smax(mask.y, mask.z));                // raytracer.cuh:71
        if (!(rng_next(c.rng) < cpdf)) {
          path_ends = true;                                             // :72-74
        } else {
          float pdf;
          V3 bmask;
          V3 o_in = bsdf_sample<MF>(bsdf, is.o_out, bmask, pdf, c.rng, a.glass_int_abs != 0);
          if (pdf == 0) {
            path_ends = true;                                           // :79
          } else {
            V3 bump = vscale(is.normal, 0.001f);
            bump = o_in.z > 0 ? bump : vneg(bump);
            next.o = vadd(is.hit_point, bump);
            next.d_arg = mmulv(is.o2w, o_in);
            next.min_t = MIRT_EPS;
            next.max_t = MIRT_INF;
            mask = vdiv(vdiv(vscale(vmul(mask, bmask), o_in.z), pdf), smin(cpdf, 1.f));   // :85
            c.depth++;
            if (c.depth > a.max_depth) path_ends = true;                // loop bound `i <= max_ray_depth` (:60)
          }
        }
      }

      if (!path_ends) {
        a.mask[s] = mk_rec3(mask.x, mask.y, mask.z);
        emit_extension(a, s, next, false, em);
      } else {
        c.flags &= ~F_HAS_RAY;
        if (n_shadow > 0) {
          // the sample is complete except for its last shadow rays: finish it next iteration, and meanwhile already
          // start the following sample (its camera ray is traced alongside those shadow rays) - of this share, or with
          // the share pool of whatever item the pool hands out
          c.flags |= F_PENDING;
          if (c.i + 1 <= share_count) {
            c.flags |= F_AHEAD;
            want = 1;
            cam_n = c.i + 1;
          } else if (pool) {
            want = 2;
          }
        } else {
          finalize();
          if (item_done) {
            want = 2;
          } else if (!finished) {
            want = 1;
            cam_n = c.i;
          }
        }
      }
    }
  }

  T.c = c; T.total = total; T.s1 = s1; T.acc4 = acc4; T.item = item; T.n_shadow = n_shadow;
  T.want = want; T.cam_n = cam_n; T.divisor = divisor; T.acc_dirty = acc_dirty; T.finished = finished;
}

// Second half of a slot's iteration: the camera ray it asked for, and its state back to memory.  `new_item`: what the pool
// gave a slot that asked for an item (NO_ITEM: the pool is empty).
__device__ __forceinline__ void slot_commit(const RenderArgs &a0, const uint32_t s, SlotTail &T, Emit &em, uint32_t new_item,
                                            uint32_t group, uint32_t group_size) {
  if (T.skip) return;
  const RenderArgs &a = late_args(a0);
  PathCtx &c = T.c;
  if (T.want == 2) {
    uint32_t it = new_item, share = 0, pslot = 0, first = 0, count = 0, px = 0, py = 0;
    while (it != NO_ITEM) {
      // an item of a pixel slot that covers no pixel (ragged tiles at the image's edge) is nobody's work: take another
      bool valid;
      item_share(a, it, share, pslot, first, count);
      pslot_to_pixel(a, pslot, px, py, valid);
      if (valid) break;
      const uint32_t j = atomicAdd(a.pool_next + group, 1u);
      it = j < (a.shares - a.streams) * group_size ? pool_item(a, group, group_size, j) : NO_ITEM;
    }
    if (it != NO_ITEM) {
      T.item = it;
      if (c.flags & F_PENDING) {
        c.flags |= F_AHEAD | F_NEWITEM;                // the finished share still waits for its shadow rays (logic_slot, finalize)
      } else {
        T.acc4 = make_float4(0, 0, 0, u2f(it));
        T.acc_dirty = true;
        c.i = 1;
      }
      c.sample_base = a.sample_begin + first;
      c.rng.key = ctr_key(py * a.width + px, a.seed);
      a.mask[s] = mk_rec3(1, 1, 1);
      generate_camera_ray(a, s, px, py, c, 1, em);
    } else if (!(c.flags & F_PENDING)) {
      T.finished = true;                               // nothing left for this slot
    }
  } else if (T.want == 1) {
    uint32_t share, pslot, first, count, px, py;
    bool valid;
    item_share(a, T.item, share, pslot, first, count);
    pslot_to_pixel(a, pslot, px, py, valid);
    a.mask[s] = mk_rec3(1, 1, 1);
    generate_camera_ray(a, s, px, py, c, T.cam_n, em);
  }
  if (T.finished) {
    c.flags = F_DONE;
    T.n_shadow = 0;
    if (!a.pool) {
      uint32_t share, pslot, first, count, px, py;
      bool valid;
      item_share(a, T.item, share, pslot, first, count);
      pslot_to_pixel(a, pslot, px, py, valid);
      a.out_accum[(size_t) share * a.width * a.height + (size_t) py * a.width + px] =
          make_float4(T.acc4.x, T.acc4.y, T.acc4.z, T.divisor);
    }
  }
  a.total[s] = make_float4(T.total.x, T.total.y, T.total.z, a.pool ? u2f(T.item) : T.s1);
  if (T.acc_dirty) a.accum[s] = T.acc4;
  a.st[s] = make_uint4(c.rng.state, c.i, c.flags | (T.n_shadow << 8), c.depth);
}

// MF = false: one thread per slot.  MF = true: workgroup-stride over the slots the main kernel deferred.
// Queue positions are reserved once per workgroup and queue (three atomics, each on a line of its own): a slot that is
// not finished always leaves a ray in one of the queues, so the queue counters double as the "anything left?" test and
// as the ray statistics (the host adds them up per iteration).
#if MIRT_LOGIC_WAVES
#define MIRT_LOGIC_ATTR __attribute__((amdgpu_waves_per_eu(MIRT_LOGIC_WAVES)))
#else
#define MIRT_LOGIC_ATTR
#endif
// FIRST: the frame's first launch (RenderArgs::first), which also initialises the slots.  An instantiation of its own: the
// initialisation costs k_logic 8 VGPRs and with them a resident wave per SIMD, which only one launch per frame should pay.
template<bool MF, bool FIRST = false>
__global__ void __launch_bounds__(256) MIRT_LOGIC_ATTR k_logic(RenderArgs a) {
  __shared__ uint32_t wave_cnt[4][5];
  __shared__ uint32_t block_base[5], block_tot[5];
  const uint32_t tid = threadIdx.x, wave = tid >> 6;
  const bool listed = !MF && a.live_in != nullptr && (uint64_t) *a.live_in_count * 16u <= (uint64_t) a.n_slots * a.live_16th;
  const uint32_t n = MF ? *a.mf_count : (listed ? *a.live_in_count : a.n_slots);
  const uint32_t K = a.n_lights * a.spl;
  if (!MF && a.clear_next != nullptr && blockIdx.x == 0)
    for (uint32_t i = tid; i < a.clear_words; i += 256u) a.clear_next[i] = 0u;
  for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {
    const uint32_t i = base + tid;
    uint32_t s = 0;
    Emit em;
    em.ext = 0; em.nsh = 0; em.cull = 0;
    SlotTail T;
    T.want = 0;
    T.skip = true;
    if (i < n) {
      s = MF ? a.mf_queue[i] : (listed ? a.live_in[i] : i);
      if (!MF && FIRST) {
        // the frame's first iteration doubles as its initialisation (no k_init launch, no state to read): every slot starts
        // its first sample; the running sums start at zero (share pool: accum[].w names the slot's first item)
        const uint4 st = initial_state(a, s);
        if (st.z & F_DONE) {
          a.st[s] = st;
        } else {
          a.accum[s] = make_float4(0, 0, 0, a.pool ? u2f(a.stream_first * a.n_pixel_slots + s) : 0.f);
          logic_slot<MF>(a, s, st, em, T, true);
        }
      } else {
        const uint4 st = a.st[s];
#if MIRT_LOGIC_EAGER
        SlotEager eg;
        eg.h = a.hit[s];
        eg.tot4 = a.total[s];
        eg.dm = eg.sc0 = eg.sc1 = mk_rec3(0, 0, 0);
        eg.occ0 = eg.occ1 = 1;
#if MIRT_LOGIC_EAGER >= 2
        eg.dm = a.dmask[s];
        eg.occ0 = a.sh_occ[s];
        eg.sc0 = a.sh_c[s];
        if (K >= 2) {
          eg.occ1 = a.sh_occ[(size_t) a.n_slots + s];
          eg.sc1 = a.sh_c[(size_t) a.n_slots + s];
        }
#endif
        if (MF || !(st.z & F_DONE)) logic_slot<MF>(a, s, st, em, T, false, &eg);
#else
        if (MF || !(st.z & F_DONE)) logic_slot<MF>(a, s, st, em, T);
#endif
      }
    }
    // Share pool: the slots that have finished their item take the next ones of their home group, one atomic per wave and
    // group (a wave's 64 slots are one tile, hence one group, unless the launch walks the live-slot list; a counter only
    // grows: once it is past its group's last item nobody adds to it any more).
    uint32_t new_item = NO_ITEM, group = 0, group_size = 1;
    if (a.pool) {
      const bool asks = !T.skip && T.want == 2;
      group = fdiv(s - fdiv(s, a.fd_pixel_slots) * a.n_pixel_slots, a.fd_pool_group);
      group_size = pool_group_size(a, group);
      uint64_t m = __ballot(asks);
      while (m) {
        const uint32_t lead = (uint32_t) __builtin_amdgcn_readfirstlane((int) (__ffsll((long long) m) - 1));
        const uint32_t g0 = (uint32_t) __builtin_amdgcn_readlane((int) group, (int) lead);
        const uint64_t same = __ballot(asks && group == g0);
        const uint32_t items = (a.shares - a.streams) * pool_group_size(a, g0);
        uint32_t got = items;
        if ((tid & 63u) == lead &&
            __hip_atomic_load(a.pool_next + g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < items)
          got = atomicAdd(a.pool_next + g0, (uint32_t) __popcll(same));
        got = (uint32_t) __builtin_amdgcn_readlane((int) got, (int) lead);
        if (asks && group == g0) {
          const uint32_t j = got + lane_rank(same);
          if (got < items && j < items) new_item = pool_item(a, g0, group_size, j);
        }
        m &= ~same;
      }
    }
    slot_commit(a, s, T, em, new_item, group, group_size);
    // (a slot that is not finished has just emitted something; one handed to the MICROFACET kernel is listed by that one)
    const uint64_t m0 = __ballot(em.ext == 1), m1 = __ballot(em.ext == 2), m2 = __ballot(em.nsh != 0),
                   m3 = __ballot(em.ext != 0 || em.nsh != 0);
    uint32_t n_culled = 0;
    if (a.cull_zero)
      for (uint32_t k = 0; k < K; k++) n_culled += (uint32_t) __popcll(__ballot((em.cull >> k) & 1u));
    if ((tid & 63) == 0) {
      wave_cnt[wave][0] = (uint32_t) __popcll(m0);
      wave_cnt[wave][1] = (uint32_t) __popcll(m1);
      wave_cnt[wave][2] = (uint32_t) __popcll(m2);
      wave_cnt[wave][3] = (uint32_t) __popcll(m3);
      wave_cnt[wave][4] = n_culled;
    }
    __syncthreads();
    if (tid < 5) {
      const uint32_t tot = wave_cnt[0][tid] + wave_cnt[1][tid] + wave_cnt[2][tid] + wave_cnt[3][tid];
      block_tot[tid] = tot;
      block_base[tid] = tot ? atomicAdd(a.qcount + CTR_STRIDE * tid, tid == 2 ? tot * K : tot) : 0u;
    }
    __syncthreads();
    uint32_t off[4] = {0, 0, 0, 0};
    for (uint32_t w = 0; w < wave; w++) {
      off[0] += wave_cnt[w][0]; off[1] += wave_cnt[w][1]; off[2] += wave_cnt[w][2]; off[3] += wave_cnt[w][3];
    }
    if (em.ext == 1) a.ext_queue[block_base[0] + off[0] + lane_rank(m0)] = s;                     // camera rays: front
    if (em.ext == 2) a.ext_queue[a.n_slots - 1 - (block_base[1] + off[1] + lane_rank(m1))] = s;   // bounced rays: back
    if (em.nsh) {
      // light-major inside the workgroup's range: consecutive entries go to the same light from neighbouring pixels
      const uint32_t p = block_base[2] + off[2] + lane_rank(m2);
      for (uint32_t k = 0; k < K; k++)
        a.sh_queue[p + k * block_tot[2]] = ((em.cull >> k) & 1u) ? QUEUE_SKIP : k * a.n_slots + s;
    }
    if (em.ext != 0 || em.nsh != 0) a.live_out[block_base[3] + off[3] + lane_rank(m3)] = s;
    __syncthreads();
  }
}

// ============================================================================================ traversal
// The closest-hit launch of an iteration starts when k_logic has finished, so the queue counters are final: its first wave
// hands them to the host through host-mapped memory (counters, then - system-scope release - the stamp the host polls).
// No copy dispatch, no event.
__device__ __forceinline__ void publish_counters(const TraceArgs &a) {
  if (a.host_slot == nullptr || blockIdx.x != 0 || threadIdx.x >= 64) return;
  if (threadIdx.x < (uint32_t) (QC_WORDS / CTR_STRIDE)) a.host_slot[threadIdx.x] = a.qcount[CTR_STRIDE * threadIdx.x];
  __threadfence_system();
  if (threadIdx.x == 0) __hip_atomic_store(a.host_slot + 7, a.stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct LaneRay {
  V3 o, d, inv;          // origin, normalised direction (M-T), reciprocal of the ctor argument (slab test)
  float min_t, best_t;
  float b1, b2;
  int32_t best_idx;
  uint32_t id;
};

__device__ __forceinline__ bool finite3(V3 v) {
  return (fabsf(v.x) < MIRT_INF) && (fabsf(v.y) < MIRT_INF) && (fabsf(v.z) < MIRT_INF);
}

// SHARED: the launch may carry the shadow rays of a render (TraceArgs::o_slots).  Only the occlusion kernels are compiled with
// it: in the closest-hit kernel the few extra instructions moved the register allocation from 72 to 77 VGPRs (one resident wave less).
template<bool SHARED>
__device__ __forceinline__ void load_ray(const TraceArgs &a, uint32_t id, LaneRay &r, float *len2 = nullptr) {
  uint32_t oi = id;
  if (SHARED && a.o_slots && oi >= a.o_slots) {                 // shadow rays of a render: entry k * n_slots + slot starts at the slot's origin
    oi -= a.o_slots;
    if (oi >= a.o_slots) oi %= a.o_slots;
  }
  const float4 ro = a.ray_o[oi], rd = a.ray_d[id];
  r.o = mk(ro.x, ro.y, ro.z);
  const V3 d_arg = mk(rd.x, rd.y, rd.z);
  r.d = vunit(d_arg);                 // ray.cuh:20
  r.inv = vrcp(d_arg);                // ray.cuh:21 (of the argument: SURVEY Q1)
  r.min_t = ro.w;
  r.best_t = rd.w;
  r.b1 = 0; r.b2 = 0;
  r.best_idx = -1;
  r.id = id;
  if (len2) *len2 = rd.x * rd.x + rd.y * rd.y + rd.z * rd.z;       // squared length of the ctor direction
}

// Sphere arm of Primitive::intersect / has_intersection (primitives.cuh:64-85, :111-136).  The two differ only in an
// extra reject clause of the shadow version (:76) that the closest-hit version reaches two lines later (:128), so one
// function serves both: `cand` = everything but the comparison against max_t, `t` = the root the reference would keep.
struct SphereHit { bool geometric, cand; float t, t0, t1; };
__device__ __forceinline__ SphereHit sphere_eval(const float4 q0, const float4 q1, const LaneRay &r) {
  const V3 a = vsub(mk(q0.x, q0.y, q0.z), r.o);
  const float b = vdot(a, r.d);
  const float c = vdot(a, a) - b * b;
  const float r2 = q1.x;
  SphereHit h;
  h.geometric = !(b < 0) && !(c > r2);
  const float disc = sqrtf(r2 - c);
  h.t0 = b - disc;
  h.t1 = b + disc;
  h.t = (h.t0 > r.min_t) ? h.t0 : h.t1;
  h.cand = h.geometric && !(h.t1 <= r.min_t);
  return h;
}

// Moeller-Trumbore exactly as primitives.cuh:137-158 evaluates it (e1, e2 are stored: same subtraction, same bits).
// Returns the reference's accept decision against `max_t`.  Word 10 of the record: 0 triangle, 1 sphere.
__device__ __forceinline__ bool tri_test(const float4 *tris, uint32_t idx, const LaneRay &r, float max_t, float &t,
                                         float &by, float &bz, uint32_t &flat) {
  const float4 q0 = tris[3 * (size_t) idx], q1 = tris[3 * (size_t) idx + 1], q2 = tris[3 * (size_t) idx + 2];
  flat = f2u(q2.y);                   // index of this primitive in the reference's flat order
  if (f2u(q2.z) != 0u) {
    const SphereHit h = sphere_eval(q0, q1, r);
    t = h.t; by = 0.f; bz = 0.f;
    return h.cand && !(h.t > max_t);
  }
  const V3 p1 = mk(q0.x, q0.y, q0.z), e1 = mk(q0.w, q1.x, q1.y), e2 = mk(q1.z, q1.w, q2.x);
  const V3 s = vsub(r.o, p1);
  const V3 s1 = vcross(r.d, e2);
  const V3 s2 = vcross(s, e1);
  const float rc = 1.0f / vdot(s1, e1);
  t = rc * vdot(s2, e2);
  by = rc * vdot(s1, s);
  bz = rc * vdot(s2, r.d);
  const float bx = 1 - by - bz;
  return !(t < 0 || t <= r.min_t || t > max_t || bx < 0 || by < 0 || bz < 0);
}

// The reference's slab arithmetic (bbox.cuh:67-79) on the primitive's OWN bounding box [lo, hi].  Every box of the
// reference's tree above the primitive contains this box, and the slab arithmetic is monotone in the box bounds (lo' <= lo
// puts (lo' - o) * inv on the outer side of (lo - o) * inv, rounding included): their entry distances are <= this box's
// entry E, their exit distances >= its exit X - whatever the reference's tree looks like.  The reference tests them with
// tmax = min(exit, m), m = its max_t at that moment, and m >= t_hit when the hit is the nearest one.  So it reaches the
// primitive (and takes the same decision on it) if  E <= X,  X > min_t  and  E <= m.  The last holds when
// E <= t_hit * (1 + NEAR): an m below that is a hit within NEAR of this one, a tie the caller flags anyway.
// (Finite operands only: the caller has set aside rays with zero / non-finite components, for which the min/max chains
// below equal the reference's std::max/std::min.)
__device__ __forceinline__ bool own_box_admits(const V3 lo, const V3 hi, const LaneRay &r, float t_limit) {
  const float ax = (lo.x - r.o.x) * r.inv.x, bx = (hi.x - r.o.x) * r.inv.x;
  const float ay = (lo.y - r.o.y) * r.inv.y, by = (hi.y - r.o.y) * r.inv.y;
  const float az = (lo.z - r.o.z) * r.inv.z, bz = (hi.z - r.o.z) * r.inv.z;
  const float tmn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
  const float tmx = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
  return tmn <= tmx && tmx > r.min_t && tmn <= t_limit;
}

// The ordered traversal's variant: same arithmetic and accept decision, plus `near` = "the reference's outcome for this
// ray may hang on the reference's own boxes or visiting order, which no other tree can know": the caller hands the ray to
// the reference walk (k_redo).  Two reasons:
//   * the primitive is hit (a triangle inside its edges, a sphere at either root) at a distance within NEAR (relative) of
//     one of the two limits the accept test compares against: a tie, decided by the order of the reference's walk;
//   * the hit is accepted but the ray fails the reference's slab test on the primitive's own bounding box (own_box_admits):
//     a hit on a vertex or edge that sits on an edge of that box (entry ~ exit, decided in the last ulp), or a grazing
//     "hit" that Moeller-Trumbore reports from outside the triangle (its barycentrics are ill-conditioned when the ray runs
//     almost in the triangle's plane).  The reference finds such hits or not depending on its boxes.
template<bool SPH>
__device__ __forceinline__ bool tri_test_near(const float4 q0, const float4 q1, const float4 q2, const float *verts, const LaneRay &r,
                                              float max_t, float near_rel, float &t, float &by, float &bz, uint32_t &flat, bool &near) {
  flat = f2u(q2.y);
  if (SPH && f2u(q2.z) != 0u) {
    const SphereHit h = sphere_eval(q0, q1, r);
    t = h.t; by = 0.f; bz = 0.f;
    const float b0 = near_rel * fabsf(h.t0), b1 = near_rel * fabsf(h.t1);
    near = h.geometric && (fabsf(h.t0 - max_t) <= b0 || fabsf(h.t0 - r.min_t) <= b0 ||
                           fabsf(h.t1 - max_t) <= b1 || fabsf(h.t1 - r.min_t) <= b1);
    const bool accept = h.cand && !(h.t > max_t);
    if (accept) {                                         // get_bbox (primitives.cuh:166-169): origin -/+ (r, r, r)
      const V3 c = mk(q0.x, q0.y, q0.z), rr = mk(q0.w, q0.w, q0.w);
      near = near || !own_box_admits(vsub(c, rr), vadd(c, rr), r, t * (1.0f + near_rel));
    }
    return accept;
  }
  const V3 p1 = mk(q0.x, q0.y, q0.z), e1 = mk(q0.w, q1.x, q1.y), e2 = mk(q1.z, q1.w, q2.x);
  const V3 s = vsub(r.o, p1);
  const V3 s1 = vcross(r.d, e2);
  const V3 s2 = vcross(s, e1);
  const float rc = 1.0f / vdot(s1, e1);
  t = rc * vdot(s2, e2);
  by = rc * vdot(s1, s);
  bz = rc * vdot(s2, r.d);
  const float bx = 1 - by - bz;
  const bool inside = !(t < 0 || bx < 0 || by < 0 || bz < 0);
  const float band = near_rel * fabsf(t);                 // t finite here or the comparisons below are false / handled by the caller
  near = inside && (fabsf(t - max_t) <= band || fabsf(t - r.min_t) <= band);
  const bool accept = inside && !(t <= r.min_t || t > max_t);
  // Own-box admission, in three steps of rising cost and falling frequency.
  // 1. The hit lies well inside the triangle - every barycentric above ADMIT_K * (|t| + L) * L / |det| (L = upper bound of the
  //    edge lengths; |det| = 2 * area * |cos incidence|), i.e. at least ~1e-5 * t / |cos| away from every edge, some 25
  //    times what the slab arithmetic and the conditioning of the barycentrics can move it: the ray passes through the
  //    interior of the triangle's box with room to spare.  ~98 % of the hits stop here.
  // 2. own_box_admits on a box that is CONTAINED in the exact one: the box of p1, p1 + e1, p1 + e2 (each within 1.5 ulp of the
  //    exact vertex) shrunk by 4e-7 relative per side, except along axes in which the triangle is exactly flat (e1 = e2 = 0
  //    there: the coordinate is exact).  Monotonicity again: what the smaller box admits, the exact one admits.
  // 3. Otherwise (a hit on the rim of the box; an axis too thin to shrink): the exact vertices from memory.
  constexpr float ADMIT_K = MIRT_ADMIT_K;
#if MIRT_ADMIT_LEVEL == 0
  if (false) {
#else
  bool well_inside = true;
  if (accept) {
    asm volatile("" ::: "memory");                        // (keeps the pre-test inside the branch: most trips of an occlusion wave accept nothing)
    const float L = fabsf(e1.x) + fabsf(e1.y) + fabsf(e1.z) + fabsf(e2.x) + fabsf(e2.y) + fabsf(e2.z);   // >= every edge length
    well_inside = fminf(bx, fminf(by, bz)) > (ADMIT_K * fabsf(rc)) * ((fabsf(t) + L) * L);
  }
  if (__builtin_expect(!well_inside, 0)) {
#endif
    const V3 vb = vadd(p1, e1), vc = vadd(p1, e2);
    V3 lo = mk(fminf(p1.x, fminf(vb.x, vc.x)), fminf(p1.y, fminf(vb.y, vc.y)), fminf(p1.z, fminf(vb.z, vc.z)));
    V3 hi = mk(fmaxf(p1.x, fmaxf(vb.x, vc.x)), fmaxf(p1.y, fmaxf(vb.y, vc.y)), fmaxf(p1.z, fmaxf(vb.z, vc.z)));
    const float ux = (e1.x == 0.f && e2.x == 0.f) ? 0.f : 4.0e-7f * fmaxf(fabsf(lo.x), fabsf(hi.x));
    const float uy = (e1.y == 0.f && e2.y == 0.f) ? 0.f : 4.0e-7f * fmaxf(fabsf(lo.y), fabsf(hi.y));
    const float uz = (e1.z == 0.f && e2.z == 0.f) ? 0.f : 4.0e-7f * fmaxf(fabsf(lo.z), fabsf(hi.z));
    lo = mk(lo.x + ux, lo.y + uy, lo.z + uz);
    hi = mk(hi.x - ux, hi.y - uy, hi.z - uz);
    const float t_limit = t * (1.0f + near_rel);
    if (!(lo.x <= hi.x && lo.y <= hi.y && lo.z <= hi.z && own_box_admits(lo, hi, r, t_limit))) {
      const float *v = verts + 9 * (size_t) flat;         // get_bbox (primitives.cuh:170-175): min / max of the exact vertices
      const V3 a = mk(v[0], v[1], v[2]), b = mk(v[3], v[4], v[5]), c = mk(v[6], v[7], v[8]);
      lo = mk(fminf(a.x, fminf(b.x, c.x)), fminf(a.y, fminf(b.y, c.y)), fminf(a.z, fminf(b.z, c.z)));
      hi = mk(fmaxf(a.x, fmaxf(b.x, c.x)), fmaxf(a.y, fmaxf(b.y, c.y)), fmaxf(a.z, fmaxf(b.z, c.z)));
      near = near || !own_box_admits(lo, hi, r, t_limit);
    }
  }
  return accept;
}

// -------------------------------------------------------------------------------------------- k_trace2 (reference walk)
// bbox.cuh:67-79 with std::max / std::min argument order
__device__ __forceinline__ bool slab_exact(const float *lo, const float *hi, const LaneRay &r, float max_t) {
  float tmin = 0.0f, tmax = max_t;
  const float o[3] = {r.o.x, r.o.y, r.o.z}, inv[3] = {r.inv.x, r.inv.y, r.inv.z};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float t1 = (lo[d] - o[d]) * inv[d];
    float t2 = (hi[d] - o[d]) * inv[d];
    tmin = smin(smax(t1, tmin), smax(t2, tmin));
    tmax = smax(smin(t1, tmax), smin(t2, tmax));
  }
  return tmin <= tmax && tmax > r.min_t;
}

// One ray, one lane: BVHAccelOpt::intersect / has_intersection as the reference runs them (bvh.cuh:49-107).
template<bool ANY>
__device__ __forceinline__ void walk_reference_lane(const TraceArgs &a, uint32_t id) {
  LaneRay r;
  load_ray<ANY>(a, id, r);
  uint32_t stack[BVH2_STACK];
  uint32_t sp = 0;
  stack[sp++] = 0;
  bool found = false;
  while (sp) {
    const float4 *np = a.nodes2 + 2 * (size_t) stack[--sp];
    const float4 n0 = np[0], n1 = np[1];
    const float lo[3] = {n0.x, n0.y, n0.z}, hi[3] = {n0.w, n1.x, n1.y};
    const uint32_t w0 = f2u(n1.z), w1 = f2u(n1.w);
    if (!slab_exact(lo, hi, r, r.best_t)) continue;
    if (w0 & 0x80000000u) {
      const uint32_t cnt = w0 & 0x7fffffffu;
      for (uint32_t i = w1; i < w1 + cnt; i++) {
        float t, by, bz;
        uint32_t flat;
        if (tri_test(a.tris2, i, r, r.best_t, t, by, bz, flat)) {
          r.best_t = t; r.b1 = by; r.b2 = bz; r.best_idx = (int32_t) i;
          if (ANY) { found = true; break; }
        }
      }
      if (ANY && found) break;
    } else if (sp + 2 <= BVH2_STACK) {
      stack[sp++] = w0;
      stack[sp++] = w1;
    } else {
      atomicOr(a.error_flag, 1u);
    }
  }
  if (ANY) {
    a.occ[id] = found ? 1u : 0u;
    if (a.any_t) a.any_t[id] = r.best_t;
  } else {
    a.hit[id] = make_float4(r.best_t, r.b1, r.b2, u2f((uint32_t) r.best_idx));
  }
}

template<bool ANY>
__global__ void __launch_bounds__(256) k_trace2(TraceArgs a) {
  if (!a.redo) publish_counters(a);
  const uint32_t total = a.redo ? *a.redo_count : (a.qcount[a.qc_front] + (a.qc_back >= 0 ? a.qcount[a.qc_back] : 0));
  const uint32_t nA = a.redo ? total : a.qcount[a.qc_front];
  if (a.redo && a.redo_stat && total && blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd(a.redo_stat, (unsigned long long) total);
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < total; w += gridDim.x * blockDim.x) {
    uint32_t id = a.redo ? a.redo_queue[w] : (w < nA ? a.queue[w] : a.queue[a.qcap - 1 - (w - nA)]);
    if (id == QUEUE_SKIP) continue;
    walk_reference_lane<ANY>(a, id);
  }
}

// -------------------------------------------------------------------------------------------- k_redo (reference walk, one wave per ray)
// The rays k_trace4 sets aside are few (edge hits, coplanar faces, non-finite arithmetic: ~2e-5 of all rays) but each is a
// long, strictly ordered walk; per lane that is hundreds of dependent memory round trips.  Here a whole wave walks one
// ray: the node walk is uniform (every lane reads the same node), and the triangles of a leaf are evaluated by 64 lanes
// at once; the reference's sequential accept rule (primitives.cuh:149-152 inside bvh.cuh:62-72) is then replayed over
// the candidates in index order, which is all the order dependence there is.
template<bool ANY>
__global__ void __launch_bounds__(256) k_redo(TraceArgs a) {
  __shared__ uint32_t stack_lds[4][BVH2_STACK];
  const uint32_t total = *a.redo_count;
  const uint32_t lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
  const uint32_t wave = blockIdx.x * 4 + wave_in_block, n_waves = gridDim.x * 4;
  uint32_t *stack = stack_lds[wave_in_block];
  if (a.redo_stat && total && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.redo_stat, (unsigned long long) total);
  // Many rays (a scene lit by an axis-parallel directional light - the reference app's - sends EVERY shadow ray here: zero
  // direction components make the reference's slab test order dependent): throughput matters, one ray per lane.  A
  // handful (ties, rim hits): latency matters, a whole wave walks each ray (below).
  if (total > 8u * n_waves) {
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < total; w += gridDim.x * blockDim.x)
      walk_reference_lane<ANY>(a, a.redo_queue[w]);
    return;
  }
  for (uint32_t w = wave; w < total; w += n_waves) {
    const uint32_t id = a.redo_queue[w];
    LaneRay r;
    load_ray<ANY>(a, id, r);
    uint32_t sp = 0;
    stack[sp++] = 0;
    bool found = false;
    while (sp) {
      const float4 *np = a.nodes2 + 2 * (size_t) stack[--sp];
      const float4 n0 = np[0], n1 = np[1];
      const float lo[3] = {n0.x, n0.y, n0.z}, hi[3] = {n0.w, n1.x, n1.y};
      const uint32_t w0 = f2u(n1.z), w1 = f2u(n1.w);
      if (!slab_exact(lo, hi, r, r.best_t)) continue;
      if (w0 & 0x80000000u) {
        const uint32_t cnt = w0 & 0x7fffffffu;
        for (uint32_t base = 0; base < cnt && !found; base += 64) {
          float t = 0, by = 0, bz = 0;
          uint32_t flat;
          bool cand = false;
          if (base + lane < cnt) cand = tri_test(a.tris2, w1 + base + lane, r, MIRT_INF, t, by, bz, flat);   // all but `t > max_t`
          uint64_t m = __ballot(cand);
          int winner = -1;
          while (m) {
            const int l = (int) __builtin_ctzll(m);
            m &= m - 1;
            const float tl = __shfl(t, l);
            if (!(tl > r.best_t)) {
              r.best_t = tl;
              winner = l;
              if (ANY) { found = true; break; }
            }
          }
          if (winner >= 0) {
            r.b1 = __shfl(by, winner);
            r.b2 = __shfl(bz, winner);
            r.best_idx = (int32_t) (w1 + base + (uint32_t) winner);
          }
        }
        if (ANY && found) break;
      } else if (sp + 2 <= BVH2_STACK) {
        stack[sp++] = w0;
        stack[sp++] = w1;
      } else {
        atomicOr(a.error_flag, 1u);
      }
    }
    if (lane == 0) {
      if (ANY) {
        a.occ[id] = found ? 1u : 0u;
        if (a.any_t) a.any_t[id] = r.best_t;
      } else {
        a.hit[id] = make_float4(r.best_t, r.b1, r.b2, u2f((uint32_t) r.best_idx));
      }
    }
  }
}

// -------------------------------------------------------------------------------------------- k_trace4 (wide BVH)
// traversal 1: the reference's BVH2 folded to 4-wide nodes, children visited in the reference's DFS order, exact slab
// arithmetic.  Provably the reference's result for finite arithmetic; the parity mode.
template<bool ANY, bool COUNT>
__global__ void __launch_bounds__(TRACE_BLOCK) k_trace4(TraceArgs a) {
  extern __shared__ uint2 lds_stack[];                    // [LDS_STACK][TRACE_BLOCK]
  const uint32_t tid = threadIdx.x;
  const uint32_t gtid = blockIdx.x * TRACE_BLOCK + tid;
  const uint32_t nthreads = gridDim.x * TRACE_BLOCK;
  const uint32_t nA = a.qcount[a.qc_front];
  const uint32_t total = nA + (a.qc_back >= 0 ? a.qcount[a.qc_back] : 0);
  publish_counters(a);

  LaneRay r;
  r.id = 0; r.best_idx = -1; r.best_t = 0; r.min_t = 0; r.b1 = r.b2 = 0;
  r.o = r.d = r.inv = mk(0, 0, 0);
  bool active = false;
  uint32_t cur = CHILD_EMPTY;
  // wave-uniform fetch state
  const uint32_t psz = ((total + WORK_PARTS - 1) / WORK_PARTS + 63u) & ~63u;        // queue positions per range
  uint32_t chunk = (total / (gridDim.x * (TRACE_BLOCK / 64) * 4u)) & ~31u;          // ~4 chunks per wave
  chunk = chunk < 32u ? 32u : (chunk > 256u ? 256u : chunk);
  uint32_t w_next = 0, w_end = 0, part = blockIdx.x & (WORK_PARTS - 1), parts_tried = 0;
  bool wave_done = (total == 0);
  int sp = 0;
  uint32_t n_nodes = 0, n_tris = 0, n_node_steps = 0, n_tri_steps = 0;   // COUNT: work items and wave-level steps

  // (a ray whose stack outgrows the LDS part + overflow region - the region is capped, capi.hip - is handed to the reference
  // walk once the node step is over)
  bool too_deep = false;
  auto push = [&](uint32_t ref, float tmn) {
    if (sp < LDS_STACK) lds_stack[sp * TRACE_BLOCK + tid] = make_uint2(ref, f2u(tmn));
    else if (sp < LDS_STACK + (int) a.ovf_depth) a.ovf[(size_t) (sp - LDS_STACK) * nthreads + gtid] = make_uint2(ref, f2u(tmn));
    else { too_deep = true; return; }
    sp++;
  };
  auto pop = [&]() {                                       // next node whose entry distance is still in range
    cur = CHILD_EMPTY;
    while (sp > 0) {
      sp--;
      uint2 e = (sp < LDS_STACK) ? lds_stack[sp * TRACE_BLOCK + tid] : a.ovf[(size_t) (sp - LDS_STACK) * nthreads + gtid];
      if (ANY || u2f(e.y) <= r.best_t) { cur = e.x; break; }
    }
  };

  while (true) {
    // ---------------------------------------------------------------- dynamic fetch
    // Idle lanes take the next queue positions of the wave's private range [w_next, w_end); the wave refills that range
    // a chunk at a time from one of WORK_PARTS counters (its home range first, then the others): few atomics, spread
    // over several lines.
    const uint64_t need = __ballot(!active);
    if (need != 0 && !wave_done) {
      if (w_next >= w_end) {
        while (parts_tried < (uint32_t) WORK_PARTS) {
          // (peeking at the counter with an agent-scope load before the atomic was measured 20 % SLOWER overall)
          uint32_t got = 0;
          if ((tid & 63) == 0) got = atomicAdd(a.work_head + CTR_STRIDE * part, chunk);
          got = __builtin_amdgcn_readfirstlane(got);
          const uint32_t lo = part * psz + got;
          if (got < psz && lo < total) {
            const uint32_t hi_part = (part + 1) * psz < total ? (part + 1) * psz : total;
            w_next = lo;
            w_end = lo + chunk < hi_part ? lo + chunk : hi_part;
            break;
          }
          part = (part + 1) & (WORK_PARTS - 1);
          parts_tried++;
        }
        if (w_next >= w_end) wave_done = true;
      }
      if (!wave_done) {
        if (!active) {
          const uint32_t w = w_next + lane_rank(need);
          const uint32_t id = (w < w_end) ? ((w < nA) ? a.queue[w] : a.queue[a.qcap - 1 - (w - nA)]) : QUEUE_SKIP;
          if (id != QUEUE_SKIP) {                          // (QUEUE_SKIP in the queue: a culled shadow ray)
            float l2;
            load_ray<ANY>(a, id, r, &l2);
            // Slab and hit arithmetic stays finite for these rays; anything else (zero direction components -> 0 * inf,
            // non-finite origins, NaN bounds) makes the reference's result depend on its visiting order: defer to k_trace2.
            bool regular = finite3(r.o) && finite3(r.d) && finite3(r.inv) && (r.best_t == r.best_t) &&
                           (r.min_t == r.min_t);
            if (a.require_unit) {
              // The reference prunes boxes with t measured along the UN-normalised ctor direction but hits with t along
              // the normalised one (ray.cuh:19-21): unless the two agree (the integrator only makes unit directions) its
              // result depends on its own boxes and visiting order, so such rays take the reference walk (traversal 0
              // only: its tree is not the reference's).
              regular = regular && (fabsf(l2 - 1.0f) < 1e-4f);
            }
            if (regular) {
              active = true;
              cur = 0;
              sp = 0;
            } else {
              const uint32_t pos = wave_reserve(a.redo_count, 1);
              a.redo_queue[pos] = id;
            }
          }
        }
        const uint32_t taken = (uint32_t) __popcll(need);
        w_next = (w_end - w_next) < taken ? w_end : w_next + taken;
      }
    }
    if (!__any(active)) {
      if (wave_done) break;
      continue;
    }

    // ---------------------------------------------------------------- inner nodes
    while (active && cur != CHILD_EMPTY && !(cur & CHILD_LEAF)) {
      const float4 *np = a.nodes4 + 8 * (size_t) cur;
      if (COUNT) {
        n_nodes++;
        if (lane_rank(__ballot(1)) == 0) n_node_steps++;
      }
      const float4 LX = np[0], LY = np[1], LZ = np[2], HX = np[3], HY = np[4], HZ = np[5];
      const float4 CHf = np[6];
      const float lox[4] = {LX.x, LX.y, LX.z, LX.w}, loy[4] = {LY.x, LY.y, LY.z, LY.w}, loz[4] = {LZ.x, LZ.y, LZ.z, LZ.w};
      const float hix[4] = {HX.x, HX.y, HX.z, HX.w}, hiy[4] = {HY.x, HY.y, HY.z, HY.w}, hiz[4] = {HZ.x, HZ.y, HZ.z, HZ.w};
      uint32_t ch[4] = {f2u(CHf.x), f2u(CHf.y), f2u(CHf.z), f2u(CHf.w)};
      float key[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        // bbox.cuh:71-75 for finite operands: the running std::max/std::min chain equals these min/max trees
        const float t1x = (lox[c] - r.o.x) * r.inv.x, t2x = (hix[c] - r.o.x) * r.inv.x;
        const float t1y = (loy[c] - r.o.y) * r.inv.y, t2y = (hiy[c] - r.o.y) * r.inv.y;
        const float t1z = (loz[c] - r.o.z) * r.inv.z, t2z = (hiz[c] - r.o.z) * r.inv.z;
        const float tmn = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fmaxf(fminf(t1z, t2z), 0.0f));
        const float tmx = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fminf(fmaxf(t1z, t2z), r.best_t));
        // the reference's exact comparison (same boxes as the reference: same decisions)
        const bool hit = (ch[c] != CHILD_EMPTY) && (tmn <= tmx && tmx > r.min_t);
        key[c] = hit ? tmn : MIRT_INF;
        if (!hit) ch[c] = CHILD_EMPTY;
      }
      // descend into the first surviving child, push the others so that the earlier ones pop first
      uint32_t first = CHILD_EMPTY;
#pragma unroll
      for (int c = 3; c >= 0; c--) {
        if (ch[c] != CHILD_EMPTY) {
          bool earlier = false;
#pragma unroll
          for (int e = 0; e < c; e++) earlier = earlier || (ch[e] != CHILD_EMPTY);
          if (earlier) push(ch[c], key[c]);
          else first = ch[c];
        }
      }
      if (first != CHILD_EMPTY) cur = first;
      else pop();
      if (too_deep) {
        const uint32_t pos = wave_reserve(a.redo_count, 1);
        a.redo_queue[pos] = r.id;
        active = false;
        too_deep = false;
      }
    }

    // ---------------------------------------------------------------- leaf
    if (active && cur != CHILD_EMPTY && (cur & CHILD_LEAF)) {
      uint32_t start = cur & ((1u << LEAF_COUNT_SHIFT) - 1);
      uint32_t cnt = (cur >> LEAF_COUNT_SHIFT) & 15u;
      if (cnt == 0) {                                     // oversized reference leaf: (start, count) lives in a side table
        const uint2 bl = a.big[start];
        start = bl.x;
        cnt = bl.y;
      }
      bool found = false, irregular = false;
      for (uint32_t i = start; i < start + cnt; i++) {
        float t, by, bz;
        if (COUNT) {
          n_tris++;
          if (lane_rank(__ballot(1)) == 0) n_tri_steps++;
        }
        uint32_t flat;
        if (tri_test(a.tris, i, r, r.best_t, t, by, bz, flat)) {
          if (!(fabsf(t) < MIRT_INF)) { irregular = true; break; }        // accepted NaN / inf hit
          r.best_t = t; r.b1 = by; r.b2 = bz; r.best_idx = (int32_t) flat;
          if (ANY) { found = true; break; }
        }
      }
      if (irregular) {
        const uint32_t pos = wave_reserve(a.redo_count, 1);
        a.redo_queue[pos] = r.id;
        active = false;
      } else if (ANY && found) {
        a.occ[r.id] = 1u;
        if (a.any_t) a.any_t[r.id] = r.best_t;
        active = false;
      } else {
        pop();
      }
    }

    // ---------------------------------------------------------------- ray finished
    if (active && cur == CHILD_EMPTY) {
      if (ANY) {
        a.occ[r.id] = 0u;
        if (a.any_t) a.any_t[r.id] = r.best_t;
      } else {
        a.hit[r.id] = make_float4(r.best_t, r.b1, r.b2, u2f((uint32_t) r.best_idx));
      }
      active = false;
    }
  }
  if (COUNT) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      n_nodes += __shfl_xor(n_nodes, o);
      n_tris += __shfl_xor(n_tris, o);
      n_node_steps += __shfl_xor(n_node_steps, o);
      n_tri_steps += __shfl_xor(n_tri_steps, o);
    }
    if ((tid & 63) == 0) {
      atomicAdd(a.count_stat + 0, (unsigned long long) n_nodes);
      atomicAdd(a.count_stat + 1, (unsigned long long) n_tris);
      atomicAdd(a.count_stat + 4, (unsigned long long) n_node_steps);
      atomicAdd(a.count_stat + 5, (unsigned long long) n_tri_steps);
    }
  }
}

// -------------------------------------------------------------------------------------------- k_trace_fast (traversal 0)
// The ordered traversal of the library's own SAH tree.  Its result must equal the reference's, not its arithmetic: the
// nearest hit of a ray does not depend on the tree, the boxes or the visiting order unless two candidates (or a candidate
// and one of the ray's t limits) lie within NEAR of each other, and those rays are handed to the reference walk (k_redo).
// So the box test only has to be conservative, which buys:
//   * 96-byte nodes with 8 children whose boxes are 8-bit planes on a power-of-two grid anchored at the node's minimum
//     corner p:  t = (p + q * 2^e - o) * inv = q * (2^e * inv) + (p - o) * inv, one cvt + one fma per plane, the near /
//     far plane picked by the direction's sign; the rounding of (p - o) * inv is covered by widening each axis by
//     AXIS_EPS * |(p - o) * inv| and the comparison by SLACK;
//   * one loop in which every lane advances by one node OR one triangle per trip ("if-if"): the while-while form left
//     75 % of the lanes idle in the node loop on this tree (tools/count_frame.py).
// The primitive test itself is the exact one (tri_test_near).  SPH: the scene has sphere primitives.
#if MIRT_TRACE_WAVES
#define MIRT_TRACE_ATTR __attribute__((amdgpu_waves_per_eu(MIRT_TRACE_WAVES, MIRT_TRACE_WAVES)))
#else
#define MIRT_TRACE_ATTR
#endif
template<bool ANY, bool COUNT, bool SPH>
__global__ void __launch_bounds__(TRACE_BLOCK) MIRT_TRACE_ATTR k_trace_fast(TraceArgs a) {
  constexpr float NEAR = 1.0e-5f;
  // Scenes with spheres: the reference's sphere test (primitives.cuh:111-136) forms c = a.a - b*b with a = centre - origin,
  // which cancels catastrophically for a sphere far from the ray's origin.  With u = 2^-24: the three subtractions of a, the
  // products and sums of a.a (5u |a|^2), of b = a.d (4u |a|) and b*b (9u |a|^2) and the final subtraction (u |a|^2) put the
  // computed c within 15u |a|^2 of the true one - SPH_LATERAL^2 = (9.5e-4)^2 > 15u.  The hit the reference works with,
  // o + t d with t = b -+ sqrt(r^2 - c), then lies within sqrt(r^2 + 15u |a|^2) of the centre instead of r: up to 0.18 units
  // outside a sphere of radius 0.2 seen from the 350 units of the reference app's camera, beside or in front of the
  // sphere's bounding box.  A traversal that prunes with the true boxes would never test such a sphere while the
  // reference, whose leaf boxes are unions over other spheres, may.  So for these scenes every box is dilated by that
  // much (node step below); with the hit inside the dilated box, distance pruning needs no slack of its own.
  // tools/soak_spheres.py: constants of 2e-4 / 3e-4 / 4e-4 leave 35 / 4 / 0.25 differences per million queries.
  constexpr float SPH_LATERAL = MIRT_SPH_LATERAL;
  constexpr float SLACK = SPH ? 1.0f + MIRT_SLACK_SPH : 1.0f + MIRT_SLACK;
  constexpr float AXIS_EPS = 4.76837158e-7f;              // 2^-21
  constexpr float INV_LIMIT = 1.0e15f;                    // steeper than this against an axis: reference walk

  extern __shared__ uint2 lds_stack[];                    // [LDS_STACK][TRACE_BLOCK]
  const uint32_t tid = threadIdx.x;
  const uint32_t gtid = blockIdx.x * TRACE_BLOCK + tid;
  const uint32_t nthreads = gridDim.x * TRACE_BLOCK;
  const uint32_t nA = a.qcount[a.qc_front];
  const uint32_t total = nA + (a.qc_back >= 0 ? a.qcount[a.qc_back] : 0);
  publish_counters(a);

  LaneRay r;
  r.id = 0; r.best_idx = -1; r.best_t = 0; r.min_t = 0; r.b1 = r.b2 = 0;
  r.o = r.d = r.inv = mk(0, 0, 0);
  bool active = false;
  uint32_t cur = CHILD_EMPTY, tri_i = 0, tri_n = 0;
  // Work distribution.  Device-scope atomics on one address retire at ~11 ns each no matter who issues them, so a scheme
  // in which every one of the ~6000 waves touches a global counter even once costs ~70 us per launch.  Instead: the first
  // static_16th/16 of the queue is cut into one slice per workgroup, handed out to its waves through an LDS counter; the
  // rest is a pool of WORK_PARTS ranges with global counters that balances the load at the end, and a workgroup probes a
  // dry range only once (LDS bitmask).
  // A launch with few rays (the tail of a frame, a small image) uses only as many workgroups as it has rays for, one ray
  // per lane; the others leave at once instead of queueing up at the pool counters (measured: 125 us -> ~20 us for a
  // launch of 4 k rays).
  // With very few rays (the last iterations of a frame) the launch is pure latency: a wave takes as many trips as the
  // longest of its rays, and every trip runs the node part and the primitive part if any lane needs them.  Such a launch
  // is spread thin - 2, 4, ... 32 rays per wave, as few as the grid's waves allow, one batch per wave, no refill.
  __shared__ uint32_t s_next, s_dry;
  const uint32_t waves_all = gridDim.x * (TRACE_BLOCK / 64u);
  uint32_t thin_r = 0;
  if (MIRT_THIN && total <= waves_all * 32u) {
    thin_r = MIRT_THIN_MIN;
    while (thin_r * waves_all < total) thin_r *= 2u;
  }
  const uint32_t per_wg = (TRACE_BLOCK / 64u) * (thin_r ? thin_r : 64u);   // rays a workgroup starts with
  const uint32_t want_blk = (total + per_wg - 1u) / per_wg;
  const uint32_t n_blk = want_blk < gridDim.x ? (want_blk ? want_blk : 1u) : gridDim.x;
  if (blockIdx.x >= n_blk) return;
  const uint32_t per_blk = (uint32_t) (((uint64_t) total * a.static_16th / 16u) / n_blk) & ~63u;
  // Workgroups go to the 8 XCDs round-robin (blockIdx % 8), each XCD with an L2 of its own.  MIRT_XCD_SLICES=1: XCD x takes
  // the x-th eighth of the slices - neighbouring tiles of the image, rays that start close together - so that its L2 holds
  // that region's part of the tree instead of all of it.  Measured: nothing (121.6 vs 121.8 ms; the tree lives in the
  // Infinity Cache and its latency is hidden as well as an L2 hit's), so the plain order stays.  (The pool ranges are per
  // XCD either way: a workgroup starts at range blockIdx % 8.)
  const uint32_t vblk = (MIRT_XCD_SLICES && (n_blk & 7u) == 0u) ? (blockIdx.x & 7u) * (n_blk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const uint32_t slice_end = (vblk + 1) * per_blk;
  const uint32_t pool_lo = n_blk * per_blk, pool_total = total - pool_lo;
  if (tid == 0) {
    s_next = vblk * per_blk;
    s_dry = 0;
  }
  __syncthreads();
  const uint32_t psz = ((pool_total + WORK_PARTS - 1) / WORK_PARTS + 63u) & ~63u;     // pool positions per range
  uint32_t chunk = (pool_total / (n_blk * (TRACE_BLOCK / 64) * MIRT_CHUNK_DIV)) & ~31u;
  chunk = chunk < 32u ? 32u : (chunk > MIRT_CHUNK_MAX ? MIRT_CHUNK_MAX : chunk);
  uint32_t w_next = 0, w_end = 0, part = blockIdx.x & (WORK_PARTS - 1), parts_tried = 0;
  bool slice_done = (per_blk == 0);
  bool wave_done = (total == 0);
  if (thin_r) {                                            // one batch: positions [wave * thin_r, + thin_r), then done
    w_next = (gtid >> 6) * thin_r;
    w_end = w_next + thin_r < total ? w_next + thin_r : total;
    if (w_next >= total) { w_next = w_end = 0; wave_done = true; }
    slice_done = true;
    parts_tried = (uint32_t) WORK_PARTS;
  }
  int sp = 0;
  uint32_t n_nodes = 0, n_tris = 0, n_node_steps = 0, n_tri_steps = 0;

  auto enter = [&](uint32_t ref) {
    cur = ref;
    if (ref != CHILD_EMPTY && (ref & CHILD_LEAF)) {
      tri_i = ref & ((1u << LEAF_COUNT_SHIFT) - 1);
      tri_n = (ref >> LEAF_COUNT_SHIFT) & 15u;              // the SAH tree has no oversized leaves
    }
  };
  // (a stack deeper than LDS part + overflow region - the region is capped, capi.hip: the entry lands on the last slot, the
  // ray is flagged and handed to the reference walk at the end of the node step)
  // Occlusion rays never look at an entry's distance: their entries are 4 bytes, and twice as many fit the lane's LDS part.
  constexpr int NSTK = ANY ? 2 * LDS_STACK : LDS_STACK;
  uint32_t *lds_stack32 = (uint32_t *) lds_stack;
  bool too_deep = false;
  const int sp_max = NSTK + (int) a.ovf_depth - 1;
  auto push = [&](uint32_t ref, float tmn) {
    too_deep = too_deep || sp > sp_max;
    const int at = sp > sp_max ? sp_max : sp;
    if (at < NSTK) {
      if (ANY) lds_stack32[at * TRACE_BLOCK + tid] = ref;
      else lds_stack[at * TRACE_BLOCK + tid] = make_uint2(ref, f2u(tmn));
    } else {
      a.ovf[(size_t) (at - NSTK) * nthreads + gtid] = make_uint2(ref, f2u(tmn));
    }
    sp = at + 1;
  };
  auto pop = [&]() {
    uint32_t next = CHILD_EMPTY;
    while (sp > 0) {
      sp--;
      if (ANY) {
        next = (sp < NSTK) ? lds_stack32[sp * TRACE_BLOCK + tid] : a.ovf[(size_t) (sp - NSTK) * nthreads + gtid].x;
        break;
      }
      uint2 e = (sp < NSTK) ? lds_stack[sp * TRACE_BLOCK + tid] : a.ovf[(size_t) (sp - NSTK) * nthreads + gtid];
      if (u2f(e.y) <= r.best_t * SLACK) { next = e.x; break; }
    }
    enter(next);
  };

  while (true) {
    // ---------------------------------------------------------------- fetch
    // (refilling costs ~150 instructions for the whole wave - normalise, three correctly rounded reciprocals - so it waits
    // until a few lanes are idle)
    const uint64_t need = __ballot(!active);
    if (need != 0 && !wave_done && ((uint32_t) __popcll(need) >= a.refill_min || need == ~0ull)) {
      if (w_next >= w_end) {
        if (!slice_done) {                                 // 1. the workgroup's own slice: LDS counter, no global traffic
          uint32_t got = 0;
          if ((tid & 63) == 0) got = atomicAdd(&s_next, 64u);
          got = __builtin_amdgcn_readfirstlane(got);
          if (got < slice_end) {
            w_next = got;
            w_end = got + 64u < slice_end ? got + 64u : slice_end;
          } else {
            slice_done = true;
          }
        }
        if (w_next >= w_end) {                             // 2. the shared pool
          while (parts_tried < (uint32_t) WORK_PARTS) {
            if (!((*(volatile uint32_t *) &s_dry >> part) & 1u)) {
              uint32_t got = 0;
              if ((tid & 63) == 0) got = atomicAdd(a.work_head + CTR_STRIDE * part, chunk);
              got = __builtin_amdgcn_readfirstlane(got);
              const uint32_t lo = pool_lo + part * psz + got;
              if (got < psz && lo < total) {
                const uint32_t hi_part = pool_lo + (part + 1) * psz < total ? pool_lo + (part + 1) * psz : total;
                w_next = lo;
                w_end = lo + chunk < hi_part ? lo + chunk : hi_part;
                break;
              }
              if ((tid & 63) == 0) atomicOr(&s_dry, 1u << part);     // tell the other waves of the workgroup
            }
            part = (part + 1) & (WORK_PARTS - 1);
            parts_tried++;
          }
          if (w_next >= w_end) wave_done = true;
        }
      }
      if (!wave_done) {
        if (!active) {
          const uint32_t w = w_next + lane_rank(need);
          const uint32_t id = (w < w_end) ? ((w < nA) ? a.queue[w] : a.queue[a.qcap - 1 - (w - nA)]) : QUEUE_SKIP;
          if (id != QUEUE_SKIP) {                          // (QUEUE_SKIP in the queue: a culled shadow ray)
            float l2;
            load_ray<ANY>(a, id, r, &l2);
            // Rays the shortcut arithmetic is not meant for take the reference walk: non-finite origin / direction /
            // reciprocal (the reference's result then depends on its visiting order), a direction almost parallel to an
            // axis plane ((p - o) * inv would lose all precision), and ctor directions that are not unit length (the
            // reference prunes boxes along the un-normalised direction but measures hits along the normalised one,
            // ray.cuh:19-21; the integrator only makes unit directions).
            const bool regular = finite3(r.o) && finite3(r.d) && (r.best_t == r.best_t) && (r.min_t == r.min_t) &&
                                 fabsf(r.inv.x) < INV_LIMIT && fabsf(r.inv.y) < INV_LIMIT && fabsf(r.inv.z) < INV_LIMIT &&
                                 (fabsf(l2 - 1.0f) < 1e-4f);
            if (regular) {
              active = true;
              sp = 0;
              enter(0);
            } else {
              const uint32_t pos = wave_reserve(a.redo_count, 1);
              a.redo_queue[pos] = id;
            }
          }
        }
        const uint32_t taken = (uint32_t) __popcll(need);
        w_next = (w_end - w_next) < taken ? w_end : w_next + taken;
      }
    }
    if (!__any(active)) {
      if (wave_done) break;
      continue;
    }

    const bool at_node = active && !(cur & CHILD_LEAF);
    // ---------------------------------------------------------------- one step per lane: a node ...
    if (at_node) {
      if (COUNT) {
        n_nodes++;
        if (lane_rank(__ballot(1)) == 0) n_node_steps++;
      }
      const uint4 *nq = (const uint4 *) ((const char *) a.nodes8 + (size_t) cur * 96u);
      const uint4 H = nq[0], C0 = nq[1], C1 = nq[2], Q0 = nq[3], Q1 = nq[4], Q2 = nq[5];
      const float sx = u2f((H.w & 0xFFu) << 23) * r.inv.x, sy = u2f(((H.w >> 8) & 0xFFu) << 23) * r.inv.y,
                  sz = u2f(((H.w >> 16) & 0xFFu) << 23) * r.inv.z;
      const float bx = (u2f(H.x) - r.o.x) * r.inv.x, by_ = (u2f(H.y) - r.o.y) * r.inv.y, bz_ = (u2f(H.z) - r.o.z) * r.inv.z;
      float ex = AXIS_EPS * fabsf(bx), ey = AXIS_EPS * fabsf(by_), ez = AXIS_EPS * fabsf(bz_);
      if (SPH) {                                           // dilation by what the reference's sphere arithmetic can be off (see above)
        // |a| <= distance from the origin to the farthest corner of the node's box, for every sphere centre in the node
        const float wx = 255.0f * u2f((H.w & 0xFFu) << 23), wy = 255.0f * u2f(((H.w >> 8) & 0xFFu) << 23),
                    wz = 255.0f * u2f(((H.w >> 16) & 0xFFu) << 23);
        const float fx = fmaxf(fabsf(u2f(H.x) - r.o.x), fabsf(u2f(H.x) + wx - r.o.x)),
                    fy = fmaxf(fabsf(u2f(H.y) - r.o.y), fabsf(u2f(H.y) + wy - r.o.y)),
                    fz = fmaxf(fabsf(u2f(H.z) - r.o.z), fabsf(u2f(H.z) + wz - r.o.z));
        const float reach2 = (fx * fx + fy * fy + fz * fz) * 1.00001f;
        // sd2 bounds the error of c; a computed hit then lies within sqrt(r^2 + sd2) of the sphere's centre, so the box of a
        // sphere of radius r has to grow by sqrt(r^2 + sd2) - r = sd2 / (sqrt(r^2 + sd2) + r) on every side: most for the
        // smallest sphere of the scene (r = 0: by sqrt(sd2))
        const float sd2 = (SPH_LATERAL * SPH_LATERAL) * reach2, rm = a.sph_rmin;
        const float m = sd2 * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(rm * rm + sd2) + rm + 1.0e-30f) * 1.0001f;
        ex += m * fabsf(r.inv.x); ey += m * fabsf(r.inv.y); ez += m * fabsf(r.inv.z);
      }
      const float cnx = bx - ex, cny = by_ - ey, cnz = bz_ - ez, cfx = bx + ex, cfy = by_ + ey, cfz = bz_ + ez;
      const bool ngx = r.inv.x < 0, ngy = r.inv.y < 0, ngz = r.inv.z < 0;
      // Q0 = qlo_x[8] qlo_y[8], Q1 = qlo_z[8] qhi_x[8], Q2 = qhi_y[8] qhi_z[8]; near plane = lo unless the ray runs backwards
      const uint32_t nxw[2] = {ngx ? Q1.z : Q0.x, ngx ? Q1.w : Q0.y}, fxw[2] = {ngx ? Q0.x : Q1.z, ngx ? Q0.y : Q1.w};
      const uint32_t nyw[2] = {ngy ? Q2.x : Q0.z, ngy ? Q2.y : Q0.w}, fyw[2] = {ngy ? Q0.z : Q2.x, ngy ? Q0.w : Q2.y};
      const uint32_t nzw[2] = {ngz ? Q2.z : Q1.x, ngz ? Q2.w : Q1.y}, fzw[2] = {ngz ? Q1.x : Q2.z, ngz ? Q1.y : Q2.w};
      const uint32_t ch[8] = {C0.x, C0.y, C0.z, C0.w, C1.x, C1.y, C1.z, C1.w};
      const float far_limit = r.best_t;
      // All eight slab tests first (straight-line VALU), then the choice of the child to enter, then the pushes: the common
      // case - the lane's pushes all fit its LDS part of the stack - is one compare, one masked ds_write and one add per
      // child (the first form spent most of its instructions on exec-mask bookkeeping around eight conditional pushes with
      // their LDS / overflow split).  Where the kernel stands now (profiles/r3_pmc_sq.json): 58 wave
      // instructions per ray at 42 % of the lanes active; running alone its SIMDs are VALU-busy 68 % of the time and its
      // waves parked on s_waitcnt 47 % of theirs; inside the overlapped frame resident waves are what it competes for.
      float key[8];
      uint32_t hits = 0;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const int w = c >> 2, sh = 8 * (c & 3);
        const float qnx = (float) ((nxw[w] >> sh) & 0xFFu), qny = (float) ((nyw[w] >> sh) & 0xFFu), qnz = (float) ((nzw[w] >> sh) & 0xFFu);
        const float qfx = (float) ((fxw[w] >> sh) & 0xFFu), qfy = (float) ((fyw[w] >> sh) & 0xFFu), qfz = (float) ((fzw[w] >> sh) & 0xFFu);
        const float tnx = fmaf(qnx, sx, cnx), tny = fmaf(qny, sy, cny), tnz = fmaf(qnz, sz, cnz);
        const float tfx = fmaf(qfx, sx, cfx), tfy = fmaf(qfy, sy, cfy), tfz = fmaf(qfz, sz, cfz);
        const float tmn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
        const float tmx = fminf(fminf(tfx, tfy), fminf(tfz, far_limit)) * SLACK;
        const bool hit = (ch[c] != CHILD_EMPTY) && (tmn <= tmx) && (tmx > r.min_t);
        key[c] = tmn;
        hits |= hit ? (1u << c) : 0u;
      }
      // the child entered next: the nearest one hit (closest-hit rays), the first one hit (occlusion rays: any order will do)
      uint32_t first = CHILD_EMPTY;
      float first_key = MIRT_INF;
      uint32_t first_bit = 0;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const bool better = ((hits >> c) & 1u) && (ANY ? (first_bit == 0u) : (first_bit == 0u || key[c] < first_key));
        first = better ? ch[c] : first;
        first_key = better ? key[c] : first_key;
        first_bit = better ? (1u << c) : first_bit;
      }
      uint32_t rest = hits & ~first_bit;                   // the others wait on the stack with their entry distance
      if (MIRT_FAR_FIRST && !ANY && (rest & (rest - 1u))) {      // two or more: the farthest goes deepest
        uint32_t far_c = 0;
        float far_key = -1.0f;
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const bool f = ((rest >> c) & 1u) && key[c] > far_key;
          far_key = f ? key[c] : far_key;
          far_c = f ? (uint32_t) c : far_c;
        }
        uint32_t far_ref = ch[0];
#pragma unroll
        for (int c = 1; c < 8; c++) far_ref = far_c == (uint32_t) c ? ch[c] : far_ref;
        push(far_ref, far_key);
        rest &= ~(1u << far_c);
      }
      if (rest) {
        if (sp + (int) __popc(rest) <= NSTK) {
#pragma unroll
          for (int c = 0; c < 8; c++)
            if ((rest >> c) & 1u) {
              if (ANY) lds_stack32[sp * TRACE_BLOCK + tid] = ch[c];
              else lds_stack[sp * TRACE_BLOCK + tid] = make_uint2(ch[c], f2u(key[c]));
              sp++;
            }
        } else {
#pragma unroll
          for (int c = 0; c < 8; c++)
            if ((rest >> c) & 1u) push(ch[c], key[c]);
        }
      }
      if (first != CHILD_EMPTY) enter(first);
      else pop();
      if (too_deep) {
        const uint32_t pos = wave_reserve(a.redo_count, 1);
        a.redo_queue[pos] = r.id;
        active = false;
        too_deep = false;
      }
    }

    // ---------------------------------------------------------------- ... or up to LEAF_BATCH primitives of a leaf (lanes that
    // just reached a leaf join in).  The records of a leaf are requested together - one memory round trip for the leaf
    // instead of one per primitive - and tested in the leaf's order against the shrinking best_t, as one per trip would.
    constexpr uint32_t LEAF_BATCH = ANY ? MIRT_LEAF_BATCH_ANY : MIRT_LEAF_BATCH;
    if (active && cur != CHILD_EMPTY && (cur & CHILD_LEAF)) {
      const float4 *tp = a.tris + 3 * (size_t) tri_i;
      float4 tq[LEAF_BATCH][3];
      const uint32_t n_batch = tri_n < LEAF_BATCH ? tri_n : LEAF_BATCH;
#pragma unroll
      for (uint32_t k = 0; k < LEAF_BATCH; k++) {
        tq[k][0] = tq[k][1] = tq[k][2] = make_float4(0, 0, 0, 0);
        if (k < n_batch) { tq[k][0] = tp[3 * k]; tq[k][1] = tp[3 * k + 1]; tq[k][2] = tp[3 * k + 2]; }
      }
      if (COUNT && lane_rank(__ballot(1)) == 0) n_tri_steps++;
#pragma unroll
      for (uint32_t k = 0; k < LEAF_BATCH; k++) {
        if (active && k < n_batch) {
          float t, by, bz;
          uint32_t flat;
          bool near = false;
          if (COUNT) n_tris++;
          const bool acc = tri_test_near<SPH>(tq[k][0], tq[k][1], tq[k][2], a.verts, r, r.best_t, NEAR, t, by, bz, flat, near);
          if (near || (acc && !(fabsf(t) < MIRT_INF))) {
            // two candidates (or a candidate and a limit) too close to call, or an accepted NaN / inf hit
            const uint32_t pos = wave_reserve(a.redo_count, 1);
            a.redo_queue[pos] = r.id;
            active = false;
          } else if (acc) {
            r.best_t = t; r.b1 = by; r.b2 = bz; r.best_idx = (int32_t) flat;
            if (ANY) {
              a.occ[r.id] = 1u;
              if (a.any_t) a.any_t[r.id] = r.best_t;
              active = false;
            }
          }
        }
      }
      tri_i += n_batch;
      tri_n -= n_batch;
      if (active && tri_n == 0) pop();
    }

    // ---------------------------------------------------------------- ray finished
    if (active && cur == CHILD_EMPTY) {
      if (ANY) {
        a.occ[r.id] = 0u;
        if (a.any_t) a.any_t[r.id] = r.best_t;
      } else {
        a.hit[r.id] = make_float4(r.best_t, r.b1, r.b2, u2f((uint32_t) r.best_idx));
      }
      active = false;
    }
  }
  if (COUNT) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      n_nodes += __shfl_xor(n_nodes, o);
      n_tris += __shfl_xor(n_tris, o);
      n_node_steps += __shfl_xor(n_node_steps, o);
      n_tri_steps += __shfl_xor(n_tri_steps, o);
    }
    if ((tid & 63) == 0) {
      atomicAdd(a.count_stat + 0, (unsigned long long) n_nodes);
      atomicAdd(a.count_stat + 1, (unsigned long long) n_tris);
      atomicAdd(a.count_stat + 4, (unsigned long long) n_node_steps);
      atomicAdd(a.count_stat + 5, (unsigned long long) n_tri_steps);
    }
  }
}

// ============================================================================================ probes
// Sample-0 camera rays with the reference seed procedure, for mirt_primary_hits.
__global__ void k_probe_primary(RenderArgs a) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n_slots) return;
  uint32_t px, py;
  bool valid;
  slot_to_pixel(a, s, px, py, valid);
  if (!valid) return;
  PathCtx c;
  c.rng.mode = a.rng_mode;
  c.rng.state = a.st[s].x;
  c.rng.key = (a.rng_mode == 1) ? ctr_key(py * a.width + px, a.seed) : 0;
  c.rng.sample = 0;
  c.i = 1; c.flags = 0; c.depth = 0;
  c.sample_base = a.sample_begin;
  Emit em;
  em.ext = 0; em.nsh = 0;
  generate_camera_ray(a, s, px, py, c, 1, em);
  const uint32_t pos = wave_reserve(a.qcount + QC_EXT_FRONT, 1);
  a.ext_queue[pos] = s;
}

__global__ void k_gather_primary(RenderArgs a, int32_t *idx_out, uint32_t *tbits_out) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n_slots) return;
  uint32_t px, py;
  bool valid;
  slot_to_pixel(a, s, px, py, valid);
  if (!valid) return;
  const float4 h = a.hit[s];
  idx_out[(size_t) py * a.width + px] = (int32_t) f2u(h.w);
  tbits_out[(size_t) py * a.width + px] = f2u(h.x);
}

__global__ void k_fill4(float4 *p, uint32_t n, float4 v) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_iota(uint32_t *q, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) q[i] = i;
}

// mirt_trace_static: the caller's ray records (n x 8 floats, device memory) into the traversal kernels' layout, plus the
// identity queue and its counter (the control block was cleared before).
__global__ void k_stage_rays(const float *rays8, uint32_t n, float4 *ray_o, float4 *ray_d, uint32_t *queue, uint32_t *qcount_slot) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *qcount_slot = n;
  if (i >= n) return;
  const float4 a = ((const float4 *) rays8)[2 * (size_t) i], b = ((const float4 *) rays8)[2 * (size_t) i + 1];
  ray_o[i] = make_float4(a.x, a.y, a.z, b.z);           // origin | min_t
  ray_d[i] = make_float4(a.w, b.x, b.y, b.w);           // ctor direction | max_t
  queue[i] = i;
}

__global__ void k_unstage_hits(uint32_t n, int any, const float4 *hit, const uint32_t *occ, const float *any_t, int32_t *idx_out,
                               uint32_t *tbits_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (any) {
    idx_out[i] = occ[i] ? 1 : -1;
    tbits_out[i] = f2u(any_t[i]);
  } else {
    const float4 h = hit[i];
    idx_out[i] = (int32_t) f2u(h.w);
    tbits_out[i] = f2u(h.x);
  }
}

// Sample streams: the per-stream partial sums (sum rgb, sample count) are added in stream order, so the result does not
// depend on scheduling.
__global__ void k_sum_planes(const float4 *planes, uint32_t n_planes, uint32_t n, float4 *out, int add_to_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 acc = planes[i];
  for (uint32_t k = 1; k < n_planes; k++) {
    const float4 v = planes[(size_t) k * n + i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (add_to_out) {                                       // progressive pass: the buffer already holds earlier passes
    const float4 v = out[i];
    acc.x = v.x + acc.x; acc.y = v.y + acc.y; acc.z = v.z + acc.z; acc.w = v.w + acc.w;
  }
  out[i] = acc;
}

// ============================================================================================ film
// One packed word per pixel (r | g << 8 | b << 16): the film's values are 0..255, and a third of the bytes cross PCIe;
// the host widens them to the reference's three ints per pixel (image.cuh:10-13).
__global__ void k_tonemap(const float4 *accum, uint32_t n, const float *table, uint32_t *rgb) {
  __shared__ float T[256];
  T[threadIdx.x & 255] = table[threadIdx.x & 255];
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = accum[i];
  const float rc = 1.0f / v.w;                    // color /= i - 1  (raytracer.cuh:124; Vector3f.cuh:133-135)
  const float c3[3] = {v.x * rc, v.y * rc, v.z * rc};
  uint32_t packed = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float c = c3[k];
    int out;
    if (c != c || c < 0.f) out = 255;             // powf of NaN / negative -> NaN -> min(1, NaN) = 1 (image.cuh:44)
    else {
      int lo = 0, hi = 256;
#pragma unroll
      for (int it = 0; it < 8; it++) {
        const int mid = (lo + hi) >> 1;
        if (c >= T[mid]) lo = mid; else hi = mid;
      }
      out = lo;
    }
    packed |= (uint32_t) out << (8 * k);
  }
  rgb[i] = packed;
}

// ============================================================================================ launchers
#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t launch_init(const RenderArgs &a, hipStream_t st) {
  k_init<<<(a.n_slots + 255) / 256, 256, 0, st>>>(a);
  LAUNCH_CHECK();
  return hipSuccess;
}
hipError_t launch_logic(const RenderArgs &a, bool deferred_microfacet, hipStream_t st) {
  if (!deferred_microfacet) {
    if (a.first) k_logic<false, true><<<(a.n_slots + 255) / 256, 256, 0, st>>>(a);
    else k_logic<false><<<(a.n_slots + 255) / 256, 256, 0, st>>>(a);
  } else {
    const uint32_t blocks = (a.n_slots + 255) / 256;
    k_logic<true><<<blocks < 2048 ? blocks : 2048, 256, 0, st>>>(a);
  }
  LAUNCH_CHECK();
  return hipSuccess;
}
hipError_t launch_probe_primary(const RenderArgs &a, hipStream_t st) {
  k_probe_primary<<<(a.n_slots + 255) / 256, 256, 0, st>>>(a);
  LAUNCH_CHECK();
  return hipSuccess;
}
hipError_t launch_gather_primary(const RenderArgs &a, int32_t *idx, uint32_t *tb, hipStream_t st) {
  k_gather_primary<<<(a.n_slots + 255) / 256, 256, 0, st>>>(a, idx, tb);
  LAUNCH_CHECK();
  return hipSuccess;
}
hipError_t launch_fill4(float4 *p, uint32_t n, float4 v, hipStream_t st) {
  if (n == 0) return hipSuccess;
  k_fill4<<<(n + 255) / 256, 256, 0, st>>>(p, n, v);
  LAUNCH_CHECK();
  return hipSuccess;
}
hipError_t launch_iota(uint32_t *q, uint32_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  k_iota<<<(n + 255) / 256, 256, 0, st>>>(q, n);
  LAUNCH_CHECK();
  return hipSuccess;
}
hipError_t launch_stage_rays(const float *rays8, uint32_t n, float4 *ray_o, float4 *ray_d, uint32_t *queue, uint32_t *qcount_slot,
                             hipStream_t st) {
  k_stage_rays<<<(n + 255) / 256, 256, 0, st>>>(rays8, n, ray_o, ray_d, queue, qcount_slot);
  LAUNCH_CHECK();
  return hipSuccess;
}
hipError_t launch_unstage_hits(uint32_t n, int any, const float4 *hit, const uint32_t *occ, const float *any_t, int32_t *idx_out,
                               uint32_t *tbits_out, hipStream_t st) {
  k_unstage_hits<<<(n + 255) / 256, 256, 0, st>>>(n, any, hit, occ, any_t, idx_out, tbits_out);
  LAUNCH_CHECK();
  return hipSuccess;
}
hipError_t launch_sum_planes(const float4 *planes, uint32_t n_planes, uint32_t n, float4 *out, int add_to_out, hipStream_t st) {
  k_sum_planes<<<(n + 255) / 256, 256, 0, st>>>(planes, n_planes, n, out, add_to_out);
  LAUNCH_CHECK();
  return hipSuccess;
}
hipError_t launch_tonemap(const void *accum, uint32_t n, const float *table, uint32_t *rgb, hipStream_t st) {
  k_tonemap<<<(n + 255) / 256, 256, 0, st>>>((const float4 *) accum, n, table, rgb);
  LAUNCH_CHECK();
  return hipSuccess;
}

size_t trace4_lds_bytes() { return (size_t) LDS_STACK * TRACE_BLOCK * sizeof(uint2); }

int trace4_blocks_per_cu() {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_fast<false, false, false>, TRACE_BLOCK, trace4_lds_bytes()) != hipSuccess || nb < 1)
    nb = 2;
  return nb;
}

// traversal: 0 ordered wide, 1 DFS-order wide, 2 reference BVH2.  `a.redo` selects the deferred-ray pass of k_trace2.
hipError_t launch_trace(const TraceArgs &a, bool any, uint32_t traversal, uint32_t grid_blocks, hipStream_t st) {
  if (a.redo) {
    if (any) k_redo<true><<<grid_blocks, 256, 0, st>>>(a);
    else k_redo<false><<<grid_blocks, 256, 0, st>>>(a);
  } else if (traversal == 2) {
    if (any) k_trace2<true><<<grid_blocks, 256, 0, st>>>(a);
    else k_trace2<false><<<grid_blocks, 256, 0, st>>>(a);
  } else {
    const size_t lds = trace4_lds_bytes();
    if (traversal == 0) {
#define MIRT_FAST(ANYV, CNTV, SPHV) k_trace_fast<ANYV, CNTV, SPHV><<<grid_blocks, TRACE_BLOCK, lds, st>>>(a)
      if (any) {
        if (a.spheres) { if (a.count) MIRT_FAST(true, true, true); else MIRT_FAST(true, false, true); }
        else { if (a.count) MIRT_FAST(true, true, false); else MIRT_FAST(true, false, false); }
      } else {
        if (a.spheres) { if (a.count) MIRT_FAST(false, true, true); else MIRT_FAST(false, false, true); }
        else { if (a.count) MIRT_FAST(false, true, false); else MIRT_FAST(false, false, false); }
      }
#undef MIRT_FAST
    } else if (any) {
      if (a.count) k_trace4<true, true><<<grid_blocks, TRACE_BLOCK, lds, st>>>(a);
      else k_trace4<true, false><<<grid_blocks, TRACE_BLOCK, lds, st>>>(a);
    } else {
      if (a.count) k_trace4<false, true><<<grid_blocks, TRACE_BLOCK, lds, st>>>(a);
      else k_trace4<false, false><<<grid_blocks, TRACE_BLOCK, lds, st>>>(a);
    }
  }
  LAUNCH_CHECK();
  return hipSuccess;
}

}  // namespace mirt
