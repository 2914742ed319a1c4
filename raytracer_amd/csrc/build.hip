// Device build of the ordered traversal's own tree (the reference builds on the host only,
// bvh/bvh.cuh:177-290).  The ordered traversal accepts ANY tree whose boxes contain their primitives (DESIGN.md 4:
// exact primitive tests, near-ties replayed on the reference's tree), so this builder is free in its topology:
//
//   boxes + centroids -> 63-bit Morton keys -> radix sort (rocPRIM) -> binary radix tree (Karras 2012: one thread per
//   inner node finds its key range and split) -> boxes bottom-up (the second thread to arrive at a node merges its
//   children) -> fold to 8-wide quantised nodes level by level (largest-area child opened first; a subtree of at most
//   `max_leaf` primitives, contiguous in sorted order, becomes a leaf) -> primitive records gathered into tree order.
//
// The 96-byte node and its conservative 8-bit planes are exactly host_bvh.cpp's (Collapser8); the kernels here redo that
// arithmetic in double, including the containment checks.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "device_build.hpp"
#include "host_bvh.hpp"

namespace mirt {
namespace {

constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr int BLOCK = 256;

__device__ __forceinline__ uint32_t f2o(float f) {          // float -> unsigned with the same order
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(uint32_t o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

struct Boxes {
  float4 *lo, *hi;                                          // per primitive, flat order
};

// scene[0..2] = min of the centroids, scene[3..5] = max (ordered-uint form); flags bit 0: a non-finite coordinate
__global__ void __launch_bounds__(BLOCK) k_prim_boxes(const float *__restrict__ pos, const uint8_t *__restrict__ ptype,
                                                      uint32_t n, Boxes bx, uint32_t *scene, uint32_t *flags) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
  bool bad = false;
  if (i < n) {
    const float *p = pos + 9 * (size_t) i;
    if (ptype && ptype[i] == 1) {                           // sphere: one ulp of room (as SahBuilder::prepare)
      for (int k = 0; k < 3; k++) {
        lo[k] = nextafterf(p[k] - p[3], -INFINITY);
        hi[k] = nextafterf(p[k] + p[3], INFINITY);
      }
    } else {
      for (int v = 0; v < 3; v++)
        for (int k = 0; k < 3; k++) {
          lo[k] = fminf(lo[k], p[3 * v + k]);
          hi[k] = fmaxf(hi[k], p[3 * v + k]);
          bad = bad || !isfinite(p[3 * v + k]);
        }
    }
    for (int k = 0; k < 3; k++) {
      bad = bad || !isfinite(lo[k]) || !isfinite(hi[k]);
      const float c = 0.5f * (lo[k] + hi[k]);
      bad = bad || !isfinite(c);
      cmin[k] = c;
      cmax[k] = c;
    }
    bx.lo[i] = make_float4(lo[0], lo[1], lo[2], 0.f);
    bx.hi[i] = make_float4(hi[0], hi[1], hi[2], 0.f);
  }
  // workgroup reduction of the centroid bounds: one atomic per component and workgroup
  __shared__ float red[6][BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < 3; k++) {
    float a = cmin[k], b = cmax[k];
    for (int o = 32; o > 0; o >>= 1) {
      a = fminf(a, __shfl_xor(a, o));
      b = fmaxf(b, __shfl_xor(b, o));
    }
    if (lane == 0) {
      red[k][wave] = a;
      red[3 + k][wave] = b;
    }
  }
  if (__any(bad) && lane == 0) atomicOr(flags, 1u);
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    float v = red[k][0];
    for (int w = 1; w < BLOCK / 64; w++) v = (k < 3) ? fminf(v, red[k][w]) : fmaxf(v, red[k][w]);
    if (isfinite(v)) {
      if (k < 3) atomicMin(&scene[k], f2o(v));
      else atomicMax(&scene[k], f2o(v));
    }
  }
}

__device__ __forceinline__ uint64_t spread3(uint32_t a) {   // 21 bits -> every third bit
  uint64_t x = a & 0x1FFFFFu;
  x = (x | x << 32) & 0x001F00000000FFFFull;
  x = (x | x << 16) & 0x001F0000FF0000FFull;
  x = (x | x << 8) & 0x100F00F00F00F00Full;
  x = (x | x << 4) & 0x10C30C30C30C30C3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

__global__ void __launch_bounds__(BLOCK) k_morton(Boxes bx, uint32_t n, const uint32_t *__restrict__ scene,
                                                  uint64_t *keys, uint32_t *vals) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const float4 lo = bx.lo[i], hi = bx.hi[i];
  const float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
  uint32_t q[3];
  for (int k = 0; k < 3; k++) {
    const float mn = o2f(scene[k]), mx = o2f(scene[3 + k]);
    const double ext = (double) mx - (double) mn;
    double t = ext > 0 ? ((double) c[k] - (double) mn) / ext * 2097152.0 : 0.0;
    t = t < 0 ? 0 : (t > 2097151.0 ? 2097151.0 : t);
    q[k] = (uint32_t) t;
  }
  keys[i] = spread3(q[0]) << 2 | spread3(q[1]) << 1 | spread3(q[2]);
  vals[i] = i;
}

// ---- binary radix tree over the sorted keys (Karras, "Maximizing parallelism in the construction of BVHs, octrees and
// k-d trees", HPG 2012, section 4).  Inner node i, leaves 0..n-1; child references carry LEAF_BIT for leaves.
struct Radix {
  const uint64_t *keys;
  uint32_t n;
  uint2 *child;                                             // per inner node: (left, right)
  uint2 *range;                                             // per inner node: (first, last) sorted positions
  uint32_t *parent_inner, *parent_leaf;
};

__device__ __forceinline__ int delta(const Radix &r, int i, int j) {
  if (j < 0 || j >= (int) r.n) return -1;
  const uint64_t a = r.keys[i], b = r.keys[j];
  if (a == b) return 64 + __clz((uint32_t) i ^ (uint32_t) j);
  return __clzll((long long) (a ^ b));
}

__global__ void __launch_bounds__(BLOCK) k_radix_tree(Radix r) {
  const int i = (int) (blockIdx.x * BLOCK + threadIdx.x);
  if (i >= (int) r.n - 1) return;
  const int d = (delta(r, i, i + 1) - delta(r, i, i - 1)) < 0 ? -1 : 1;
  const int dmin = delta(r, i, i - d);
  int lmax = 2;
  while (delta(r, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(r, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(r, i, j);
  int s = 0, t = l;
  do {
    t = (t + 1) >> 1;
    if (delta(r, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const int gamma = i + s * d + (d < 0 ? d : 0);
  const int first = i < j ? i : j, last = i < j ? j : i;
  const uint32_t left = (first == gamma) ? (LEAF_BIT | (uint32_t) gamma) : (uint32_t) gamma;
  const uint32_t right = (last == gamma + 1) ? (LEAF_BIT | (uint32_t) (gamma + 1)) : (uint32_t) (gamma + 1);
  r.child[i] = make_uint2(left, right);
  r.range[i] = make_uint2((uint32_t) first, (uint32_t) last);
  if (left & LEAF_BIT) r.parent_leaf[gamma] = (uint32_t) i;
  else r.parent_inner[gamma] = (uint32_t) i;
  if (right & LEAF_BIT) r.parent_leaf[gamma + 1] = (uint32_t) i;
  else r.parent_inner[gamma + 1] = (uint32_t) i;
  if (i == 0) r.parent_inner[0] = 0xFFFFFFFFu;
}

// ---- boxes bottom-up.  One thread per leaf climbs; at every inner node the first arrival stops, the second (which by
// then can see both children's boxes: acquire/release on the arrival counter) merges them and climbs on.
struct Refit {
  uint32_t n;
  const uint32_t *sorted;                                   // sorted position -> flat primitive
  Boxes prim;
  const uint2 *child;
  const uint32_t *parent_inner, *parent_leaf;
  uint32_t *arrived;
  float4 *nlo, *nhi;                                        // per inner node
};

__device__ __forceinline__ void ref_box(const Refit &r, uint32_t ref, float4 &lo, float4 &hi) {
  if (ref & LEAF_BIT) {
    const uint32_t p = r.sorted[ref & ~LEAF_BIT];
    lo = r.prim.lo[p];
    hi = r.prim.hi[p];
  } else {
    lo = r.nlo[ref];
    hi = r.nhi[ref];
  }
}

__global__ void __launch_bounds__(BLOCK) k_refit(Refit r) {
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= r.n || r.n < 2) return;
  uint32_t cur = r.parent_leaf[j];
  while (cur != 0xFFFFFFFFu) {
    const uint32_t before = __hip_atomic_fetch_add(&r.arrived[cur], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (before == 0) return;
    const uint2 c = r.child[cur];
    float4 alo, ahi, blo, bhi;
    ref_box(r, c.x, alo, ahi);
    ref_box(r, c.y, blo, bhi);
    r.nlo[cur] = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), 0.f);
    r.nhi[cur] = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), 0.f);
    cur = r.parent_inner[cur];
  }
}

// ---- PLOC (Meister & Bittner, "Parallel locally-ordered clustering for bounding volume hierarchy construction", TVCG
// 2018): the clusters, kept in Morton order, each look for the neighbour within `radius` positions whose union with
// them has the smallest area; mutual nearest neighbours merge; the list is compacted (order kept) and the search repeats
// until one cluster is left.  Bottom-up with a surface-area criterion: about the traversal cost of a binned SAH tree.
constexpr int PLOC_RADIUS_MAX = 64;

struct Ploc {
  uint32_t m;
  int radius;
  const uint32_t *cl_in;                                    // cluster = node reference (LEAF_BIT | sorted position, or inner id)
  const float4 *lo_in, *hi_in;                              // its box; lo.w carries the primitive count (bits)
  uint32_t *nn;
  uint32_t *cl_out, *valid;
  float4 *lo_out, *hi_out;
  uint2 *child;
  float4 *nlo, *nhi;
  uint32_t *count;
  uint32_t *parent_inner, *parent_leaf;
  uint32_t *node_counter;
};

__device__ __forceinline__ float union_area(const float4 &alo, const float4 &ahi, const float4 &blo, const float4 &bhi) {
  const float dx = fmaxf(ahi.x, bhi.x) - fminf(alo.x, blo.x), dy = fmaxf(ahi.y, bhi.y) - fminf(alo.y, blo.y),
              dz = fmaxf(ahi.z, bhi.z) - fminf(alo.z, blo.z);
  return dx * dy + dy * dz + dz * dx;
}

__global__ void __launch_bounds__(BLOCK) k_ploc_nn(Ploc p) {
  __shared__ float4 slo[BLOCK + 2 * PLOC_RADIUS_MAX], shi[BLOCK + 2 * PLOC_RADIUS_MAX];
  const int PLOC_RADIUS = p.radius;
  const int base = (int) (blockIdx.x * BLOCK) - PLOC_RADIUS;
  for (int t = threadIdx.x; t < BLOCK + 2 * PLOC_RADIUS; t += BLOCK) {
    const int g = base + t;
    if (g >= 0 && g < (int) p.m) {
      slo[t] = p.lo_in[g];
      shi[t] = p.hi_in[g];
    }
  }
  __syncthreads();
  const int i = (int) (blockIdx.x * BLOCK + threadIdx.x);
  if (i >= (int) p.m) return;
  const float4 mlo = slo[threadIdx.x + PLOC_RADIUS], mhi = shi[threadIdx.x + PLOC_RADIUS];
  float best_a = INFINITY;
  int best = -1;
  // Among equal areas the pair partner i ^ 1 wins, then the lowest index.  Either way the closest pair of a pass is mutual
  // (so every pass merges something), and a run of identical boxes pairs up (2k, 2k + 1) instead of one pair per pass.
  for (int o = -PLOC_RADIUS; o <= PLOC_RADIUS; o++) {
    const int g = i + o;
    if (o == 0 || g < 0 || g >= (int) p.m) continue;
    const float a = union_area(mlo, mhi, slo[threadIdx.x + PLOC_RADIUS + o], shi[threadIdx.x + PLOC_RADIUS + o]);
    if (best < 0 || a < best_a || (a == best_a && g == (i ^ 1))) {
      best_a = a;
      best = g;
    }
  }
  p.nn[i] = (uint32_t) best;
}

__global__ void __launch_bounds__(BLOCK) k_ploc_merge(Ploc p) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  const bool in = i < p.m;
  uint32_t j = 0;
  bool mutual = false;
  if (in) {
    j = p.nn[i];
    mutual = p.nn[j] == i;
  }
  const bool leader = in && mutual && i < j;
  // node ids: one atomic per wave
  const unsigned long long mask = __ballot(leader);
  uint32_t id = 0;
  if (mask) {
    const int lane = threadIdx.x & 63;
    const int first = __ffsll((long long) mask) - 1;
    uint32_t wave_base = 0;
    if (lane == first) wave_base = atomicAdd(p.node_counter, (uint32_t) __popcll(mask));
    wave_base = __shfl(wave_base, first);
    id = wave_base + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
  }
  if (!in) return;
  if (leader) {
    const uint32_t a = p.cl_in[i], b = p.cl_in[j];
    const float4 alo = p.lo_in[i], ahi = p.hi_in[i], blo = p.lo_in[j], bhi = p.hi_in[j];
    const uint32_t cnt = __float_as_uint(alo.w) + __float_as_uint(blo.w);
    const float4 lo = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), __uint_as_float(cnt));
    const float4 hi = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), 0.f);
    p.child[id] = make_uint2(a, b);
    p.nlo[id] = lo;
    p.nhi[id] = hi;
    p.count[id] = cnt;
    if (a & LEAF_BIT) p.parent_leaf[a & ~LEAF_BIT] = id; else p.parent_inner[a] = id;
    if (b & LEAF_BIT) p.parent_leaf[b & ~LEAF_BIT] = id; else p.parent_inner[b] = id;
    p.cl_out[i] = id;
    p.lo_out[i] = lo;
    p.hi_out[i] = hi;
    p.valid[i] = 1;
  } else if (mutual) {
    p.valid[i] = 0;
  } else {
    p.cl_out[i] = p.cl_in[i];
    p.lo_out[i] = p.lo_in[i];
    p.hi_out[i] = p.hi_in[i];
    p.valid[i] = 1;
  }
}

__global__ void __launch_bounds__(BLOCK) k_ploc_compact(uint32_t m, const uint32_t *valid, const uint32_t *pos,
                                                        const uint32_t *cl, const float4 *lo, const float4 *hi,
                                                        uint32_t *cl_next, float4 *lo_next, float4 *hi_next, uint32_t *m_next) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= m) return;
  if (valid[i]) {
    const uint32_t q = pos[i];
    cl_next[q] = cl[i];
    lo_next[q] = lo[i];
    hi_next[q] = hi[i];
  }
  if (i == m - 1) *m_next = pos[i] + valid[i];
}

__global__ void __launch_bounds__(BLOCK) k_ploc_init(uint32_t n, const uint32_t *sorted, Boxes prim, uint32_t *cl, float4 *lo,
                                                     float4 *hi) {
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= n) return;
  const uint32_t q = sorted[j];
  float4 a = prim.lo[q];
  a.w = __uint_as_float(1u);
  cl[j] = LEAF_BIT | j;
  lo[j] = a;
  hi[j] = prim.hi[q];
}

// Depth-first position of every leaf / first position of every inner node: climbing to the root, add the size of the left
// sibling wherever the path comes up from a right child.
struct Order {
  uint32_t n;
  const uint2 *child;
  const uint32_t *count, *parent_inner, *parent_leaf;
};
__device__ __forceinline__ uint32_t climb(const Order &o, uint32_t me, uint32_t cur) {
  uint32_t pos = 0;
  while (cur != 0xFFFFFFFFu) {
    const uint2 c = o.child[cur];
    if (c.y == me) pos += (c.x & LEAF_BIT) ? 1u : o.count[c.x];
    me = cur;
    cur = o.parent_inner[cur];
  }
  return pos;
}
__global__ void __launch_bounds__(BLOCK) k_ploc_leaf_order(Order o, const uint32_t *sorted, uint32_t *leaf_pos, uint32_t *sorted_out) {
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= o.n) return;
  const uint32_t pos = climb(o, LEAF_BIT | j, o.parent_leaf[j]);
  leaf_pos[j] = pos;
  sorted_out[pos] = sorted[j];
}
__global__ void __launch_bounds__(BLOCK) k_ploc_ranges(Order o, const uint32_t *leaf_pos, uint2 *child_rw, uint2 *range) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= o.n - 1) return;
  const uint32_t first = climb(o, i, o.parent_inner[i]);
  range[i] = make_uint2(first, first + o.count[i] - 1);
}
__global__ void __launch_bounds__(BLOCK) k_ploc_relabel(uint32_t n, const uint32_t *leaf_pos, uint2 *child) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n - 1) return;
  uint2 c = child[i];
  if (c.x & LEAF_BIT) c.x = LEAF_BIT | leaf_pos[c.x & ~LEAF_BIT];
  if (c.y & LEAF_BIT) c.y = LEAF_BIT | leaf_pos[c.y & ~LEAF_BIT];
  child[i] = c;
}

// ---- fold to 8-wide quantised nodes, one level per launch.  Entry e of the level's queue is the binary-tree reference
// that becomes wide node base + e; inner children are appended to the next level's queue, whose position is their index.
struct Fold {
  uint32_t n, max_leaf;
  float leaf_ct;                                            // what dividing a would-be leaf costs, in primitive tests (MIRT_SAH_LEAF_CT)
  const uint32_t *sorted;
  Boxes prim;
  const uint2 *child, *range;
  const float4 *nlo, *nhi;
  const uint32_t *queue;
  uint32_t count, base;
  uint32_t *next_queue, *next_count;                        // next_count[0]: entries, next_count[1]: error flags
  uint32_t next_base;
  uint32_t *nodes;                                          // 24 words per wide node
};

__device__ __forceinline__ void fold_box(const Fold &f, uint32_t ref, float lo[3], float hi[3]) {
  float4 a, b;
  if (ref & LEAF_BIT) {
    const uint32_t p = f.sorted[ref & ~LEAF_BIT];
    a = f.prim.lo[p];
    b = f.prim.hi[p];
  } else {
    a = f.nlo[ref];
    b = f.nhi[ref];
  }
  lo[0] = a.x; lo[1] = a.y; lo[2] = a.z;
  hi[0] = b.x; hi[1] = b.y; hi[2] = b.z;
}
__device__ __forceinline__ uint32_t fold_count(const Fold &f, uint32_t ref) {
  if (ref & LEAF_BIT) return 1;
  const uint2 r = f.range[ref];
  return r.y - r.x + 1;
}
__device__ __forceinline__ uint32_t fold_first(const Fold &f, uint32_t ref) {
  return (ref & LEAF_BIT) ? (ref & ~LEAF_BIT) : f.range[ref].x;
}

__device__ __forceinline__ float fold_area(const Fold &f, uint32_t ref) {
  float lo[3], hi[3];
  fold_box(f, ref, lo, hi);
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return dx * dy + dy * dz + dz * dx;
}
// A subtree of at most max_leaf primitives becomes a leaf unless dividing it where the binary tree does still pays: the same
// rule as the host builder's (host_bvh.cpp, SahBuilder::build), which searches for the division; here it is the tree's own.
__device__ __forceinline__ bool fold_is_leaf(const Fold &f, uint32_t ref) {
  const uint32_t cnt = fold_count(f, ref);
  if (cnt > f.max_leaf) return false;
  if (cnt < 2 || !(f.leaf_ct < 1.0e9f)) return true;
  const float a = fold_area(f, ref);
  if (!(a > 0.f)) return true;
  const uint2 c = f.child[ref];
  const float divided = fold_area(f, c.x) * (float) fold_count(f, c.x) + fold_area(f, c.y) * (float) fold_count(f, c.y);
  return !(f.leaf_ct * a + divided < (float) cnt * a);
}

__global__ void __launch_bounds__(64) k_fold_level(Fold f) {
  const uint32_t e = blockIdx.x * 64 + threadIdx.x;
  if (e >= f.count) return;
  const uint32_t root = f.queue[e];
  uint32_t kids[8];
  float area[8];                                            // < 0: a leaf (not to be opened)
  int nk = 0;
  auto add = [&](int k, uint32_t ref) {
    kids[k] = ref;
    area[k] = fold_is_leaf(f, ref) ? -1.f : fold_area(f, ref);
  };
  if (fold_is_leaf(f, root)) {
    add(nk++, root);                                        // a scene that is one leaf
  } else {
    const uint2 c = f.child[root];
    add(nk++, c.x);
    add(nk++, c.y);
    while (nk < 8) {
      int best = -1;
      float best_a = -1.f;
      for (int k = 0; k < nk; k++)
        if (area[k] >= 0.f && !(area[k] <= best_a)) { best_a = area[k]; best = k; }
      if (best < 0) break;
      const uint2 c2 = f.child[kids[best]];
      add(best, c2.x);
      add(nk++, c2.y);
    }
  }
  float rlo[3], rhi[3];
  fold_box(f, root, rlo, rhi);
  uint32_t words[24];
  for (int k = 0; k < 24; k++) words[k] = 0;
  uint32_t err = 0;
  int eb[3];
  uint8_t qlo[3][8], qhi[3][8];
  for (int a = 0; a < 3; a++)
    for (int k = 0; k < 8; k++) qlo[a][k] = qhi[a][k] = 0;
  for (int a = 0; a < 3; a++) {
    words[a] = __float_as_uint(rlo[a]);
    const double p = (double) rlo[a];
    const double ext = (double) rhi[a] - p;
    int ex = -126;
    if (ext > 0) {
      int fe;
      (void) frexp(ext / 255.0, &fe);                       // ext / 255 = m * 2^fe, m in [0.5, 1)  =>  2^fe >= ext / 255
      ex = fe;
    }
    if (ex < -126) ex = -126;
    if (ex > 127) { err |= 2; ex = 127; }
    eb[a] = ex + 127;
    double step = ldexp(1.0, ex);
    for (int tries = 0; tries < 4; tries++) {               // (a second pass only when rounding pushes a plane past 255)
      bool fits = true;
      for (int k = 0; k < nk && fits; k++) {
        float clo[3], chi[3];
        fold_box(f, kids[k], clo, chi);
        const double cmin = (double) clo[a], cmax = (double) chi[a];
        double lo = floor((cmin - p) / step), hi = ceil((cmax - p) / step);
        if (lo < 0) lo = 0;
        while (lo > 0 && p + lo * step > cmin) lo -= 1;
        while (p + hi * step < cmax && hi <= 255) hi += 1;
        if (hi > 255 || p + lo * step > cmin) { fits = false; break; }
        qlo[a][k] = (uint8_t) lo;
        qhi[a][k] = (uint8_t) hi;
      }
      if (fits) break;
      if (tries == 3 || ++eb[a] > 254) { err |= 2; break; }
      step *= 2.0;
    }
  }
  words[3] = (uint32_t) eb[0] | ((uint32_t) eb[1] << 8) | ((uint32_t) eb[2] << 16);
  for (int k = 0; k < 8; k++) words[4 + k] = CHILD_EMPTY;
  for (int k = 0; k < nk; k++) {
    const uint32_t cnt = fold_count(f, kids[k]);
    if (fold_is_leaf(f, kids[k])) {
      words[4 + k] = CHILD_LEAF | (cnt << LEAF_COUNT_SHIFT) | fold_first(f, kids[k]);
    } else {
      const uint32_t pos = atomicAdd(&f.next_count[0], 1u);
      f.next_queue[pos] = kids[k];
      words[4 + k] = f.next_base + pos;
    }
  }
  for (int a = 0; a < 3; a++)
    for (int k = 0; k < 8; k++) {
      words[12 + 2 * a + (k >> 2)] |= (uint32_t) qlo[a][k] << (8 * (k & 3));
      words[18 + 2 * a + (k >> 2)] |= (uint32_t) qhi[a][k] << (8 * (k & 3));
    }
  uint4 *out = (uint4 *) (f.nodes + (size_t) (f.base + e) * 24);
  for (int k = 0; k < 6; k++) out[k] = make_uint4(words[4 * k], words[4 * k + 1], words[4 * k + 2], words[4 * k + 3]);
  if (err) atomicOr(&f.next_count[1], err);
}

__global__ void __launch_bounds__(BLOCK) k_gather_records(const uint4 *__restrict__ tris, const uint32_t *__restrict__ sorted,
                                                          uint32_t n, uint4 *__restrict__ out) {
  const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
  if (t >= 3 * n) return;
  const uint32_t j = t / 3, q = t % 3;
  out[t] = tris[(size_t) sorted[j] * 3 + q];
}

struct Scratch {                                            // frees what the build allocated and does not hand over
  std::vector<void *> ptrs;
  ~Scratch() { for (void *p: ptrs) (void) hipFree(p); }
  template<class T> hipError_t get(T **out, size_t count) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, count * sizeof(T) + 16);
    if (e == hipSuccess) ptrs.push_back(p);
    *out = (T *) p;
    return e;
  }
};

}  // namespace

// (an early return must not leave copies in flight that target this function's stack variables, or kernels that use the
// scratch the destructor of `sc` frees: the failure path drains the stream first)
#define BCHK(expr)                                                                              \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      res.error = std::string(#expr) + ": " + hipGetErrorString(e_);                            \
      (void) hipStreamSynchronize(stream);                                                      \
      return -1;                                                                                \
    }                                                                                           \
  } while (0)

int build_fast_tree_device(const float *pos_flat_host, const uint8_t *ptype_flat_host, const void *tris_dev, uint32_t n,
                           uint32_t max_leaf, int method, hipStream_t stream, DeviceTree &res) {
  res = DeviceTree();
  if (n < 2) { res.error = "fewer than two primitives"; return 1; }
  if (max_leaf < 1 || max_leaf > LEAF_MAX) max_leaf = 3;
  Scratch sc;
  float *d_pos = nullptr;
  uint8_t *d_type = nullptr;
  Boxes bx{};
  uint32_t *d_scene = nullptr;                               // 6 bounds + flags + pad
  uint64_t *keys_a = nullptr, *keys_b = nullptr;
  uint32_t *vals_a = nullptr, *vals_b = nullptr;
  BCHK(sc.get(&d_pos, (size_t) n * 9));
  if (ptype_flat_host) BCHK(sc.get(&d_type, n));
  BCHK(sc.get(&bx.lo, n));
  BCHK(sc.get(&bx.hi, n));
  BCHK(sc.get(&d_scene, 8));
  BCHK(sc.get(&keys_a, n));
  BCHK(sc.get(&keys_b, n));
  BCHK(sc.get(&vals_a, n));
  BCHK(sc.get(&vals_b, n));
  hipEvent_t ev[5];
  for (auto &e: ev) BCHK(hipEventCreate(&e));
  struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 5; i++) (void) hipEventDestroy(e[i]); } } guard{ev};

  BCHK(hipMemcpyAsync(d_pos, pos_flat_host, (size_t) n * 9 * sizeof(float), hipMemcpyHostToDevice, stream));
  if (d_type) BCHK(hipMemcpyAsync(d_type, ptype_flat_host, n, hipMemcpyHostToDevice, stream));
  const uint32_t init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
  BCHK(hipMemcpyAsync(d_scene, init, sizeof init, hipMemcpyHostToDevice, stream));
  BCHK(hipEventRecord(ev[0], stream));
  const uint32_t nb = (n + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL(k_prim_boxes, dim3(nb), dim3(BLOCK), 0, stream, d_pos, d_type, n, bx, d_scene, d_scene + 6);
  hipLaunchKernelGGL(k_morton, dim3(nb), dim3(BLOCK), 0, stream, bx, n, d_scene, keys_a, vals_a);
  BCHK(hipGetLastError());
  size_t tmp_bytes = 0;
  BCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_a, keys_b, vals_a, vals_b, n, 0, 63, stream));
  char *d_tmp = nullptr;
  BCHK(sc.get(&d_tmp, tmp_bytes));
  BCHK(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, keys_a, keys_b, vals_a, vals_b, n, 0, 63, stream));
  BCHK(hipEventRecord(ev[1], stream));
  uint32_t h_flags = 0;
  BCHK(hipMemcpyAsync(&h_flags, d_scene + 6, 4, hipMemcpyDeviceToHost, stream));
  BCHK(hipStreamSynchronize(stream));
  if (h_flags & 1) { res.error = "non-finite vertex: no quantised tree"; return 1; }

  // binary tree over the sorted primitives: child / range / boxes per inner node, `sorted` = leaf position -> flat primitive
  uint2 *t_child = nullptr, *t_range = nullptr;
  float4 *t_nlo = nullptr, *t_nhi = nullptr;
  uint32_t *t_parent_inner = nullptr, *t_parent_leaf = nullptr;
  const uint32_t *t_sorted = vals_b;
  uint32_t root_ref = 0;
  BCHK(sc.get(&t_child, n));
  BCHK(sc.get(&t_range, n));
  BCHK(sc.get(&t_parent_inner, n));
  BCHK(sc.get(&t_parent_leaf, n));
  BCHK(sc.get(&t_nlo, n));
  BCHK(sc.get(&t_nhi, n));
  bool ploc = method == 2;
  if (ploc) {
    Ploc pl{};
    pl.radius = 32;                                          // measured: tools/ploc_sweep.sh
    if (const char *e = getenv("MIRT_PLOC_RADIUS")) {
      int v = atoi(e);
      if (v >= 1 && v <= PLOC_RADIUS_MAX) pl.radius = v;
    }
    uint32_t *cl_x = nullptr, *cl_y = nullptr, *d_pos_scan = nullptr, *d_m = nullptr, *leaf_pos = nullptr, *sorted2 = nullptr;
    float4 *lo_x = nullptr, *hi_x = nullptr, *lo_y = nullptr, *hi_y = nullptr;
    BCHK(sc.get(&cl_x, n));
    BCHK(sc.get(&cl_y, n));
    BCHK(sc.get(&lo_x, n));
    BCHK(sc.get(&hi_x, n));
    BCHK(sc.get(&lo_y, n));
    BCHK(sc.get(&hi_y, n));
    BCHK(sc.get(&pl.nn, n));
    BCHK(sc.get(&pl.valid, n));
    BCHK(sc.get(&pl.count, n));
    BCHK(sc.get(&d_pos_scan, n));
    BCHK(sc.get(&d_m, 2));
    BCHK(sc.get(&leaf_pos, n));
    BCHK(sc.get(&sorted2, n));
    pl.child = t_child;
    pl.nlo = t_nlo;
    pl.nhi = t_nhi;
    pl.parent_inner = t_parent_inner;
    pl.parent_leaf = t_parent_leaf;
    pl.node_counter = d_m + 1;
    BCHK(hipMemsetAsync(d_m, 0, 8, stream));
    BCHK(hipMemsetAsync(t_parent_inner, 0xFF, (size_t) n * 4, stream));
    hipLaunchKernelGGL(k_ploc_init, dim3(nb), dim3(BLOCK), 0, stream, n, vals_b, bx, cl_x, lo_x, hi_x);
    size_t scan_bytes = 0;
    BCHK(rocprim::exclusive_scan(nullptr, scan_bytes, pl.valid, d_pos_scan, 0u, n, rocprim::plus<uint32_t>(), stream));
    char *d_scan_tmp = nullptr;
    BCHK(sc.get(&d_scan_tmp, scan_bytes));
    uint32_t m = n, passes = 0;
    while (m > 1) {
      if (++passes > 1024) { ploc = false; break; }          // degenerate input: the Morton radix tree below takes over
      const uint32_t mb = (m + BLOCK - 1) / BLOCK;
      pl.m = m;
      pl.cl_in = cl_x;
      pl.lo_in = lo_x;
      pl.hi_in = hi_x;
      pl.cl_out = cl_y;
      pl.lo_out = lo_y;
      pl.hi_out = hi_y;
      hipLaunchKernelGGL(k_ploc_nn, dim3(mb), dim3(BLOCK), 0, stream, pl);
      hipLaunchKernelGGL(k_ploc_merge, dim3(mb), dim3(BLOCK), 0, stream, pl);
      BCHK(hipGetLastError());
      size_t sb = scan_bytes;
      BCHK(rocprim::exclusive_scan(d_scan_tmp, sb, pl.valid, d_pos_scan, 0u, m, rocprim::plus<uint32_t>(), stream));
      hipLaunchKernelGGL(k_ploc_compact, dim3(mb), dim3(BLOCK), 0, stream, m, pl.valid, d_pos_scan, cl_y, lo_y, hi_y, cl_x, lo_x,
                         hi_x, d_m);
      uint32_t m_next = 0;
      BCHK(hipMemcpyAsync(&m_next, d_m, 4, hipMemcpyDeviceToHost, stream));
      BCHK(hipStreamSynchronize(stream));
      if (m_next == 0 || m_next >= m) { res.error = "PLOC made no progress"; (void) hipStreamSynchronize(stream); return -1; }
      m = m_next;
    }
    res.passes = passes;
    if (ploc) {
    BCHK(hipMemcpyAsync(&root_ref, cl_x, 4, hipMemcpyDeviceToHost, stream));
    Order od{n, t_child, pl.count, t_parent_inner, t_parent_leaf};
    hipLaunchKernelGGL(k_ploc_leaf_order, dim3(nb), dim3(BLOCK), 0, stream, od, vals_b, leaf_pos, sorted2);
    hipLaunchKernelGGL(k_ploc_ranges, dim3(nb), dim3(BLOCK), 0, stream, od, leaf_pos, t_child, t_range);
    hipLaunchKernelGGL(k_ploc_relabel, dim3(nb), dim3(BLOCK), 0, stream, n, leaf_pos, t_child);
    BCHK(hipGetLastError());
    t_sorted = sorted2;
    }
  }
  if (!ploc) {
    Radix rt{};
    rt.keys = keys_b;
    rt.n = n;
    rt.child = t_child;
    rt.range = t_range;
    rt.parent_inner = t_parent_inner;
    rt.parent_leaf = t_parent_leaf;
    hipLaunchKernelGGL(k_radix_tree, dim3(nb), dim3(BLOCK), 0, stream, rt);
    Refit rf{};
    rf.n = n;
    rf.sorted = vals_b;
    rf.prim = bx;
    rf.child = t_child;
    rf.parent_inner = t_parent_inner;
    rf.parent_leaf = t_parent_leaf;
    rf.nlo = t_nlo;
    rf.nhi = t_nhi;
    BCHK(sc.get(&rf.arrived, n));
    BCHK(hipMemsetAsync(rf.arrived, 0, (size_t) n * 4, stream));
    hipLaunchKernelGGL(k_refit, dim3(nb), dim3(BLOCK), 0, stream, rf);
    BCHK(hipGetLastError());
  }
  BCHK(hipEventRecord(ev[2], stream));

  // fold, level by level; at most n - 1 wide nodes (every one but a lone root has two or more children)
  uint32_t *d_nodes = nullptr, *queue_a = nullptr, *queue_b = nullptr, *d_next = nullptr;
  BCHK(sc.get(&d_nodes, (size_t) n * 24));
  BCHK(sc.get(&queue_a, n));
  BCHK(sc.get(&queue_b, n));
  BCHK(sc.get(&d_next, 2));
  BCHK(hipStreamSynchronize(stream));                        // (root_ref of the PLOC path has arrived)
  BCHK(hipMemcpyAsync(queue_a, &root_ref, 4, hipMemcpyHostToDevice, stream));
  BCHK(hipStreamSynchronize(stream));
  Fold fd{};
  fd.n = n;
  fd.max_leaf = max_leaf;
  fd.leaf_ct = 0.5f;
  if (const char *e = std::getenv("MIRT_SAH_LEAF_CT")) fd.leaf_ct = (float) std::atof(e);
  fd.sorted = t_sorted;
  fd.prim = bx;
  fd.child = t_child;
  fd.range = t_range;
  fd.nlo = t_nlo;
  fd.nhi = t_nhi;
  fd.nodes = d_nodes;
  fd.next_count = d_next;
  uint32_t count = 1, base = 0, depth = 0;
  uint32_t *q_in = queue_a, *q_out = queue_b;
  while (count) {
    if (base + count > n) { res.error = "wide-node pool exhausted"; return -1; }
    BCHK(hipMemsetAsync(d_next, 0, 8, stream));
    fd.queue = q_in;
    fd.count = count;
    fd.base = base;
    fd.next_queue = q_out;
    fd.next_base = base + count;
    hipLaunchKernelGGL(k_fold_level, dim3((count + 63) / 64), dim3(64), 0, stream, fd);
    BCHK(hipGetLastError());
    uint32_t h_next[2] = {0, 0};
    BCHK(hipMemcpyAsync(h_next, d_next, 8, hipMemcpyDeviceToHost, stream));
    BCHK(hipStreamSynchronize(stream));
    if (h_next[1]) { res.error = "scene extent too large for the quantised tree"; return 1; }
    base += count;
    count = h_next[0];
    std::swap(q_in, q_out);
    if (count) depth++;
    if (depth > 200) { res.error = "own tree deeper than 200 wide levels"; return 1; }
  }
  BCHK(hipEventRecord(ev[3], stream));
  res.n_nodes = base;
  res.depth8 = depth;
  res.nodes_bytes = (size_t) base * 96;
  res.tris_bytes = (size_t) n * 48;
  void *out_nodes = nullptr, *out_tris = nullptr;
  BCHK(hipMalloc(&out_nodes, res.nodes_bytes));
  if (hipMalloc(&out_tris, res.tris_bytes) != hipSuccess) {
    (void) hipFree(out_nodes);
    res.error = "out of device memory";
    return -1;
  }
  hipError_t e1 = hipMemcpyAsync(out_nodes, d_nodes, res.nodes_bytes, hipMemcpyDeviceToDevice, stream);
  hipLaunchKernelGGL(k_gather_records, dim3((3 * n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, (const uint4 *) tris_dev,
                     t_sorted, n, (uint4 *) out_tris);
  hipError_t e2 = hipGetLastError();
  hipError_t e3 = hipEventRecord(ev[4], stream);
  hipError_t e4 = hipStreamSynchronize(stream);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
    (void) hipFree(out_nodes);
    (void) hipFree(out_tris);
    res.error = std::string("device build: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e3 != hipSuccess ? e3 : e4);
    return -1;
  }
  res.nodes8q = out_nodes;
  res.trisf = out_tris;
  for (int k = 0; k < 4; k++) (void) hipEventElapsedTime(&res.ms[k], ev[k], ev[k + 1]);
  return 0;
}

}  // namespace mirt
