// Host-side acceleration-structure build for libmirt (product code; independent of oracle/).
//
// Two stages:
//  1. build_reference_bvh2(): reproduces, bit for bit, the BVH the reference builds on the host
//     (bvh/bvh.cuh:187-225 construct_bvh, :272-290 flatten): mean-centroid split on the longest axis,
//     libstdc++'s bidirectional std::partition, pre-order flatten with the RIGHT child first.  Its flat
//     primitive order is what "hit index" means in every parity test, so it is not negotiable.
//  2. collapse_to_wide(): folds that binary tree into 4-wide, 128-byte, SoA nodes for the HIP kernels,
//     keeping the children of every wide node in the reference's DFS order so the `traversal = 1` mode can
//     visit leaves exactly as the reference does.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

namespace mirt {

// Threads the host side may use (OMP_NUM_THREADS is the GPU box's CPU share; MIRT_BUILD_THREADS overrides), and a plain
// blocked loop over [0, n) on them.  Results never depend on the thread count.
unsigned host_threads();
void parallel_for(size_t n, const std::function<void(size_t, size_t)> &body, size_t min_chunk = 8192);

// Host threads that outlive a call, for work that recurs every frame (the film's widening, capi.hip mirt_tonemap): starting
// and joining 16 std::threads costs ~0.4 ms, as much as the work they are started for.  run() hands jobs 0 .. n_jobs - 1 to the
// workers and the calling thread and returns when all are done.  One run() at a time per pool.
class WorkerPool {
 public:
  explicit WorkerPool(unsigned workers);
  ~WorkerPool();
  WorkerPool(const WorkerPool &) = delete;
  WorkerPool &operator=(const WorkerPool &) = delete;
  void run(size_t n_jobs, const std::function<void(size_t)> &job);
  unsigned workers() const;

 private:
  struct Impl;
  Impl *impl;
};

struct Node2 {            // = BVHNodeOpt (bvh/bvh.cuh:12-30), 32 bytes
  float minp[3];
  float maxp[3];
  uint32_t w0;            // inner: idx_left   | leaf: count | 0x80000000
  uint32_t w1;            // inner: idx_right  | leaf: idx_start
};
static_assert(sizeof(Node2) == 32, "Node2 layout");

constexpr uint32_t WIDE = 4;
constexpr uint32_t CHILD_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t CHILD_LEAF = 0x80000000u;
constexpr uint32_t LEAF_MAX = 15;            // primitives per inline leaf reference; larger leaves: count field 0, low bits index HostBVH::big
constexpr uint32_t LEAF_COUNT_SHIFT = 27;    // ref = LEAF | count << 27 | start  (start < 2^27)

struct alignas(128) Node4 {                  // 128 bytes, one or two cache lines, SoA over the four children
  float lox[4], loy[4], loz[4];
  float hix[4], hiy[4], hiz[4];
  uint32_t child[4];                         // CHILD_EMPTY | inner node index | leaf reference
  uint32_t pad[4];
};
static_assert(sizeof(Node4) == 128, "Node4 layout");

struct HostBVH {
  std::vector<Node2> nodes2;
  std::vector<uint32_t> flat_to_orig;        // flat slot -> caller's triangle index
  uint32_t depth2 = 0;
  std::vector<Node4> nodes4;
  std::vector<uint32_t> big;                 // (start, count) pairs of reference leaves with more than LEAF_MAX primitives
  uint32_t depth4 = 0;
  // the ordered traversal's own tree: binned SAH folded to 8-wide nodes with 8-bit quantised child boxes (96 bytes, Node8Q)
  std::vector<uint32_t> fast_to_flat;        // that tree's primitive order -> flat (reference-order) index
  std::vector<uint32_t> nodes8q;             // 24 words per node
  uint32_t depth8 = 0;
};

// 8-wide node of the ordered traversal, 96 bytes = 6 x dwordx4.  Child k's box is, per axis,
//   [ p + qlo[k] * 2^(e - 127),  p + qhi[k] * 2^(e - 127) ]   (qlo rounded down, qhi rounded up: it contains the exact box)
// The ordered traversal only needs conservative boxes, so 8 bits per plane are enough.
struct Node8Q {
  float p[3];                                // minimum corner of the node's own box
  uint32_t exps;                             // biased float exponents of the x, y, z grid steps: ex | ey << 8 | ez << 16
  uint32_t child[8];                         // CHILD_EMPTY | inner node index | leaf reference
  uint8_t qlo_x[8], qlo_y[8], qlo_z[8];      // byte k = child k
  uint8_t qhi_x[8], qhi_y[8], qhi_z[8];
};
static_assert(sizeof(Node8Q) == 96, "Node8Q layout");

// prim_pos: n x 9 floats (triangle: p1 p2 p3; sphere: centre, radius, 5 unused); prim_type: NULL (all triangles) or one
// byte per primitive with the reference's enum (1 SPHERE, 2 TRIANGLE, primitives.cuh:11-15).  Throws on limits.
void build_reference_bvh2(const float *prim_pos, const uint8_t *prim_type, uint32_t n, uint32_t max_leaf_size, HostBVH &out);
void collapse_to_wide(HostBVH &bvh);
// prim_pos_flat / prim_type_flat: the primitives in the reference's flat order
void build_fast_tree(const float *prim_pos_flat, const uint8_t *prim_type_flat, uint32_t n, HostBVH &bvh);

}  // namespace mirt
