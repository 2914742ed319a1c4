#include <cstdlib>
// See host_bvh.hpp.  Compiled without FMA contraction: every float here must round as the reference's g++ build does.
#include "host_bvh.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <limits>
#include <stdexcept>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace mirt {

unsigned host_threads() {
  unsigned t = std::thread::hardware_concurrency();
  if (const char *e = std::getenv("OMP_NUM_THREADS")) {       // the GPU box's CPU share (nproc shows the whole machine there)
    int v = std::atoi(e);
    if (v >= 1) t = (unsigned) v;
  }
  if (const char *e = std::getenv("MIRT_BUILD_THREADS")) {
    int v = std::atoi(e);
    if (v >= 1) t = (unsigned) v;
  }
  if (t < 1) t = 1;
  if (t > 64) t = 64;
  return t;
}

void parallel_for(size_t n, const std::function<void(size_t, size_t)> &body, size_t min_chunk) {
  size_t parts = host_threads();
  if (n / (min_chunk ? min_chunk : 1) < parts) parts = n / (min_chunk ? min_chunk : 1);
  if (parts <= 1) {
    if (n) body(0, n);
    return;
  }
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err(parts);
  for (size_t k = 0; k < parts; k++) {
    const size_t b = n * k / parts, e = n * (k + 1) / parts;
    th.emplace_back([&, k, b, e] {
      try { body(b, e); } catch (...) { err[k] = std::current_exception(); }
    });
  }
  for (auto &x: th) x.join();
  for (auto &x: err) if (x) std::rethrow_exception(x);
}

struct WorkerPool::Impl {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable wake, done;
  const std::function<void(size_t)> *job = nullptr;
  size_t n_jobs = 0, finished = 0;
  std::atomic<size_t> next{0};
  unsigned busy = 0;            // workers inside the job loop of the current run
  uint64_t epoch = 0;
  bool stop = false;

  void drain(const std::function<void(size_t)> &f, size_t n) {      // take jobs until none is left; returns how many this thread ran
    size_t ran = 0;
    for (size_t j = next.fetch_add(1); j < n; j = next.fetch_add(1)) {
      try { f(j); } catch (...) {}                                    // (jobs report through their own state)
      ran++;
    }
    std::lock_guard<std::mutex> lk(m);
    finished += ran;
  }
  void worker() {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(size_t)> *f;
      size_t n;
      {
        std::unique_lock<std::mutex> lk(m);
        wake.wait(lk, [&] { return stop || epoch != seen; });
        if (stop) return;
        seen = epoch;
        // a worker that wakes after its run() has returned (the caller did all the jobs itself) finds no job: it must not
        // pair that run's job count with the claim counter of the next run
        if (job == nullptr) continue;
        f = job;
        n = n_jobs;
        busy++;
      }
      drain(*f, n);
      {
        std::lock_guard<std::mutex> lk(m);
        busy--;
      }
      done.notify_all();
    }
  }
};

WorkerPool::WorkerPool(unsigned workers) : impl(new Impl) {
  for (unsigned k = 0; k < workers; k++) impl->th.emplace_back([this] { impl->worker(); });
}
WorkerPool::~WorkerPool() {
  {
    std::lock_guard<std::mutex> lk(impl->m);
    impl->stop = true;
  }
  impl->wake.notify_all();
  for (auto &t: impl->th) t.join();
  delete impl;
}
unsigned WorkerPool::workers() const { return (unsigned) impl->th.size(); }
void WorkerPool::run(size_t n_jobs, const std::function<void(size_t)> &job) {
  if (n_jobs == 0) return;
  {
    std::lock_guard<std::mutex> lk(impl->m);
    impl->job = &job;
    impl->n_jobs = n_jobs;
    impl->finished = 0;
    impl->next.store(0);
    impl->epoch++;
  }
  impl->wake.notify_all();
  impl->drain(job, n_jobs);
  // all jobs done AND no worker still inside this run's loop (one that woke late must not meet the next run's counter)
  std::unique_lock<std::mutex> lk(impl->m);
  impl->done.wait(lk, [&] { return impl->finished == n_jobs && impl->busy == 0; });
  impl->job = nullptr;
}

namespace {

// Both top-down builders hand the two halves of a large range to two threads for the first few levels (the halves are
// independent: disjoint ranges of `order`, node records drawn from a pre-sized pool).  The trees do not depend on it.
int fork_levels() {
  const unsigned t = host_threads();
  int d = 0;
  while ((1u << d) < 2 * t && t > 1) d++;                      // about two tasks per thread
  return d;
}
struct StageClock {                                            // MIRT_BUILD_TIMING=1: stage times on stderr
  bool on = std::getenv("MIRT_BUILD_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char *what) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[mirt build] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};
// Node pool of a top-down builder: room for the worst case (2n - 1 records), but pages are only touched as records are
// handed out (a value-initialised std::vector of that size costs more than the build of a small tree).
template<class T>
struct Pool {
  T *p = nullptr;
  size_t cap = 0;
  std::atomic<uint32_t> next{0};
  Pool() = default;
  Pool(const Pool &) = delete;
  Pool &operator=(const Pool &) = delete;
  ~Pool() { std::free(p); }
  void reserve(size_t n) {
    std::free(p);
    p = (T *) std::malloc(n * sizeof(T));
    if (!p) throw std::bad_alloc();
    cap = n;
    next = 0;
  }
  uint32_t take() {
    const uint32_t id = next.fetch_add(1);
    if (id >= cap) throw std::runtime_error("node pool exhausted");
    new (&p[id]) T();
    return id;
  }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
  size_t size() const { return next.load(); }
};
constexpr uint32_t FORK_MIN = 16384;                           // primitives below which a range is not worth a thread

template<class L, class R>
void fork_join(bool parallel, L &&left, R &&right) {
  if (!parallel) {
    left();
    right();
    return;
  }
  std::exception_ptr err;
  std::thread t([&] {
    try { left(); } catch (...) { err = std::current_exception(); }
  });
  try {
    right();
  } catch (...) {
    t.join();
    throw;
  }
  t.join();
  if (err) std::rethrow_exception(err);
}

// util/misc.cuh:28-29 on the g++ path: maxf = std::max, minf = std::min (argument order matters for NaN)
inline float maxf(float a, float b) { return (a < b) ? b : a; }
inline float minf(float a, float b) { return (b < a) ? b : a; }

struct Box {
  float lo[3], hi[3];
};

struct Builder {
  const float *pos;
  const uint8_t *ptype = nullptr;
  uint32_t max_leaf;
  std::vector<Box> pbox;                // per input triangle (primitives.cuh:170-175)
  std::vector<float> pcent;             // centroid per triangle x3 (bbox.cuh:48-50)
  std::vector<uint32_t> order;          // plays the role of BVHAccel::primitives (bvh.cuh:172)

  struct TNode {
    Box bb;
    int32_t l = -1, r = -1;
    uint32_t start = 0, end = 0;
    uint32_t n_nodes = 1, n_prims = 0, height = 0;      // of the subtree: lets flatten() place subtrees independently
  };
  Pool<TNode> tree;                       // flatten() assigns the final order
  int fork_depth = 0;

  void prepare(uint32_t n) {
    pbox.resize(n);
    pcent.resize(3 * (size_t) n);
    order.resize(n);
    tree.reserve(2 * (size_t) n);
    fork_depth = fork_levels();
    parallel_for(n, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; i++) {
      const float *p = pos + 9 * (size_t) i;
      Box b;
      if (ptype && ptype[i] == 1) {                      // SPHERE: origin -/+ (r, r, r)   primitives.cuh:166-169
        for (int a = 0; a < 3; a++) {
          b.lo[a] = p[a] - p[3];
          b.hi[a] = p[a] + p[3];
        }
      } else {
        for (int a = 0; a < 3; a++) {
          b.lo[a] = p[a];
          b.hi[a] = p[a];                               // BBox(p1)
        }
        for (int v = 1; v < 3; v++)
          for (int a = 0; a < 3; a++) {                  // expand(p2), expand(p3)   bbox.cuh:38-46
            b.lo[a] = minf(b.lo[a], p[3 * v + a]);
            b.hi[a] = maxf(b.hi[a], p[3 * v + a]);
          }
      }
      pbox[i] = b;
      const float rc = (float) (1.0 / (double) 2.0f);    // Vector3f::operator/(float)  Vector3f.cuh:106-109
      for (int a = 0; a < 3; a++) pcent[3 * (size_t) i + a] = rc * (b.lo[a] + b.hi[a]);
      order[i] = (uint32_t) i;
    }
    });
  }

  int32_t construct(uint32_t start, uint32_t end, int depth = 0) {      // bvh.cuh:187-225
    Box bb;
    for (int a = 0; a < 3; a++) {
      bb.lo[a] = std::numeric_limits<float>::infinity();
      bb.hi[a] = -std::numeric_limits<float>::infinity();
    }
    for (uint32_t p = start; p < end; p++) {
      const Box &b = pbox[order[p]];
      for (int a = 0; a < 3; a++) {
        bb.lo[a] = minf(bb.lo[a], b.lo[a]);
        bb.hi[a] = maxf(bb.hi[a], b.hi[a]);
      }
    }
    const int32_t id = (int32_t) tree.take();
    tree[id].bb = bb;
    tree[id].n_prims = end - start;
    if ((int64_t) end - (int64_t) start <= (int64_t) max_leaf) {
      tree[id].start = start;
      tree[id].end = end;
      return id;
    }
    float ext[3] = {bb.hi[0] - bb.lo[0], bb.hi[1] - bb.lo[1], bb.hi[2] - bb.lo[2]};
    int axis = 0;                                        // std::max_element: first maximum under operator<
    for (int a = 1; a < 3; a++)
      if (ext[axis] < ext[a]) axis = a;
    float c[3] = {0.f, 0.f, 0.f};
    for (uint32_t p = start; p < end; p++)
      for (int a = 0; a < 3; a++) c[a] += pcent[3 * (size_t) order[p] + a];
    const float rc = (float) (1. / (double) (float) (end - start));   // operator/=  Vector3f.cuh:133-135
    const float cax = c[axis] * rc;
    auto pred = [&](uint32_t slot) { return pcent[3 * (size_t) order[slot] + axis] < cax; };
    // libstdc++ std::partition for bidirectional iterators (bits/stl_algo.h, __partition)
    uint32_t first = start, last = end;
    for (;;) {
      bool done = false;
      for (;;) {
        if (first == last) { done = true; break; }
        if (pred(first)) ++first; else break;
      }
      if (done) break;
      --last;
      for (;;) {
        if (first == last) { done = true; break; }
        if (!pred(last)) --last; else break;
      }
      if (done) break;
      std::swap(order[first], order[last]);
      ++first;
    }
    const uint32_t middle = first;
    if (middle == start || middle == end) {              // one-sided: leaf with all of them (bvh.cuh:216-220)
      tree[id].start = start;
      tree[id].end = end;
      return id;
    }
    int32_t l = -1, r = -1;
    fork_join(depth < fork_depth && end - start >= FORK_MIN, [&] { l = construct(start, middle, depth + 1); },
              [&] { r = construct(middle, end, depth + 1); });
    tree[id].l = l;
    tree[id].r = r;
    tree[id].n_nodes = 1 + tree[l].n_nodes + tree[r].n_nodes;
    tree[id].height = 1 + std::max(tree[l].height, tree[r].height);   // get_max_depth, bvh.cuh:297-300
    return id;
  }

  // bvh.cuh:272-290: pre-order, the right subtree first (nodes and primitives alike).  With the subtree sizes known the
  // two subtrees have fixed places in the output and can be written independently.
  void flatten(int32_t id, HostBVH &out, uint32_t n_idx, uint32_t prim_start, int depth = 0) {
    for (int a = 0; a < 3; a++) {
      out.nodes2[n_idx].minp[a] = tree[id].bb.lo[a];
      out.nodes2[n_idx].maxp[a] = tree[id].bb.hi[a];
    }
    if (tree[id].l < 0) {
      out.nodes2[n_idx].w0 = 0x80000000u | (tree[id].end - tree[id].start);
      out.nodes2[n_idx].w1 = prim_start;
      for (uint32_t p = tree[id].start; p < tree[id].end; p++) out.flat_to_orig[prim_start++] = order[p];
    } else {
      const int32_t l = tree[id].l, r = tree[id].r;
      const uint32_t right_idx = n_idx + 1, left_idx = right_idx + tree[r].n_nodes;
      out.nodes2[n_idx].w1 = right_idx;                             // idx_right = ++node_start
      out.nodes2[n_idx].w0 = left_idx;                              // idx_left
      fork_join(depth < fork_depth && tree[id].n_prims >= FORK_MIN,
                [&] { flatten(r, out, right_idx, prim_start, depth + 1); },
                [&] { flatten(l, out, left_idx, prim_start + tree[r].n_prims, depth + 1); });
    }
  }
};

inline float half_area(const Node2 &n) {
  float dx = n.maxp[0] - n.minp[0], dy = n.maxp[1] - n.minp[1], dz = n.maxp[2] - n.minp[2];
  return dx * dy + dy * dz + dz * dx;
}

struct CollapseIO {
  const std::vector<Node2> &nodes2;
  std::vector<Node4> &nodes4;
  std::vector<uint32_t> &big;
};

struct Collapser {
  CollapseIO b;
  uint32_t max_depth = 0;
  explicit Collapser(CollapseIO io) : b(io) {}

  static void set_child(Node4 &n, int k, const Node2 &box, uint32_t ref) {
    n.lox[k] = box.minp[0]; n.loy[k] = box.minp[1]; n.loz[k] = box.minp[2];
    n.hix[k] = box.maxp[0]; n.hiy[k] = box.maxp[1]; n.hiz[k] = box.maxp[2];
    n.child[k] = ref;
  }
  static void clear(Node4 &n) {
    for (int k = 0; k < 4; k++) {
      n.lox[k] = n.loy[k] = n.loz[k] = std::numeric_limits<float>::infinity();
      n.hix[k] = n.hiy[k] = n.hiz[k] = -std::numeric_limits<float>::infinity();
      n.child[k] = CHILD_EMPTY;
      n.pad[k] = 0;
    }
  }
  static uint32_t leaf_ref(uint32_t start, uint32_t count) {
    return CHILD_LEAF | (count << LEAF_COUNT_SHIFT) | start;
  }

  uint32_t child_ref(uint32_t n2, uint32_t depth) {
    const Node2 &n = b.nodes2[n2];
    if (n.w0 & 0x80000000u) {
      uint32_t count = n.w0 & 0x7fffffffu;
      if (count <= LEAF_MAX) return leaf_ref(n.w1, count);
      // A reference leaf can hold any number of primitives (one-sided partitions, bvh.cuh:216-220).  It stays ONE leaf
      // (one box test, all primitives tested in order, as the reference does): count field 0 + index into `big`.
      b.big.push_back(n.w1);
      b.big.push_back(count);
      return leaf_ref((uint32_t) (b.big.size() / 2 - 1), 0);
    }
    return build(n2, depth + 1);
  }

  uint32_t build(uint32_t n2, uint32_t depth) {
    uint32_t id = (uint32_t) b.nodes4.size();
    b.nodes4.emplace_back();
    clear(b.nodes4[id]);
    if (depth > max_depth) max_depth = depth;
    const Node2 &root = b.nodes2[n2];
    uint32_t kids[4];
    int nk = 0;
    if (root.w0 & 0x80000000u) {
      kids[nk++] = n2;                                   // a scene that is one leaf
    } else {
      kids[nk++] = root.w1;                              // right first: the reference pops it first (bvh.cuh:100-102)
      kids[nk++] = root.w0;
      while (nk < 4) {
        int best = -1;
        float best_a = -1.f;
        for (int k = 0; k < nk; k++) {
          const Node2 &c = b.nodes2[kids[k]];
          if (c.w0 & 0x80000000u) continue;
          float a = half_area(c);
          if (!(a <= best_a)) { best_a = a; best = k; }
        }
        if (best < 0) break;
        const Node2 &c = b.nodes2[kids[best]];
        for (int k = nk; k > best + 1; k--) kids[k] = kids[k - 1];
        kids[best] = c.w1;
        kids[best + 1] = c.w0;
        nk++;
      }
    }
    for (int k = 0; k < nk; k++) {
      uint32_t ref = child_ref(kids[k], depth);
      set_child(b.nodes4[id], k, b.nodes2[kids[k]], ref);   // (re-index: nodes4 may have been reallocated)
    }
    return id;
  }
};

}  // namespace

void build_reference_bvh2(const float *tri_pos, const uint8_t *prim_type, uint32_t n, uint32_t max_leaf_size, HostBVH &out) {
  if (n == 0) throw std::runtime_error("scene has no triangles");
  if (n >= (1u << LEAF_COUNT_SHIFT)) throw std::runtime_error("more than 2^27-1 triangles");
  Builder bd;
  bd.pos = tri_pos;
  bd.ptype = prim_type;
  bd.max_leaf = max_leaf_size;
  StageClock clk;
  bd.prepare(n);
  clk.lap("reference: boxes, centroids");
  int32_t root = bd.construct(0, n);
  clk.lap("reference: construct");
  out.nodes2.assign(bd.tree[root].n_nodes, Node2());
  out.flat_to_orig.assign(n, 0);
  bd.flatten(root, out, 0, 0);
  out.depth2 = bd.tree[root].height;
  clk.lap("reference: flatten");
}

void collapse_to_wide(HostBVH &bvh) {
  bvh.nodes4.clear();
  bvh.big.clear();
  bvh.nodes4.reserve(bvh.nodes2.size() / 2 + 4);
  Collapser c(CollapseIO{bvh.nodes2, bvh.nodes4, bvh.big});
  c.build(0, 0);
  bvh.depth4 = c.max_depth;
}

// ------------------------------------------------------------------------------------------------ fast tree
// The tree the ordered traversal (mirt_options.traversal = 0) walks.  It is NOT the reference's: binned-SAH splits over
// tight per-triangle boxes, leaves of at most 3 triangles, its own triangle order (each packed triangle carries its flat
// index of the reference order, which is what hit indices and the tie rule are expressed in).  Any tree over the same
// triangles yields the reference's closest hit (minimum t, ties to the larger flat index) as long as its boxes contain
// their triangles, which exact min/max of the vertices guarantees.
namespace {

struct SahBuilder {
  const float *pos;                       // primitives in FLAT order, 9 floats each
  const uint8_t *ptype = nullptr;
  std::vector<Box> pbox;
  std::vector<float> cent, parea;
  std::vector<uint32_t> order;
  Pool<Node2> &out;
  std::vector<uint32_t> &perm;            // = order when the build is done (a leaf's primitives are its range of `order`)
  int fork_depth = 0;
  uint32_t max_leaf;
  static constexpr int MAX_BINS = 64;
  int BINS = 16;                          // MIRT_SAH_BINS
  float leaf_ct = 0.5f;                   // MIRT_SAH_LEAF_CT: what dividing a would-be leaf costs, in primitive tests (1e30 = never divide, the
                                          // rule until round 3: conference* -1.5 %, bunny* -2.5 %, buddha* -3.4 %, sportsCar* -3.9 %, tools/leaf_sweep.sh)
  float peel = 0.5f;                      // primitives whose box area exceeds peel x the node's go into a subtree of their own
                                          // (0 = off; tools/peel_sweep.sh: conference* 135.7 -> 133.3 ms/frame)

  SahBuilder(Pool<Node2> &o, std::vector<uint32_t> &p) : out(o), perm(p) {}

  // plain compare-and-select (one minss / maxss; std::fmin is a libm call that costs more than the rest of the loop): every
  // operand is finite, prepare() refuses anything else
  static float lo_of(float a, float b) { return b < a ? b : a; }
  static float hi_of(float a, float b) { return a < b ? b : a; }
  static float area(const Box &b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
  }
  static void grow(Box &a, const Box &b) {
    for (int k = 0; k < 3; k++) {
      a.lo[k] = lo_of(a.lo[k], b.lo[k]);
      a.hi[k] = hi_of(a.hi[k], b.hi[k]);
    }
  }
  static Box empty() {
    Box b;
    for (int k = 0; k < 3; k++) {
      b.lo[k] = std::numeric_limits<float>::infinity();
      b.hi[k] = -std::numeric_limits<float>::infinity();
    }
    return b;
  }

  void prepare(uint32_t n) {
    pbox.resize(n);
    cent.resize(3 * (size_t) n);
    parea.resize(n);
    order.resize(n);
    out.reserve(2 * (size_t) n);
    fork_depth = fork_levels();
    std::atomic<bool> bad{false};
    parallel_for(n, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; i++) {
      const float *p = pos + 9 * (size_t) i;
      Box b = empty();
      if (ptype && ptype[i] == 1) {
        // sphere: one ulp of room on each side (the intersector's t is not computed from these planes)
        for (int k = 0; k < 3; k++) {
          b.lo[k] = std::nextafter(p[k] - p[3], -std::numeric_limits<float>::infinity());
          b.hi[k] = std::nextafter(p[k] + p[3], std::numeric_limits<float>::infinity());
        }
      } else {
        for (int v = 0; v < 3; v++)
          for (int k = 0; k < 3; k++) {
            b.lo[k] = lo_of(b.lo[k], p[3 * v + k]);
            b.hi[k] = hi_of(b.hi[k], p[3 * v + k]);
          }
      }
      pbox[i] = b;
      for (int k = 0; k < 3; k++) cent[3 * (size_t) i + k] = 0.5f * (b.lo[k] + b.hi[k]);
      parea[i] = area(b);
      order[i] = (uint32_t) i;
      bool finite = std::isfinite(parea[i]);
      const int used = (ptype && ptype[i] == 1) ? 4 : 9;     // a sphere's record: centre, radius, 5 unused
      for (int k = 0; k < used; k++) finite = finite && std::isfinite(p[k]);
      for (int k = 0; k < 3; k++)
        finite = finite && std::isfinite(b.lo[k]) && std::isfinite(b.hi[k]) && std::isfinite(cent[3 * (size_t) i + k]);
      if (!finite) bad.store(true, std::memory_order_relaxed);
    }
    });
    if (bad.load()) throw std::runtime_error("non-finite vertex: no quantised tree");
  }

  uint32_t build(uint32_t s, uint32_t e, int depth = 0) {
    Box bb = empty(), cb = empty();
    float max_area = 0.f;                                  // of a single primitive's box
    for (uint32_t i = s; i < e; i++) {
      grow(bb, pbox[order[i]]);
      max_area = hi_of(max_area, parea[order[i]]);
      for (int k = 0; k < 3; k++) {
        float c = cent[3 * (size_t) order[i] + k];
        cb.lo[k] = lo_of(cb.lo[k], c);
        cb.hi[k] = hi_of(cb.hi[k], c);
      }
    }
    const uint32_t id = out.take();
    for (int k = 0; k < 3; k++) { out[id].minp[k] = bb.lo[k]; out[id].maxp[k] = bb.hi[k]; }
    const uint32_t n = e - s;
    if (n <= max_leaf) {
      // Few enough for a leaf - unless dividing them still pays.  The traversal tests a leaf's primitives one per trip, the
      // price of a node step, and the fold to 8-wide nodes mostly absorbs a split at this level into the parent's free child
      // slots: so a leaf of n costs n tests, a split leaf_ct + the area-weighted tests of its halves.  Every division of the
      // (at most 15) primitives along their centroid order on each axis is tried.
      uint32_t cut = 0;
      if (n >= 2 && leaf_ct < 1.0e9f && area(bb) > 0.f) {
        uint32_t best_idx[LEAF_MAX];
        float best_cost = std::numeric_limits<float>::infinity();
        for (int ax = 0; ax < 3; ax++) {
          uint32_t idx[LEAF_MAX];
          for (uint32_t i = 0; i < n; i++) idx[i] = order[s + i];
          for (uint32_t i = 1; i < n; i++) {               // insertion sort by (centroid, index)
            const uint32_t v = idx[i];
            const float cv = cent[3 * (size_t) v + ax];
            uint32_t j = i;
            while (j > 0 && (cent[3 * (size_t) idx[j - 1] + ax] > cv || (cent[3 * (size_t) idx[j - 1] + ax] == cv && idx[j - 1] > v))) {
              idx[j] = idx[j - 1];
              j--;
            }
            idx[j] = v;
          }
          float ra[LEAF_MAX];
          Box acc = empty();
          for (uint32_t k = n - 1; k > 0; k--) { grow(acc, pbox[idx[k]]); ra[k] = area(acc); }
          acc = empty();
          for (uint32_t k = 1; k < n; k++) {
            grow(acc, pbox[idx[k - 1]]);
            const float cost = area(acc) * (float) k + ra[k] * (float) (n - k);
            if (cost < best_cost) {
              best_cost = cost;
              cut = k;
              for (uint32_t i = 0; i < n; i++) best_idx[i] = idx[i];
            }
          }
        }
        if (cut != 0 && leaf_ct * area(bb) + best_cost < (float) n * area(bb)) {
          for (uint32_t i = 0; i < n; i++) order[s + i] = best_idx[i];
        } else {
          cut = 0;
        }
      }
      if (cut == 0) {
        out[id].w0 = 0x80000000u | n;
        out[id].w1 = s;
        return id;
      }
      const uint32_t l = build(s, s + cut, depth + 1);
      const uint32_t r = build(s + cut, e, depth + 1);
      out[id].w0 = l;
      out[id].w1 = r;
      return id;
    }
    // Primitives that span most of the node (a room's walls around its furniture) would inflate whichever half they fall
    // into: set them aside as one child of this node, the rest as the other.
    if (peel > 0.f && n > max_leaf && max_area > peel * area(bb)) {
      const float limit = peel * area(bb);
      uint32_t i = s, j = e;
      while (i < j) {
        if (parea[order[i]] > limit) i++;
        else std::swap(order[i], order[--j]);
      }
      const uint32_t big = i - s;
      if (big > 0 && big <= 8 && big < n) {
        uint32_t l = 0, r = 0;
        l = build(s, s + big, depth + 1);
        r = build(s + big, e, depth + 1);
        out[id].w0 = l;
        out[id].w1 = r;
        return id;
      }
    }
    // binned SAH over the three axes
    int best_axis = -1, best_bin = -1;
    float best_cost = std::numeric_limits<float>::infinity();
    Box bins[3][MAX_BINS];
    uint32_t cnt[3][MAX_BINS];
    float lo3[3], scale3[3];
    bool use[3];
    for (int ax = 0; ax < 3; ax++) {
      const float ext = cb.hi[ax] - cb.lo[ax];
      use[ax] = ext > 0;
      lo3[ax] = cb.lo[ax];
      scale3[ax] = use[ax] ? BINS / ext : 0.f;
      for (int k = 0; k < BINS; k++) { bins[ax][k] = empty(); cnt[ax][k] = 0; }
    }
    for (uint32_t i = s; i < e; i++) {                     // one pass over the primitives fills the bins of all three axes
      const uint32_t pi = order[i];
      const Box &pb = pbox[pi];
      for (int ax = 0; ax < 3; ax++) {
        if (!use[ax]) continue;
        int k = (int) ((cent[3 * (size_t) pi + ax] - lo3[ax]) * scale3[ax]);
        k = k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
        grow(bins[ax][k], pb);
        cnt[ax][k]++;
      }
    }
    for (int ax = 0; ax < 3; ax++) {
      if (!use[ax]) continue;
      float ra[MAX_BINS];
      uint32_t rn[MAX_BINS];
      Box acc = empty();
      uint32_t c = 0;
      for (int k = BINS - 1; k > 0; k--) { grow(acc, bins[ax][k]); c += cnt[ax][k]; ra[k] = area(acc); rn[k] = c; }
      acc = empty();
      c = 0;
      for (int k = 0; k < BINS - 1; k++) {
        grow(acc, bins[ax][k]);
        c += cnt[ax][k];
        if (c == 0 || rn[k + 1] == 0) continue;
        float cost = area(acc) * (float) c + ra[k + 1] * (float) rn[k + 1];
        if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = k; }
      }
    }
    uint32_t mid;
    if (best_axis < 0) {
      mid = s + n / 2;                                   // all centroids coincide: split by count
    } else {
      const float lo = cb.lo[best_axis], scale = BINS / (cb.hi[best_axis] - cb.lo[best_axis]);
      uint32_t i = s, j = e;
      while (i < j) {
        int b = (int) ((cent[3 * (size_t) order[i] + best_axis] - lo) * scale);
        b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
        if (b <= best_bin) i++;
        else std::swap(order[i], order[--j]);
      }
      mid = i;
      if (mid == s || mid == e) mid = s + n / 2;
    }
    uint32_t l = 0, r = 0;
    fork_join(depth < fork_depth && n >= FORK_MIN, [&] { l = build(s, mid, depth + 1); }, [&] { r = build(mid, e, depth + 1); });
    out[id].w0 = l;
    out[id].w1 = r;
    return id;
  }
};

}  // namespace

// Folds the binary SAH tree to 8-wide nodes (largest-area child expanded first) and quantises the child boxes.
struct Collapser8 {
  const Pool<Node2> &n2;
  std::vector<Node8Q> out;
  uint32_t max_depth = 0;
  explicit Collapser8(const Pool<Node2> &nodes) : n2(nodes) {}

  // Which binary nodes become 8-wide nodes: the choice that minimises the summed surface area of the wide nodes (the
  // expected number of node steps of a random ray; the leaves are the binary tree's, so their cost is the same whatever is
  // chosen).  Dynamic programme over the binary tree (Ylitie, Karras, Laine 2017, section 4.1, with fixed leaves):
  //   c(n, i)  = least cost of handing the subtree of n to its ancestor as at most i children, i = 1..7
  //   d(n, j)  = min over k of c(left, k) + c(right, j - k): n dissolved, its two sides share j slots, j = 2..8
  //   c(n, 1)  = area(n) + d(n, 8)   (n becomes a wide node of its own);   c(n, i) = min(d(n, i), c(n, i - 1))
  // Measured (tools/fold_ab.sh, profiles/r3d_fold_ab.txt): 37 % fewer wide nodes on conference*, and no gain on any workload
  // (node visits 4.31 -> 4.07 but primitive tests 2.60 -> 2.83 there, both up on the other three): leaves pulled up into
  // large nodes get their boxes on that node's coarse 8-bit grid.  Off by default; MIRT_FOLD_DP=1 turns it on.
  bool optimal = false;                                  // false: the greedy rule (largest-area child dissolved first)
  std::vector<float> cst;                                // [node][7]
  std::vector<uint8_t> kk;                               // [node][7]: the k of d(n, j), j = 2..8
  std::vector<uint8_t> from_d;                           // [node]: bit i set if c(n, i) = d(n, i) < c(n, i - 1)
  void plan(uint32_t n) {
    const Node2 &nd = n2[n];
    if (nd.w0 & 0x80000000u) {
      for (int i = 0; i < 7; i++) cst[7 * (size_t) n + i] = 0.f;
      return;
    }
    plan(nd.w0);
    plan(nd.w1);
    const float *cl = &cst[7 * (size_t) nd.w0], *cr = &cst[7 * (size_t) nd.w1];
    float d[9];
    for (int j = 2; j <= 8; j++) {
      float best = std::numeric_limits<float>::infinity();
      int bk = 1;
      for (int k = (j - 7 > 1 ? j - 7 : 1); k <= (j - 1 < 7 ? j - 1 : 7); k++) {
        const float v = cl[k - 1] + cr[j - k - 1];
        if (v < best) { best = v; bk = k; }
      }
      d[j] = best;
      kk[7 * (size_t) n + (j - 2)] = (uint8_t) bk;
    }
    float *c = &cst[7 * (size_t) n];
    c[0] = d[8] + half_area(nd);
    uint8_t bits = 0;
    for (int i = 2; i <= 7; i++) {
      if (d[i] < c[i - 2]) { c[i - 1] = d[i]; bits |= (uint8_t) (1u << i); }
      else c[i - 1] = c[i - 2];
    }
    from_d[n] = bits;
  }
  void place(uint32_t n, int i, uint32_t *kids, int &nk) const {      // the subtree of n as at most i children
    const Node2 &nd = n2[n];
    if (!(nd.w0 & 0x80000000u)) {
      while (i > 1 && !((from_d[n] >> i) & 1u)) i--;
      if (i > 1) { expand(n, i, kids, nk); return; }
    }
    kids[nk++] = n;
  }
  void expand(uint32_t n, int j, uint32_t *kids, int &nk) const {     // n dissolved into j slots
    const int k = kk[7 * (size_t) n + (j - 2)];
    place(n2[n].w0, k, kids, nk);
    place(n2[n].w1, j - k, kids, nk);
  }
  void prepare(uint32_t root_id) {
    if (!optimal) return;
    cst.assign(7 * n2.size(), 0.f);
    kk.assign(7 * n2.size(), 1);
    from_d.assign(n2.size(), 0);
    plan(root_id);
  }

  uint32_t build(uint32_t root_id, uint32_t depth) {
    const uint32_t id = (uint32_t) out.size();
    out.emplace_back();
    if (depth > max_depth) max_depth = depth;
    const Node2 &root = n2[root_id];
    uint32_t kids[8];
    int nk = 0;
    if (root.w0 & 0x80000000u) {
      kids[nk++] = root_id;                              // a scene that is one leaf
    } else if (optimal) {
      expand(root_id, 8, kids, nk);
    } else {
      kids[nk++] = root.w0;
      kids[nk++] = root.w1;
      while (nk < 8) {
        int best = -1;
        float best_a = -1.f;
        for (int k = 0; k < nk; k++) {
          const Node2 &c = n2[kids[k]];
          if (c.w0 & 0x80000000u) continue;
          const float a = half_area(c);
          if (!(a <= best_a)) { best_a = a; best = k; }
        }
        if (best < 0) break;
        const Node2 c = n2[kids[best]];
        kids[best] = c.w0;
        kids[nk++] = c.w1;
      }
    }
    Node8Q q;
    memset(&q, 0, sizeof q);
    for (int k = 0; k < 8; k++) q.child[k] = CHILD_EMPTY;
    // grid: origin = the node's minimum corner, step = smallest power of two that spans the extent in 255 steps
    int eb[3];
    double step[3];
    for (int a = 0; a < 3; a++) {
      q.p[a] = root.minp[a];
      if (!std::isfinite(root.minp[a]) || !std::isfinite(root.maxp[a]))
        throw std::runtime_error("non-finite vertex: no quantised tree");
      const double ext = (double) root.maxp[a] - (double) root.minp[a];
      int e = -126;
      if (ext > 0) {
        int fe;
        (void) std::frexp(ext / 255.0, &fe);             // ext / 255 = m * 2^fe, m in [0.5, 1)  =>  2^fe >= ext / 255
        e = fe;
      }
      if (e < -126) e = -126;
      if (e > 127) throw std::runtime_error("scene extent too large for the quantised tree");
      eb[a] = e + 127;
      step[a] = std::ldexp(1.0, e);
    }
    uint8_t *qlo[3] = {q.qlo_x, q.qlo_y, q.qlo_z}, *qhi[3] = {q.qhi_x, q.qhi_y, q.qhi_z};
    for (int a = 0; a < 3; a++) {
      for (;;) {                                          // (a second pass only when rounding pushes a plane past 255)
        bool fits = true;
        for (int k = 0; k < nk && fits; k++) {
          const Node2 &c = n2[kids[k]];
          const double p = (double) q.p[a], cmin = (double) c.minp[a], cmax = (double) c.maxp[a];
          double lo = std::floor((cmin - p) / step[a]);
          double hi = std::ceil((cmax - p) / step[a]);
          if (lo < 0) lo = 0;
          // p + q * 2^e is exact in double; the quotient above is not when the operands differ by more than 2^53
          while (lo > 0 && p + lo * step[a] > cmin) lo -= 1;
          while (p + hi * step[a] < cmax) hi += 1;
          if (hi > 255 || p + lo * step[a] > cmin) { fits = false; break; }
          qlo[a][k] = (uint8_t) lo;
          qhi[a][k] = (uint8_t) hi;
        }
        if (fits) break;
        if (++eb[a] > 254) throw std::runtime_error("scene extent too large for the quantised tree");
        step[a] *= 2.0;
      }
    }
    q.exps = (uint32_t) eb[0] | ((uint32_t) eb[1] << 8) | ((uint32_t) eb[2] << 16);
    out[id] = q;
    for (int k = 0; k < nk; k++) {
      const Node2 &c = n2[kids[k]];
      uint32_t ref;
      if (c.w0 & 0x80000000u) {
        const uint32_t count = c.w0 & 0x7fffffffu;
        if (count == 0 || count > LEAF_MAX) throw std::runtime_error("SAH leaf outside 1..15 primitives");
        ref = CHILD_LEAF | (count << LEAF_COUNT_SHIFT) | c.w1;
      } else {
        ref = build(kids[k], depth + 1);
      }
      out[id].child[k] = ref;                            // (re-index: `out` may have been reallocated)
    }
    return id;
  }
};

void build_fast_tree(const float *tri_pos_flat, const uint8_t *prim_type_flat, uint32_t n, HostBVH &bvh) {
  Pool<Node2> n2;
  bvh.fast_to_flat.clear();
  SahBuilder sb(n2, bvh.fast_to_flat);
  sb.pos = tri_pos_flat;
  sb.ptype = prim_type_flat;
  sb.max_leaf = 3;                                         // measured best of 1..8 on the bench workload (tools/leaf_sweep.sh, with leaf_ct)
  if (const char *e = std::getenv("MIRT_FAST_LEAF")) {     // tuning knob (1..15: the inline leaf count field)
    int v = std::atoi(e);
    if (v >= 1 && v <= (int) LEAF_MAX) sb.max_leaf = (uint32_t) v;
  }
  StageClock clk;
  if (const char *e = std::getenv("MIRT_SAH_PEEL")) sb.peel = (float) std::atof(e);
  if (const char *e = std::getenv("MIRT_SAH_LEAF_CT")) sb.leaf_ct = (float) std::atof(e);
  if (const char *e = std::getenv("MIRT_SAH_BINS")) {
    int v = std::atoi(e);
    if (v >= 2 && v <= SahBuilder::MAX_BINS) sb.BINS = v;
  }
  sb.prepare(n);
  clk.lap("own tree: boxes, centroids");
  sb.build(0, n);
  clk.lap("own tree: binned SAH");
  bvh.fast_to_flat = sb.order;
  Collapser8 c8(n2);
  if (const char *e = std::getenv("MIRT_FOLD_DP")) c8.optimal = std::atoi(e) != 0;
  c8.prepare(0);
  c8.build(0, 0);
  clk.lap("own tree: fold to 8-wide");
  bvh.depth8 = c8.max_depth;
  bvh.nodes8q.resize(c8.out.size() * (sizeof(Node8Q) / 4));
  memcpy(bvh.nodes8q.data(), c8.out.data(), c8.out.size() * sizeof(Node8Q));
}

}  // namespace mirt
