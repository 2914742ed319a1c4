// libmirt C ABI (include/mirt.h): host orchestration of the wavefront path tracer.  No CPU fallback anywhere: every
// compute entry point needs a HIP device and fails with MIRT_E_DEVICE without one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mirt.h"
#include "device_build.hpp"
#include "host_bvh.hpp"
#include "host_film.hpp"
#include "kernels.hpp"

using namespace mirt;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

static const bool g_debug = getenv("MIRT_DEBUG") != nullptr;
static const bool g_debug_iters = getenv("MIRT_DEBUG_ITERS") != nullptr;    // per-iteration queue counters on stderr
#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    if (g_debug) { fprintf(stderr, "[mirt] %s:%d %s\n", __func__, __LINE__, #expr); fflush(stderr); }    \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return fail(MIRT_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));                   \
  } while (0)

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  hipError_t ensure(size_t n) {
    if (n <= bytes && p) return hipSuccess;
    if (p) (void) hipFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, n ? n : 16);
    if (e == hipSuccess) bytes = n ? n : 16;
    return e;
  }
  void release() {
    if (p) (void) hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template<class T> T *as() const { return (T *) p; }
};

constexpr int RING = 4;
constexpr int MAX_PIPES = 4;

struct Global {
  bool ready = false;
  int device = -1;
  int n_cu = 0;
  int blocks_per_cu = 2;
  uint32_t refill_min = 16;             // MIRT_REFILL_MIN: idle lanes before a wave fetches rays (tools/env_sweep.sh at 8 streams: 8 -> 16, -1 %)
  bool serial = false;
  uint32_t static_16th = 8;             // MIRT_STATIC_16TH: share of a queue handed out as per-workgroup slices (12 -> 8: -1.5 %; the rest is the balancing pool)
  // Two pipes: a frame with sample streams is rendered as two independent wavefront loops over halves of the slots, each
  // with its own pair of streams (main: k_logic + closest-hit traversal, side: shadow traversal), so that the HBM-bound
  // k_logic of one half runs while the other half is in its traversals.  `stream` / `side` are pipe 0's.
  hipStream_t stream = nullptr, side = nullptr;
  hipStream_t pstream[MAX_PIPES] = {}, pside[MAX_PIPES] = {};
  hipEvent_t fork_ev[MAX_PIPES][RING] = {}, join_ev[MAX_PIPES][RING] = {};
  hipEvent_t pipe_ev[MAX_PIPES] = {};
  int pipes = 2;                        // MIRT_PIPES: independent wavefront loops per frame (1 ... 4; needs as many sample streams)
  bool one_pipe = false;
  uint32_t pool_group = 64;             // MIRT_POOL_GROUP: pixel slots per home group of the share pool (a multiple of 64)
  uint32_t live_16th = 4;               // MIRT_LIVE_16TH: k_logic walks the live-slot list once it is below this share of the slots
  bool blocks_forced = false;           // MIRT_BLOCKS_PER_CU given
  float table[256];
  bool table_exact = false;
  DevBuf d_table, d_rgb;
  DevBuf ovf;                            // traversal-stack overflow regions (beyond the LDS part), kept across scenes: grows, never shrinks
  uint32_t *h_film = nullptr;            // pinned: the packed film on its way to the caller
  size_t h_film_words = 0;
  hipEvent_t film_ev[8] = {};
  WorkerPool *film_pool = nullptr;       // host threads that widen the film as its pieces land (mirt_tonemap), kept across frames
  uint32_t *h_counts = nullptr;          // host-mapped, coherent: MAX_PIPES x RING slots of HSLOT words, written by k_logic
  uint32_t *d_counts = nullptr;          // the same memory as the device sees it
  uint32_t *h_tail = nullptr;            // pinned: a frame's control blocks (error flags) and ray statistics, copied before the frame's last event
  uint32_t seq = 0;                      // stamp of the last iteration issued (never 0)
  hipEvent_t ring_ev[MAX_PIPES][RING] = {};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;    // first launch / accumulation buffer complete (render_ms)
  hipEvent_t ev_in = nullptr;               // entry: recorded on the caller's NULL stream, waited for by `stream`
  std::vector<hipEvent_t> pool;          // timing events for profiled renders (recorded without host syncs)
  size_t pool_used = 0;
  hipEvent_t pool_next() {
    if (pool_used == pool.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    return pool[pool_used++];
  }
};
// One context per host thread: a process that drives several GPUs (mirt_group_*, group.hip) runs one thread per device, each
// with its own streams, events and pinned memory; a single-threaded caller sees exactly one context, as before.
thread_local Global G;

struct Timed { int kind; hipEvent_t a, b; };   // kind 0 logic, 1 closest trace, 2 shadow trace

// control block layout (uint32 words, every counter on a 128-byte line of its own, see kernels.hpp):
// queue counters | fetch counters of the closest launch | of the shadow launch | redo counts | deferred-microfacet
// count | error flag.  Everything before the error flag is cleared for every iteration - by k_logic of
// the iteration before (three blocks rotate per pipe), not by a memset dispatch.
enum {
  CTRL_QCOUNT = 0,
  CTRL_WORK = QC_WORDS,
  CTRL_REDO = CTRL_WORK + 2 * WORK_PARTS * CTR_STRIDE,
  CTRL_MF = CTRL_REDO + 2 * CTR_STRIDE,
  CTRL_ERR = CTRL_MF + CTR_STRIDE,
  CTRL_WORDS = CTRL_ERR + CTR_STRIDE
};
constexpr int CTRL_BLOCKS = 3;                           // per pipe: iteration i uses block i % 3 and clears block (i + 1) % 3
constexpr size_t TAIL_BYTES = (size_t) CTRL_BLOCKS * MAX_PIPES * CTRL_WORDS * 4 + MAX_PIPES * 16 * 8;   // Global::h_tail
constexpr size_t RAY_STATS_BYTES = (size_t) MAX_PIPES * 16 * 8;
constexpr int HSLOT = 8;                                 // words per host ring slot: 5 queue counters, 2 spare, stamp

// ---- mirt_trace_static: everything its kernels need beyond the caller's inputs and outputs lives in ONE file-scope device
// array (each GPU the library runs on holds its own copy until the process ends).  Host code never uses the array's
// name as an address: mirt_init asks the runtime for the address on the current device (hipGetSymbolAddress) and keeps it.
constexpr uint32_t STATIC_MAX_RAYS = MIRT_TRACE_STATIC_MAX_RAYS;
constexpr uint32_t STATIC_GRID = 128;                                    // workgroups of its traversal launches
constexpr uint32_t STATIC_OVF_DEPTH = 24;                                // stack entries per lane beyond the LDS part (deeper rays: k_redo)
constexpr size_t align256(size_t v) { return (v + 255) & ~(size_t) 255; }
constexpr size_t SO_CTRL = 0;
constexpr size_t SO_RAY_O = align256(SO_CTRL + CTRL_WORDS * 4);
constexpr size_t SO_RAY_D = SO_RAY_O + (size_t) STATIC_MAX_RAYS * 16;
constexpr size_t SO_HIT = SO_RAY_D + (size_t) STATIC_MAX_RAYS * 16;
constexpr size_t SO_OCC = SO_HIT + (size_t) STATIC_MAX_RAYS * 16;
constexpr size_t SO_ANY_T = SO_OCC + (size_t) STATIC_MAX_RAYS * 4;
constexpr size_t SO_QUEUE = SO_ANY_T + (size_t) STATIC_MAX_RAYS * 4;
constexpr size_t SO_REDO = SO_QUEUE + (size_t) STATIC_MAX_RAYS * 4;
constexpr size_t SO_OVF = SO_REDO + (size_t) STATIC_MAX_RAYS * 4;
constexpr size_t SCRATCH_BYTES = SO_OVF + (size_t) STATIC_OVF_DEPTH * STATIC_GRID * TRACE_BLOCK * sizeof(uint2);   // ~ 10 MB
constexpr int MAX_DEVICES = 64;
unsigned char *g_scratch_ptr[MAX_DEVICES] = {};                          // per device: its copy's address
std::mutex g_scratch_mutex;

}  // namespace

static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

namespace mirt {
hipStream_t current_stream() { return G.stream; }                        // for group.hip
int set_error(int code, const std::string &msg) { return fail(code, msg); }
}  // namespace mirt

struct mirt_scene {
  HostBVH bvh;
  uint32_t n_tris = 0, n_mats = 0, n_lights = 0;
  bool has_microfacet = false;
  bool has_spheres = false;
  float sph_rmin = 0.f;                  // smallest sphere radius (sphere scenes)
  bool fast_ok = true;                   // false: the SAH tree could not be built (non-finite vertices): traversal 0 runs as 1
  uint32_t own_builder = 0;              // 0 host binned SAH, 1 device (build.hip)
  uint32_t n_nodes8 = 0, depth8 = 0;
  float build_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  DevBuf nodes4, nodes2, tris, tri_nrm, mats, lights, big, trisf, nodes8q, verts;
  uint32_t ovf_depth = 0;
  uint32_t trace_threads = 0;
  int device = -1;                       // the HIP device its buffers live on (the creating thread's context)
  // work buffers (sized per render)
  DevBuf st, mask, total, accum, dmask, ray_o, ray_d, hit, sh_o, sh_d, sh_c, sh_occ, ext_queue, sh_queue, redo_q[2], mf_queue, live[2];
  DevBuf ctrl, ray_stats, out_accum, rgb, probe_idx, probe_t, any_t, planes;
  std::vector<mirt_light> host_lights;   // for the table of constant shadow-ray directions (directional lights)
  float const_dir[4] = {0, 0, 0, 0};     // rebuilt when samples_per_light changes
  uint32_t const_dir_spl = 0, const_k = NO_CONST_K;
  struct FillSig {                       // what the pre-filled sh_d records of entry const_k were written for (fill_constant_directions)
    const void *buf = nullptr;
    uint32_t pixel_slots = 0, streams = 0, K = 0, pipes = 0, k = 0, pad = 0;     // (no padding bytes: compared with memcmp)
    float dir[4] = {0, 0, 0, 0};
  } fill_sig;
  size_t device_bytes() const {
    const DevBuf *all[] = {&verts, &nodes4, &nodes2, &tris, &tri_nrm, &mats, &lights, &big, &trisf, &nodes8q, &st, &mask, &total, &accum, &dmask,
                           &ray_o, &ray_d, &hit, &sh_o, &sh_d, &sh_c, &sh_occ, &ext_queue, &sh_queue, &redo_q[0],
                           &redo_q[1], &ctrl, &ray_stats, &out_accum, &rgb, &probe_idx, &probe_t, &any_t, &mf_queue, &planes, &live[0], &live[1]};
    size_t s = 0;
    for (auto *b: all) s += b->bytes;
    return s;
  }
  void release_all() {
    DevBuf *all[] = {&verts, &nodes4, &nodes2, &tris, &tri_nrm, &mats, &lights, &big, &trisf, &nodes8q, &st, &mask, &total, &accum, &dmask,
                     &ray_o, &ray_d, &hit, &sh_o, &sh_d, &sh_c, &sh_occ, &ext_queue, &sh_queue, &redo_q[0], &redo_q[1],
                     &ctrl, &ray_stats, &out_accum, &rgb, &probe_idx, &probe_t, &any_t, &mf_queue, &planes, &live[0], &live[1]};
    for (auto *b: all) b->release();
  }
};

extern "C" {

const char *mirt_last_error(void) { return g_err.c_str(); }
const char *mirt_version(void) { return "mirt 0.1 (gfx950 wavefront path tracer)"; }

int mirt_init(int device) {
  if (G.ready && G.device == device) return MIRT_OK;
  if (G.ready) mirt_shutdown();
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(MIRT_E_DEVICE, "no HIP device available (libmirt has no CPU fallback)");
  if (device < 0 || device >= n) return fail(MIRT_E_ARG, "device index out of range");
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  G.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  for (int p = 0; p < MAX_PIPES; p++) {
    HIPCHK(hipStreamCreateWithFlags(&G.pstream[p], hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&G.pside[p], hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&G.pipe_ev[p], hipEventDisableTiming));
    for (int i = 0; i < RING; i++) {
      HIPCHK(hipEventCreateWithFlags(&G.ring_ev[p][i], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&G.fork_ev[p][i], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&G.join_ev[p][i], hipEventDisableTiming));
    }
  }
  G.stream = G.pstream[0];
  G.side = G.pside[0];
  if (device < MAX_DEVICES) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
      void *ptr = nullptr;
      HIPCHK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));             // this device's copy (hipSetDevice above)
      g_scratch_ptr[device] = (unsigned char *) ptr;
    }
  }
  G.table_exact = build_tonemap_table(G.table);
  HIPCHK(G.d_table.ensure(sizeof(G.table)));
  HIPCHK(hipMemcpy(G.d_table.p, G.table, sizeof(G.table), hipMemcpyHostToDevice));
  HIPCHK(hipHostMalloc((void **) &G.h_counts, MAX_PIPES * RING * QC_WORDS * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
  memset(G.h_counts, 0, MAX_PIPES * RING * QC_WORDS * sizeof(uint32_t));
  HIPCHK(hipHostGetDevicePointer((void **) &G.d_counts, G.h_counts, 0));
  HIPCHK(hipHostMalloc((void **) &G.h_tail, TAIL_BYTES, hipHostMallocDefault));
  for (auto &e: G.film_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(hipEventCreate(&G.ev0));
  HIPCHK(hipEventCreate(&G.ev1));
  HIPCHK(hipEventCreateWithFlags(&G.ev_in, hipEventDisableTiming));
  G.blocks_per_cu = trace4_blocks_per_cu();
  if (const char *s = getenv("MIRT_BLOCKS_PER_CU")) {
    int v = atoi(s);
    if (v >= 1 && v <= 16) { G.blocks_per_cu = v; G.blocks_forced = true; }
  }
  if (const char *s = getenv("MIRT_SERIAL")) G.serial = atoi(s) != 0;
  if (const char *s = getenv("MIRT_ONE_PIPE")) G.one_pipe = atoi(s) != 0;
  if (const char *s = getenv("MIRT_PIPES")) {
    const int v = atoi(s);
    if (v >= 1 && v <= MAX_PIPES) G.pipes = v;
  }
  if (const char *s = getenv("MIRT_LIVE_16TH")) {
    int v = atoi(s);
    if (v >= 0 && v <= 16) G.live_16th = (uint32_t) v;
  }
  if (const char *s = getenv("MIRT_POOL_GROUP")) {
    int v = atoi(s);
    if (v >= 64 && (v % 64) == 0) G.pool_group = (uint32_t) v;
  }
  if (const char *s = getenv("MIRT_STATIC_16TH")) {
    int v = atoi(s);
    if (v >= 0 && v <= 16) G.static_16th = (uint32_t) v;
  }
  if (const char *s = getenv("MIRT_REFILL_MIN")) {
    int v = atoi(s);
    if (v >= 1 && v <= 64) G.refill_min = (uint32_t) v;
  }
  G.device = device;
  G.ready = true;
  return MIRT_OK;
}

void mirt_shutdown(void) {
  if (!G.ready) return;
  for (int p = 0; p < MAX_PIPES; p++) {
    if (G.pstream[p]) (void) hipStreamSynchronize(G.pstream[p]);
    if (G.pside[p]) (void) hipStreamSynchronize(G.pside[p]);
    if (G.pipe_ev[p]) (void) hipEventDestroy(G.pipe_ev[p]);
    for (int i = 0; i < RING; i++) {
      if (G.ring_ev[p][i]) (void) hipEventDestroy(G.ring_ev[p][i]);
      if (G.fork_ev[p][i]) (void) hipEventDestroy(G.fork_ev[p][i]);
      if (G.join_ev[p][i]) (void) hipEventDestroy(G.join_ev[p][i]);
    }
    if (G.pside[p]) (void) hipStreamDestroy(G.pside[p]);
  }
  if (G.ev0) (void) hipEventDestroy(G.ev0);
  if (G.ev1) (void) hipEventDestroy(G.ev1);
  if (G.ev_in) (void) hipEventDestroy(G.ev_in);
  if (G.h_counts) (void) hipHostFree(G.h_counts);
  if (G.h_tail) (void) hipHostFree(G.h_tail);
  G.h_tail = nullptr;
  if (G.h_film) (void) hipHostFree(G.h_film);
  for (auto e: G.film_ev) if (e) (void) hipEventDestroy(e);
  delete G.film_pool;
  G.film_pool = nullptr;
  for (auto e: G.pool) (void) hipEventDestroy(e);
  G.d_table.release();
  G.d_rgb.release();
  G.ovf.release();
  for (int p = 0; p < MAX_PIPES; p++) if (G.pstream[p]) (void) hipStreamDestroy(G.pstream[p]);
  G = Global();
}

int mirt_scene_create(const float *tri_pos, const float *tri_nrm, const uint32_t *tri_mat, uint32_t n_tris,
                      const mirt_bsdf *mats, uint32_t n_mats, const mirt_light *lights, uint32_t n_lights,
                      uint32_t leaf_size, mirt_scene **out) {
  return mirt_scene_create_prims(tri_pos, tri_nrm, tri_mat, nullptr, n_tris, mats, n_mats, lights, n_lights, leaf_size, out);
}

int mirt_scene_create_prims(const float *tri_pos, const float *tri_nrm, const uint32_t *tri_mat, const uint8_t *prim_type,
                            uint32_t n_tris, const mirt_bsdf *mats, uint32_t n_mats, const mirt_light *lights,
                            uint32_t n_lights, uint32_t leaf_size, mirt_scene **out) {
  return mirt_scene_create_opts(tri_pos, tri_nrm, tri_mat, prim_type, n_tris, mats, n_mats, lights, n_lights, leaf_size, nullptr,
                                out);
}

static int scene_create_impl(const float *tri_pos, const float *tri_nrm, const uint32_t *tri_mat, const uint8_t *prim_type,
                             uint32_t n_tris, const mirt_bsdf *mats, uint32_t n_mats, const mirt_light *lights,
                             uint32_t n_lights, uint32_t leaf_size, const mirt_build_options *build, mirt_scene **out);
static thread_local mirt_scene *g_building = nullptr;   // the scene under construction, for the exception path below

int mirt_scene_create_opts(const float *tri_pos, const float *tri_nrm, const uint32_t *tri_mat, const uint8_t *prim_type,
                           uint32_t n_tris, const mirt_bsdf *mats, uint32_t n_mats, const mirt_light *lights,
                           uint32_t n_lights, uint32_t leaf_size, const mirt_build_options *build, mirt_scene **out) {
  // no C++ exception crosses the C boundary (std::bad_alloc from the staging vectors, std::system_error from the
  // builder threads)
  try {
    return scene_create_impl(tri_pos, tri_nrm, tri_mat, prim_type, n_tris, mats, n_mats, lights, n_lights, leaf_size, build, out);
  } catch (const std::exception &ex) {
    if (g_building) {
      (void) hipDeviceSynchronize();
      g_building->release_all();
      delete g_building;
      g_building = nullptr;
    }
    if (out) *out = nullptr;
    return fail(MIRT_E_LIMIT, std::string("scene creation failed: ") + ex.what());
  }
}

static int scene_create_impl(const float *tri_pos, const float *tri_nrm, const uint32_t *tri_mat, const uint8_t *prim_type,
                             uint32_t n_tris, const mirt_bsdf *mats, uint32_t n_mats, const mirt_light *lights,
                             uint32_t n_lights, uint32_t leaf_size, const mirt_build_options *build, mirt_scene **out) {
  uint32_t builder = 0;
  if (build) {
    builder = build->own_tree_builder;
  } else if (const char *e = getenv("MIRT_BUILDER")) {
    builder = (!strcmp(e, "device") || !strcmp(e, "1")) ? 1u : (!strcmp(e, "ploc") || !strcmp(e, "2")) ? 2u : 0u;
  }
  if (builder > 2) return fail(MIRT_E_ARG, "own_tree_builder must be 0 (host), 1 (device, Morton radix tree) or 2 (device, PLOC)");
  using clk = std::chrono::steady_clock;
  auto ms_since = [](clk::time_point t) { return std::chrono::duration<float, std::milli>(clk::now() - t).count(); };
  if (!G.ready) return fail(MIRT_E_STATE, "mirt_init() has not been called");
  if (!out) return fail(MIRT_E_ARG, "out is null");
  *out = nullptr;
  if (!tri_pos || !tri_nrm || !tri_mat || n_tris == 0) return fail(MIRT_E_ARG, "SCENE NOT SET!");          // renderer.cuh:122
  if (!mats || n_mats == 0) return fail(MIRT_E_ARG, "no materials");
  if (n_lights && !lights) return fail(MIRT_E_ARG, "lights is null");
  for (uint32_t i = 0; i < n_tris; i++) {
    if (tri_mat[i] >= n_mats) return fail(MIRT_E_ARG, "material index out of range");
    if (prim_type && prim_type[i] != 1 && prim_type[i] != 2) return fail(MIRT_E_ARG, "prim_type must be 1 (SPHERE) or 2 (TRIANGLE)");
  }
  auto *s = new mirt_scene();
  g_building = s;
  try {
    auto t0 = clk::now();
    build_reference_bvh2(tri_pos, prim_type, n_tris, leaf_size, s->bvh);
    s->build_ms[0] = ms_since(t0);
    t0 = clk::now();
    collapse_to_wide(s->bvh);
    s->build_ms[1] = ms_since(t0);
  } catch (const std::exception &ex) {
    delete s;
    g_building = nullptr;
    return fail(MIRT_E_LIMIT, ex.what());
  }
  if (s->bvh.depth2 + 2 > (uint32_t) BVH2_STACK) {
    delete s;
    g_building = nullptr;
    return fail(MIRT_E_LIMIT, "reference BVH deeper than the traversal stack (62 levels)");
  }
  s->device = G.device;
  s->n_tris = n_tris;
  s->n_mats = n_mats;
  s->n_lights = n_lights;
  if (lights) s->host_lights.assign(lights, lights + n_lights);

  // triangles in the reference's flat order: (p1, e1, e2) for the intersector, (n1, n2, n3, material, original id) for shading
  std::vector<float> tri((size_t) n_tris * 12), nrm((size_t) n_tris * 12);
  std::atomic<bool> sphere_seen{false};
  parallel_for(n_tris, [&](size_t f0, size_t f1) {
  for (uint32_t f = (uint32_t) f0; f < (uint32_t) f1; f++) {
    const uint32_t o = s->bvh.flat_to_orig[f];
    const float *p = tri_pos + 9 * (size_t) o, *q = tri_nrm + 9 * (size_t) o;
    float *t = &tri[(size_t) f * 12], *m = &nrm[(size_t) f * 12];
    const uint32_t kind = (prim_type && prim_type[o] == 1) ? 1u : 0u;  // word 10 of both records: 0 triangle, 1 sphere
    if (kind) sphere_seen.store(true, std::memory_order_relaxed);
    if (kind) {
      t[0] = p[0]; t[1] = p[1]; t[2] = p[2];                          // origin
      t[3] = p[3];                                                    // r
      t[4] = p[3] * p[3];                                             // r2 = radius * radius (primitives.cuh:188)
      t[5] = t[6] = t[7] = t[8] = 0.f;
      memset(m, 0, 9 * sizeof(float));
      m[0] = p[0]; m[1] = p[1]; m[2] = p[2];
    } else {
      t[0] = p[0]; t[1] = p[1]; t[2] = p[2];
      t[3] = p[3] - p[0]; t[4] = p[4] - p[1]; t[5] = p[5] - p[2];     // e1 = p2 - p1 (primitives.cuh:138)
      t[6] = p[6] - p[0]; t[7] = p[7] - p[1]; t[8] = p[8] - p[2];     // e2 = p3 - p1 (primitives.cuh:139)
      memcpy(m, q, 9 * sizeof(float));
    }
    memcpy(t + 9, &f, 4);                                             // its own flat index (uniform with the fast tree's copy)
    memcpy(t + 10, &kind, 4);
    t[11] = 0.f;
    memcpy(m + 9, &tri_mat[o], 4);
    memcpy(m + 10, &o, 4);
    memcpy(m + 11, &kind, 4);
  }
  });
  s->has_spheres = sphere_seen.load();
  if (s->has_spheres) {
    float rmin = std::numeric_limits<float>::infinity();
    for (uint32_t i = 0; i < n_tris; i++)
      if (prim_type[i] == 1) rmin = std::min(rmin, std::fabs(tri_pos[9 * (size_t) i + 3]));
    s->sph_rmin = std::isfinite(rmin) ? rmin : 0.f;
  }
  // the ordered traversal's own tree over the same primitives, in its own order: binned SAH on the host, or the device builder
  std::vector<float> trif;
  std::vector<float> flat_pos((size_t) n_tris * 9);
  std::vector<uint8_t> flat_type;
  if (prim_type) flat_type.resize(n_tris);
  parallel_for(n_tris, [&](size_t f0, size_t f1) {
    for (size_t f = f0; f < f1; f++) {
      memcpy(&flat_pos[f * 9], tri_pos + 9 * (size_t) s->bvh.flat_to_orig[f], 9 * sizeof(float));
      if (prim_type) flat_type[f] = prim_type[s->bvh.flat_to_orig[f]];
    }
  });
  if (builder >= 1 && n_tris < 2) builder = 0;
  s->own_builder = builder;
  if (builder == 0) {
    auto t0 = clk::now();
    trif.resize((size_t) n_tris * 12);
    try {
      build_fast_tree(flat_pos.data(), prim_type ? flat_type.data() : nullptr, n_tris, s->bvh);
    } catch (const std::exception &) {
      s->fast_ok = false;
      s->bvh.nodes8q.clear();
      s->bvh.depth8 = 0;
    }
    if (s->fast_ok)
      parallel_for(n_tris, [&](size_t j0, size_t j1) {
        for (size_t j = j0; j < j1; j++)
          memcpy(&trif[j * 12], &tri[(size_t) s->bvh.fast_to_flat[j] * 12], 12 * sizeof(float));
      });
    s->n_nodes8 = (uint32_t) (s->bvh.nodes8q.size() / (sizeof(Node8Q) / 4));
    s->depth8 = s->bvh.depth8;
    s->build_ms[2] = ms_since(t0);
  }
  // materials: mirt_bsdf records (9 dwords) padded to three float4
  std::vector<float> mats_padded((size_t) n_mats * 12, 0.f);
  for (uint32_t m = 0; m < n_mats; m++) {
    memcpy(&mats_padded[(size_t) m * 12], &mats[m], sizeof(mirt_bsdf));
    if (mats[m].type == BSDF_MICROFACET) s->has_microfacet = true;
  }
  int rc = MIRT_OK;
  auto up = [&](DevBuf &b, const void *src, size_t bytes) -> bool {
    if (b.ensure(bytes) != hipSuccess) return false;
    return hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  auto t_up = clk::now();
  bool ok = up(s->tris, tri.data(), tri.size() * 4) && up(s->tri_nrm, nrm.data(), nrm.size() * 4) &&
            up(s->nodes4, s->bvh.nodes4.data(), s->bvh.nodes4.size() * sizeof(Node4)) &&
            up(s->nodes2, s->bvh.nodes2.data(), s->bvh.nodes2.size() * sizeof(Node2)) &&
            up(s->mats, mats_padded.data(), mats_padded.size() * 4) &&
            up(s->verts, flat_pos.data(), flat_pos.size() * 4) &&          // exact vertices, flat order (rare path of tri_test_near)
            up(s->big, s->bvh.big.empty() ? (const void *) mats : (const void *) s->bvh.big.data(), s->bvh.big.empty() ? 8 : s->bvh.big.size() * 4) &&
            up(s->lights, lights ? (const void *) lights : (const void *) mats, n_lights ? (size_t) n_lights * sizeof(mirt_light) : 4);
  if (ok && builder == 0)
    ok = up(s->trisf, trif.data(), trif.size() * 4) &&
         up(s->nodes8q, s->bvh.nodes8q.empty() ? (const void *) tri.data() : (const void *) s->bvh.nodes8q.data(),
            s->bvh.nodes8q.empty() ? 16 : s->bvh.nodes8q.size() * 4);
  s->build_ms[7] = ms_since(t_up);
  if (ok && builder >= 1) {
    auto t0 = clk::now();
    DeviceTree dt;
    uint32_t max_leaf = 3;
    if (const char *e = getenv("MIRT_FAST_LEAF")) {
      int v = atoi(e);
      if (v >= 1 && v <= (int) LEAF_MAX) max_leaf = (uint32_t) v;
    }
    const int rc_b = build_fast_tree_device(flat_pos.data(), prim_type ? flat_type.data() : nullptr, s->tris.p, n_tris, max_leaf,
                                            (int) builder, G.stream, dt);
    if (rc_b < 0) {
      rc = fail(MIRT_E_DEVICE, "device tree build failed: " + dt.error);
      s->release_all();
      delete s;
    g_building = nullptr;
      return rc;
    }
    if (rc_b > 0) {                                        // as for the host builder: traversal 0 runs as 1
      s->fast_ok = false;
      ok = s->trisf.ensure(16) == hipSuccess && s->nodes8q.ensure(16) == hipSuccess;
    } else {
      s->nodes8q.p = dt.nodes8q;
      s->nodes8q.bytes = dt.nodes_bytes;
      s->trisf.p = dt.trisf;
      s->trisf.bytes = dt.tris_bytes;
      s->n_nodes8 = dt.n_nodes;
      s->depth8 = dt.depth8;
      for (int k = 0; k < 4; k++) s->build_ms[3 + k] = dt.ms[k];
    }
    s->build_ms[2] = ms_since(t0);
  }
  if (ok) {
    s->trace_threads = (uint32_t) (G.n_cu * G.blocks_per_cu * TRACE_BLOCK);
    // Worst case of a nearest-first walk: every level leaves all its other children waiting (3 per level of the 4-wide tree,
    // 7 of the 8-wide one).  Real rays stay far below that, and the region costs 8 B x resident lanes x launches in flight
    // per level (~25 MB), so it is capped: a ray that outgrows LDS part + region is handed to the reference walk (k_redo),
    // whose result is the reference's by construction.  MIRT_OVF_CAP: levels beyond the LDS part (default 24).
    const int ovf_cap = getenv("MIRT_OVF_CAP") ? std::max(1, atoi(getenv("MIRT_OVF_CAP"))) : 24;
    int need = std::max(3 * ((int) s->bvh.depth4 + 2), 7 * ((int) s->depth8 + 2)) - LDS_STACK + 1;
    need = std::min(need, ovf_cap);
    s->ovf_depth = (uint32_t) (need > 1 ? need : 1);
    // one region per launch in flight: the closest-hit and the shadow launch of an iteration run concurrently, in each pipe.
    // The buffer belongs to the library context, not the scene: frame loops that rebuild the scene keep it.
    ok = G.ovf.ensure(2 * MAX_PIPES * (size_t) s->ovf_depth * s->trace_threads * sizeof(uint2)) == hipSuccess;
  }
  if (!ok) {
    rc = fail(MIRT_E_DEVICE, std::string("scene upload failed: ") + hipGetErrorString(hipGetLastError()));
    s->release_all();
    delete s;
    g_building = nullptr;
    return rc;
  }
  g_building = nullptr;
  *out = s;
  return MIRT_OK;
}

}  // extern "C"

namespace mirt {
// A replica of `src` on the calling thread's device (mirt_group_set_scene: the first member builds the scene - host trees,
// packed records, uploads - the others copy its device buffers instead of repeating the host build N times at once).  `src`
// may live on another device: the copies are plain unified-address copies.  Nothing may be using `src` meanwhile.
int scene_clone(const mirt_scene *src, mirt_scene **out) {
  if (!G.ready) return fail(MIRT_E_STATE, "mirt_init() has not been called");
  if (!src || !out) return fail(MIRT_E_ARG, "null argument");
  *out = nullptr;
  mirt_scene *s = nullptr;
  try {
    s = new mirt_scene();
    s->bvh = src->bvh;                                   // host-side arrays behind mirt_scene_info / _bvh2 / _own_tree
  } catch (const std::exception &ex) {
    delete s;
    return fail(MIRT_E_LIMIT, std::string("scene replica failed: ") + ex.what());
  }
  s->n_tris = src->n_tris; s->n_mats = src->n_mats; s->n_lights = src->n_lights;
  s->has_microfacet = src->has_microfacet; s->has_spheres = src->has_spheres; s->sph_rmin = src->sph_rmin;
  s->fast_ok = src->fast_ok; s->own_builder = src->own_builder; s->n_nodes8 = src->n_nodes8; s->depth8 = src->depth8;
  memcpy(s->build_ms, src->build_ms, sizeof s->build_ms);
  s->ovf_depth = src->ovf_depth;
  s->host_lights = src->host_lights;
  s->device = G.device;
  s->trace_threads = (uint32_t) (G.n_cu * G.blocks_per_cu * TRACE_BLOCK);
  const DevBuf mirt_scene::*data[] = {&mirt_scene::nodes4, &mirt_scene::nodes2, &mirt_scene::tris, &mirt_scene::tri_nrm, &mirt_scene::mats,
                                      &mirt_scene::lights, &mirt_scene::big, &mirt_scene::trisf, &mirt_scene::nodes8q, &mirt_scene::verts};
  bool ok = true;
  for (auto m: data) {
    const DevBuf &from = src->*m;
    DevBuf &to = const_cast<DevBuf &>(s->*m);
    if (!from.p || !from.bytes) continue;
    ok = ok && to.ensure(from.bytes) == hipSuccess && hipMemcpy(to.p, from.p, from.bytes, hipMemcpyDefault) == hipSuccess;
  }
  ok = ok && G.ovf.ensure(2 * MAX_PIPES * (size_t) s->ovf_depth * s->trace_threads * sizeof(uint2)) == hipSuccess;
  if (!ok) {
    const int rc = fail(MIRT_E_DEVICE, std::string("scene replica failed: ") + hipGetErrorString(hipGetLastError()));
    s->release_all();
    delete s;
    return rc;
  }
  *out = s;
  return MIRT_OK;
}
}  // namespace mirt

extern "C" {

void mirt_scene_destroy(mirt_scene *s) {
  if (!s) return;
  if (G.ready) (void) hipDeviceSynchronize();                            // every pipe and side stream
  s->release_all();
  delete s;
}

int mirt_scene_info(const mirt_scene *s, uint32_t info[8]) {
  if (!s || !info) return fail(MIRT_E_ARG, "null argument");
  info[0] = (uint32_t) s->bvh.nodes2.size();
  info[1] = s->bvh.depth2;
  info[2] = (uint32_t) s->bvh.nodes4.size();
  info[3] = s->bvh.depth4;
  info[4] = s->n_tris;
  info[5] = s->n_mats;
  info[6] = s->n_lights;
  info[7] = (uint32_t) (s->device_bytes() / 1024);
  return MIRT_OK;
}

int mirt_scene_build_times(const mirt_scene *s, float ms[8]) {
  if (!s || !ms) return fail(MIRT_E_ARG, "null argument");
  memcpy(ms, s->build_ms, sizeof s->build_ms);
  return MIRT_OK;
}

int mirt_scene_own_tree(const mirt_scene *s, uint32_t counts[4], uint32_t *nodes8q_out, uint32_t *prims_out) {
  if (!s || !counts) return fail(MIRT_E_ARG, "null argument");
  counts[0] = s->fast_ok ? s->n_nodes8 : 0;
  counts[1] = s->depth8;
  counts[2] = s->n_tris;
  counts[3] = s->own_builder;
  if (!s->fast_ok) return MIRT_OK;
  if (nodes8q_out) HIPCHK(hipMemcpy(nodes8q_out, s->nodes8q.p, (size_t) s->n_nodes8 * sizeof(Node8Q), hipMemcpyDeviceToHost));
  if (prims_out) HIPCHK(hipMemcpy(prims_out, s->trisf.p, (size_t) s->n_tris * 48, hipMemcpyDeviceToHost));
  return MIRT_OK;
}

int mirt_scene_bvh2(const mirt_scene *s, uint32_t *nodes_out, uint32_t *flat_to_orig) {
  if (!s) return fail(MIRT_E_ARG, "null scene");
  if (nodes_out) memcpy(nodes_out, s->bvh.nodes2.data(), s->bvh.nodes2.size() * sizeof(Node2));
  if (flat_to_orig) memcpy(flat_to_orig, s->bvh.flat_to_orig.data(), s->bvh.flat_to_orig.size() * 4);
  return MIRT_OK;
}

// host restatement of the Camera ctor (scene/camera.cuh:19-36); same float operations as linalg/Vector3f.cuh
int mirt_camera_look_at(const float from[3], const float at[3], const float up[3], float hFov, float vFov, float nClip,
                        float fClip, float aperture, float focal_distance, mirt_camera *out) {
  if (!from || !at || !up || !out) return fail(MIRT_E_ARG, "null argument");
  auto sub = [](const float *a, const float *b, float *r) { for (int i = 0; i < 3; i++) r[i] = a[i] - b[i]; };
  auto cross = [](const float *a, const float *b, float *r) {
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
  };
  auto norm = [](const float *a) { return (float) std::sqrt((double) (a[0] * a[0] + a[1] * a[1] + a[2] * a[2])); };
  auto unit = [&](const float *a, float *r) {
    float rn = (float) (1. / (double) norm(a));
    for (int i = 0; i < 3; i++) r[i] = a[i] * rn;
  };
  float d[3], x[3], y[3], t[3];
  sub(from, at, d);
  if (focal_distance == 0) focal_distance = norm(d);                     // renderer.cuh:97-99
  cross(up, d, t);
  unit(t, x);
  cross(d, x, t);
  unit(t, y);
  for (int i = 0; i < 3; i++) {
    out->origin[i] = from[i];
    out->c2w[i] = x[i];
    out->c2w[3 + i] = y[i];
  }
  unit(d, out->c2w + 6);
  out->hFov = hFov; out->vFov = vFov; out->nClip = nClip; out->fClip = fClip;
  out->aperture = aperture; out->focal_distance = focal_distance;
  return MIRT_OK;
}

}  // extern "C"

namespace {

DCamera to_device_camera(const mirt_camera &c) {
  DCamera d;
  d.origin = {c.origin[0], c.origin[1], c.origin[2]};
  d.c2w.c0 = {c.c2w[0], c.c2w[1], c.c2w[2]};
  d.c2w.c1 = {c.c2w[3], c.c2w[4], c.c2w[5]};
  d.c2w.c2 = {c.c2w[6], c.c2w[7], c.c2w[8]};
  const float PI_F = 3.14159265358979323f;
  d.tan_h = tanf(c.hFov * PI_F / 360);                                   // camera.cuh:39, host libm like the oracle
  d.tan_v = tanf(c.vFov * PI_F / 360);
  d.nClip = c.nClip; d.fClip = c.fClip; d.aperture = c.aperture; d.focal_distance = c.focal_distance;
  return d;
}

struct Partition {
  uint32_t tiles_x, tiles_y, rank, world, n_slots;
};
Partition make_partition(uint32_t w, uint32_t h, const mirt_options &o) {
  Partition p;
  p.tiles_x = (w + 7) / 8;
  p.tiles_y = (h + 7) / 8;
  p.world = o.tile_world ? o.tile_world : 1;
  p.rank = o.tile_rank;
  const uint64_t total = (uint64_t) p.tiles_x * p.tiles_y;
  const uint64_t local = total > p.rank ? (total - p.rank + p.world - 1) / p.world : 0;
  p.n_slots = (uint32_t) (local * 64);
  return p;
}

// The traversal-stack overflow regions live in the library context of the CALLING thread (Global::ovf), a scene's ovf_depth and
// trace_threads are fixed when it is created: a scene used from another host thread's context, or after mirt_init() moved this
// thread to another device (which releases the context's buffers), must not find a missing or smaller buffer - the kernels
// index it unchecked.  Called by every entry point that launches a traversal with G.ovf.
int ensure_overflow_region(const mirt_scene *s) {
  if (s->device != G.device)
    return fail(MIRT_E_STATE, "the scene was created on device " + std::to_string(s->device) + ", this thread's context is on device " +
                                  std::to_string(G.device) + " (mirt_init): a scene is used on the device it was created on");
  HIPCHK(G.ovf.ensure(2 * MAX_PIPES * (size_t) s->ovf_depth * s->trace_threads * sizeof(uint2)));
  return MIRT_OK;
}

int ensure_work(mirt_scene *s, uint32_t n_slots, uint32_t K) {
  if (int rc = ensure_overflow_region(s)) return rc;
  const size_t n = n_slots ? n_slots : 64, kn = (size_t) (K ? K : 1) * n;
  HIPCHK(s->st.ensure(n * 16));
  HIPCHK(s->mask.ensure(n * sizeof(rec3)));
  HIPCHK(s->total.ensure(n * 16));
  HIPCHK(s->accum.ensure(n * 16));
  HIPCHK(s->dmask.ensure(n * sizeof(rec3)));
  HIPCHK(s->ray_o.ensure(n * 16));
  HIPCHK(s->ray_d.ensure(n * 16));
  HIPCHK(s->hit.ensure(n * 16));
  HIPCHK(s->sh_o.ensure((MIRT_SHARED_ORIGIN ? n : kn) * 16));     // one origin per slot (mirt_trace: per ray, K = 1)
  HIPCHK(s->sh_d.ensure(kn * 16));
  HIPCHK(s->sh_c.ensure(kn * sizeof(rec3)));
  HIPCHK(s->sh_occ.ensure(kn * 4));
  HIPCHK(s->ext_queue.ensure(n * 4));
  HIPCHK(s->sh_queue.ensure(kn * 4));
  HIPCHK(s->redo_q[0].ensure(n * 4));
  HIPCHK(s->redo_q[1].ensure(kn * 4));
  HIPCHK(s->mf_queue.ensure(n * 4));
  HIPCHK(s->live[0].ensure(n * 4));
  HIPCHK(s->live[1].ensure(n * 4));
  HIPCHK(s->ctrl.ensure(CTRL_BLOCKS * MAX_PIPES * CTRL_WORDS * 4));    // per pipe three blocks in rotation
  HIPCHK(s->ray_stats.ensure(RAY_STATS_BYTES + 256 + 4 * ((size_t) n_slots / 64 + 1)));   // + the share pool's counters (at most one per tile)
  return MIRT_OK;
}

// `pipe` selects the control block, statistics and stack-overflow regions; `off` is the pipe's first slot (shadow arrays
// hold K entries per slot, so their offset is K * off), `n_slots` the number of its slots.
TraceArgs common_args(mirt_scene *s, uint32_t traversal) {
  TraceArgs t;
  memset(&t, 0, sizeof t);
  t.nodes4 = s->nodes4.as<float4>();
  t.nodes2 = s->nodes2.as<float4>();
  t.tris = traversal == 0 ? s->trisf.as<float4>() : s->tris.as<float4>();
  t.tris2 = s->tris.as<float4>();
  t.require_unit = traversal == 0 ? 1 : 0;
  t.nodes8 = s->nodes8q.p;
  t.verts = s->verts.as<float>();
  t.spheres = s->has_spheres ? 1 : 0;
  t.sph_rmin = s->sph_rmin;
  t.refill_min = G.refill_min;
  t.static_16th = G.static_16th;
  t.big = s->big.as<uint2>();
  t.ovf_depth = s->ovf_depth;
  return t;
}
TraceArgs closest_args(mirt_scene *s, uint32_t n_slots, uint32_t traversal, int pipe = 0, size_t off = 0, int par = 0 /* control block */) {
  TraceArgs t = common_args(s, traversal);
  uint32_t *ctrl = s->ctrl.as<uint32_t>() + (size_t) (CTRL_BLOCKS * pipe + par) * CTRL_WORDS;
  unsigned long long *rs = s->ray_stats.as<unsigned long long>() + (size_t) pipe * 16;
  t.ray_o = s->ray_o.as<float4>() + off;
  t.ray_d = s->ray_d.as<float4>() + off;
  t.queue = s->ext_queue.as<uint32_t>() + off;
  t.qcount = ctrl + CTRL_QCOUNT;
  t.qc_front = QC_EXT_FRONT;
  t.qc_back = QC_EXT_BACK;
  t.qcap = n_slots;
  t.hit = s->hit.as<float4>() + off;
  t.work_head = ctrl + CTRL_WORK + 0;
  t.ovf = G.ovf.as<uint2>() + (size_t) (2 * pipe) * s->ovf_depth * s->trace_threads;
  t.redo_queue = s->redo_q[0].as<uint32_t>() + off;
  t.redo_count = ctrl + CTRL_REDO + 0;
  t.error_flag = ctrl + CTRL_ERR;
  t.redo_stat = rs + 3;
  t.count_stat = rs + 4;
  return t;
}
// (o_slots != 0: the shadow rays of a render - one origin per slot, at sh_o + off, entry k * o_slots + slot; 0: mirt_trace's
// occlusion rays, one origin per ray)
TraceArgs shadow_args(mirt_scene *s, uint32_t traversal, int pipe = 0, size_t koff = 0, int par = 0, size_t off = 0, uint32_t o_slots = 0) {
  TraceArgs t = common_args(s, traversal);
  uint32_t *ctrl = s->ctrl.as<uint32_t>() + (size_t) (CTRL_BLOCKS * pipe + par) * CTRL_WORDS;
  unsigned long long *rs = s->ray_stats.as<unsigned long long>() + (size_t) pipe * 16;
  t.ray_o = s->sh_o.as<float4>() + (o_slots ? off : koff);
  t.o_slots = o_slots;

  t.ray_d = s->sh_d.as<float4>() + koff;
  t.queue = s->sh_queue.as<uint32_t>() + koff;
  t.qcount = ctrl + CTRL_QCOUNT;
  t.qc_front = QC_SHADOW;
  t.qc_back = -1;
  t.occ = s->sh_occ.as<uint32_t>() + koff;
  t.work_head = ctrl + CTRL_WORK + WORK_PARTS * CTR_STRIDE;
  t.ovf = G.ovf.as<uint2>() + (size_t) (2 * pipe + 1) * s->ovf_depth * s->trace_threads;
  t.redo_queue = s->redo_q[1].as<uint32_t>() + koff;
  t.redo_count = ctrl + CTRL_REDO + CTR_STRIDE;
  t.error_flag = ctrl + CTRL_ERR;
  t.redo_stat = rs + 3;
  t.count_stat = rs + 6;
  return t;
}

// One trace launch plus the reference walk over the rays it set aside, on `stream`.
int run_trace(mirt_scene *s, TraceArgs t, bool any, uint32_t traversal, std::vector<Timed> *timed, bool count,
              hipStream_t stream, uint32_t blocks_per_cu = 0, uint32_t grid_cap = 0) {
  uint32_t grid = s->trace_threads / TRACE_BLOCK;                       // what the overflow regions are sized for
  if (blocks_per_cu && blocks_per_cu * (uint32_t) G.n_cu < grid) grid = blocks_per_cu * (uint32_t) G.n_cu;
  if (grid_cap && grid_cap < grid) grid = grid_cap;
  Timed tm{any ? 2 : 1, nullptr, nullptr};
  if (timed) {
    tm.a = G.pool_next();
    tm.b = G.pool_next();
    if (!tm.a || !tm.b) return fail(MIRT_E_DEVICE, "could not create timing events");
    HIPCHK(hipEventRecord(tm.a, stream));
  }
  t.count = count ? 1 : 0;
  HIPCHK(launch_trace(t, any, traversal, grid, stream));
  if (timed) {
    HIPCHK(hipEventRecord(tm.b, stream));
    timed->push_back(tm);
  }
#ifdef MIRT_TIMING_EXPERIMENTS                               // tools/redo_bound.sh builds a variant with it: WRONG results, timing only
  static const int skip_redo = getenv("MIRT_SKIP_REDO") ? atoi(getenv("MIRT_SKIP_REDO")) : 0;
#else
  constexpr int skip_redo = 0;                               // the shipped library always runs the reference walk
#endif
  if (traversal != 2 && !((skip_redo & 1) && any) && !((skip_redo & 2) && !any)) {
    t.redo = 1;
    HIPCHK(launch_trace(t, any, 2, 256, stream));
  }
  return MIRT_OK;
}

int render_impl(mirt_scene *s, const mirt_camera *camera, const mirt_params *prm, const mirt_options *opts_in,
                uint32_t w, uint32_t h, void *accum_dev, mirt_stats *stats) {
  if (!G.ready) return fail(MIRT_E_STATE, "mirt_init() has not been called");
  if (!s) return fail(MIRT_E_STATE, "SCENE NOT SET!");                     // renderer.cuh:122
  if (!camera) return fail(MIRT_E_STATE, "CAMERA NOT SET!");               // renderer.cuh:118
  if (s->n_lights == 0) return fail(MIRT_E_STATE, "LIGHTS NOT SET!");      // renderer.cuh:126
  if (w == 0 || h == 0) return fail(MIRT_E_STATE, "SAMPLE BUFFER EMPTY");  // renderer.cuh:130
  if (!prm || !accum_dev) return fail(MIRT_E_ARG, "null argument");
  if (prm->samples_per_batch == 0) return fail(MIRT_E_ARG, "samples_per_batch must be > 0 (the reference divides by it)");
  if ((uint64_t) w * h >= (1ull << 31)) return fail(MIRT_E_LIMIT, "image larger than 2^31 pixels");
  mirt_options o;
  memset(&o, 0, sizeof o);
  if (opts_in) o = *opts_in;
  if (o.rng_mode > 1 || o.traversal > 2) return fail(MIRT_E_ARG, "bad rng_mode / traversal");
  if (o.traversal == 0 && !s->fast_ok) o.traversal = 1;
  if (o.tile_world > 1 && o.tile_rank >= o.tile_world) return fail(MIRT_E_ARG, "tile_rank >= tile_world");
  uint32_t streams = o.streams > 1 ? o.streams : 1;
  const bool sliced = o.sample_count != 0 || o.sample_begin != 0 || streams > 1 || o.shares > 1 || (o.flags & 4u);
  if (sliced && o.rng_mode != 1)
    return fail(MIRT_E_ARG, "spp slices and sample streams need rng_mode 1 (the xorshift stream is sequential per pixel)");
  if (streams > 16) return fail(MIRT_E_ARG, "at most 16 sample streams");
  const uint32_t spp_here = o.sample_count ? o.sample_count : prm->samples_per_pixel;
  if (streams > spp_here) streams = spp_here ? spp_here : 1;
  // Share pool: W = o.shares planes per pixel, `streams` slots per pixel slot that take (share, pixel) items dynamically.
  // W <= streams is plain sample streams.  Equal shares only: the kernel prices every item at spp / W samples.
  uint32_t shares = streams;
  if (o.shares > streams) {
    if (o.rng_mode != 1) return fail(MIRT_E_ARG, "the share pool needs rng_mode 1 (the xorshift stream is sequential per pixel)");
    if (o.shares > spp_here) return fail(MIRT_E_ARG, "more shares than samples per pixel");
    if (spp_here % o.shares != 0) return fail(MIRT_E_ARG, "shares must divide the number of samples per pixel");
    if (o.shares > 4096) return fail(MIRT_E_ARG, "at most 4096 shares");
    shares = o.shares;
  }
  const bool pool = shares > streams;
  const uint64_t K64 = (uint64_t) s->n_lights * prm->samples_per_light;
  if (K64 > 0xFFFF) return fail(MIRT_E_LIMIT, "more than 65535 shadow rays per hit");
  const uint32_t K = (uint32_t) K64;
  const Partition part = make_partition(w, h, o);
  const uint64_t total_slots = (uint64_t) part.n_slots * streams;
  if (total_slots * (K ? K : 1) >= (1ull << 32)) return fail(MIRT_E_LIMIT, "shadow queue beyond 2^32 entries");
  if ((uint64_t) part.n_slots * shares >= (1ull << 32) - 1) return fail(MIRT_E_LIMIT, "more than 2^32 share items");
  auto wall0 = std::chrono::steady_clock::now();

  int rc = ensure_work(s, (uint32_t) total_slots, K);
  if (rc) return rc;
  // progressive pass (flags bit 2): the samples are added to what accum_dev already holds, through the plane buffer
  const bool accumulate = (o.flags & 4u) != 0;
  const bool via_planes = shares > 1 || accumulate;
  if (accumulate && o.rng_mode != 1) return fail(MIRT_E_ARG, "progressive passes need rng_mode 1");
  if (via_planes) HIPCHK(s->planes.ensure((size_t) shares * w * h * 16));

  RenderArgs a;
  memset(&a, 0, sizeof a);
  a.n_slots = (uint32_t) total_slots; a.n_pixel_slots = part.n_slots; a.streams = streams;
  if (total_slots >= (1ull << 31) || (uint64_t) part.n_slots * shares >= (1ull << 31))
    return fail(MIRT_E_LIMIT, "more than 2^31 path slots or share items");             // (fastdiv.hpp: dividends below 2^31)
  a.shares = shares; a.pool = pool ? 1u : 0u;
  a.fd_pixel_slots = make_fastdiv(part.n_slots); a.fd_tiles_x = make_fastdiv(part.tiles_x);
  a.share_len = (o.sample_count ? o.sample_count : prm->samples_per_pixel) / (shares ? shares : 1u);
  if (o.pool_group % 64u) return fail(MIRT_E_ARG, "pool_group must be a multiple of 64 (whole 8x8 tiles)");
  a.pool_group = o.pool_group ? o.pool_group : G.pool_group;
  a.fd_pool_group = make_fastdiv(a.pool_group);
  a.pool_next = (uint32_t *) ((char *) s->ray_stats.p + RAY_STATS_BYTES + 256);
  const size_t pool_groups = pool ? ((size_t) part.n_slots + a.pool_group - 1) / a.pool_group : 0;
  a.width = w; a.height = h;
  a.tiles_x = part.tiles_x; a.tiles_y = part.tiles_y; a.tile_rank = part.rank; a.tile_world = part.world;
  a.rng_mode = o.rng_mode; a.seed = o.seed; a.sample_begin = o.sample_begin;
  a.spp = o.sample_count ? o.sample_count : prm->samples_per_pixel;
  a.spl = prm->samples_per_light; a.batch = prm->samples_per_batch; a.max_depth = prm->max_ray_depth;
  a.adaptive = sliced ? 0 : 1;
  a.glass_int_abs = o.glass_int_abs; a.n_lights = s->n_lights; a.tol = prm->max_tolerance;
  a.cam = to_device_camera(*camera);
  a.st = s->st.as<uint4>(); a.mask = s->mask.as<rec3>(); a.total = s->total.as<float4>();
  a.accum = s->accum.as<float4>(); a.dmask = s->dmask.as<rec3>();
  a.ray_o = s->ray_o.as<float4>(); a.ray_d = s->ray_d.as<float4>(); a.hit = s->hit.as<float4>();
  a.sh_o = s->sh_o.as<float4>(); a.sh_d = s->sh_d.as<float4>(); a.sh_c = s->sh_c.as<rec3>();
  a.sh_occ = s->sh_occ.as<uint32_t>();
  a.ext_queue = s->ext_queue.as<uint32_t>(); a.sh_queue = s->sh_queue.as<uint32_t>();
  uint32_t *ctrl = s->ctrl.as<uint32_t>();
  a.qcount = ctrl + CTRL_QCOUNT;
  a.ray_stats = s->ray_stats.as<unsigned long long>();
  a.out_accum = via_planes ? s->planes.as<float4>() : (float4 *) accum_dev;
  a.tri_nrm = s->tri_nrm.as<float4>(); a.mats = s->mats.as<float4>(); a.lights = s->lights.as<float>();
  a.mf_queue = s->mf_queue.as<uint32_t>(); a.mf_count = ctrl + CTRL_MF;
  a.live_16th = G.live_16th;
  a.cull_zero = ((o.flags & 16u) && K >= 1 && K <= 32) ? 1u : 0u;

  // shadow entry k = light * samples_per_light + j; entries towards a directional light need no direction record
  if (K && s->const_dir_spl != prm->samples_per_light) {
    s->const_k = NO_CONST_K;
    for (uint32_t k = 0; k < K && s->const_k == NO_CONST_K; k++) {
      const mirt_light &L = s->host_lights[k / prm->samples_per_light];
      if (L.type == LIGHT_DIRECTIONAL) {
        s->const_dir[0] = L.p[3]; s->const_dir[1] = L.p[4]; s->const_dir[2] = L.p[5];
        s->const_dir[3] = std::numeric_limits<float>::infinity();      // dist - EPS_F with dist = INF (lights.cuh:96-101, raytracer.cuh:40)
        s->const_k = k;
      }
    }
    s->const_dir_spl = prm->samples_per_light;
  }
  if (getenv("MIRT_NO_CONST_DIR")) s->const_k = NO_CONST_K;          // (A/B: k_logic writes every shadow ray's direction record)
  a.const_k = K ? s->const_k : NO_CONST_K;

  const bool profile = stats && (o.flags & 1u);
  const bool profile_closest = stats && (o.flags & 9u);                  // bit 3: time the closest-hit launches only
  const bool counting = stats && (o.flags & 2u);
  double ms_kind[3] = {0, 0, 0};
  uint32_t n_kind[3] = {0, 0, 0};
  std::vector<Timed> timed;
  G.pool_used = 0;
  const hipEvent_t evA = G.ev0, evB = G.ev1;
  // Whatever fails below, nothing of this call is left running on any of the library's streams when it returns.
  struct Drain {
    bool armed = true;
    ~Drain() { if (armed) (void) hipDeviceSynchronize(); }
  } drain;
  // accum_dev belongs to the caller, who may have queued work on it: everything here is ordered after what the device's
  // NULL stream holds at this point (torch's default stream); see mirt.h for the contract.
  HIPCHK(hipEventRecord(G.ev_in, nullptr));
  HIPCHK(hipStreamWaitEvent(G.stream, G.ev_in, 0));

  // What has to start at zero.  Every pixel of this partition is written exactly once per sample stream when its slot
  // finishes, and k_sum_planes overwrites the whole accumulation buffer: the buffers only need clearing for the pixels that
  // belong to other members of a tile partition (and for an empty partition).  At 8 streams that is 300 MB less to write.
  const bool foreign_pixels = part.world > 1 || part.n_slots == 0;
  if (!accumulate && foreign_pixels)
    HIPCHK(hipMemsetAsync(accum_dev, 0, (size_t) w * h * 16, G.stream));
  if (via_planes && foreign_pixels) HIPCHK(hipMemsetAsync(s->planes.p, 0, (size_t) shares * w * h * 16, G.stream));
  HIPCHK(hipMemsetAsync(s->ray_stats.p, 0, RAY_STATS_BYTES + 256 + 4 * pool_groups, G.stream));
  HIPCHK(hipMemsetAsync(s->ctrl.p, 0, CTRL_BLOCKS * MAX_PIPES * CTRL_WORDS * 4, G.stream));
  HIPCHK(hipEventRecord(evA, G.stream));

  // Pipes: with at least two sample streams the slots are split into two halves that run as independent wavefront loops
  // on their own streams (see Global).  One pipe otherwise.
  int n_pipes = (streams >= 2 && !G.serial && !G.one_pipe && part.n_slots) ? G.pipes : 1;
  while (n_pipes > 1 && (uint32_t) n_pipes > streams) n_pipes /= 2;
  // with two pipes four persistent launches share the GPU: 4 workgroups per CU each measured best (tools/sweep_env.sh)
  const uint32_t pipe_blocks = (n_pipes >= 2 && !G.blocks_forced) ? (n_pipes == 2 ? 4u : 3u) : 0u;
  // (the closest-hit launches are what the frame waits for: the shadow launches get by with one workgroup per CU less -
  // except under the share pool, whose launches stay full until the frame's end: tools/shadow_blocks.sh, conference* 64 spp
  // 118.8 -> 116.0 ms with 4, the other workloads and the smaller shares unchanged)
  uint32_t shadow_blocks = pipe_blocks ? (pool ? pipe_blocks : pipe_blocks - 1) : 0u;
  if (const char *e = getenv("MIRT_SHADOW_BLOCKS")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 16) shadow_blocks = (uint32_t) v;
  }
  // Every pipe has three control blocks and two live-slot lists.  Iteration i counts into block i % 3, reads the live count
  // of block (i - 1) % 3 and the list iteration i - 1 wrote, and its k_logic zeroes block (i + 1) % 3 - last used by
  // iteration i - 2, whose launches have all finished - for the next iteration: no memset dispatch per iteration.
  struct Pipe {
    RenderArgs a;                                   // the pipe's share of the slot arrays; per-iteration fields set at launch
    TraceArgs tc[CTRL_BLOCKS], ts[CTRL_BLOCKS];
    uint32_t *ctrl[CTRL_BLOCKS];
    size_t off = 0;
    uint64_t it = 0, accounted = 0;
    uint32_t stamp[RING] = {};
    bool done = false, tail = false;
  } pipe[MAX_PIPES];
  for (int p = 0; p < n_pipes; p++) {
    const uint32_t s_first = streams * (uint32_t) p / (uint32_t) n_pipes;          // a contiguous range of the sample streams
    const uint32_t s_count = streams * (uint32_t) (p + 1) / (uint32_t) n_pipes - s_first;
    const size_t off = (size_t) s_first * part.n_slots, koff = off * (K ? K : 1);
    Pipe &pp = pipe[p];
    pp.off = off;
    RenderArgs &pa = pp.a;
    pa = a;
    pa.n_slots = part.n_slots * s_count;
    pa.stream_first = s_first;
    pa.st += off; pa.mask += off; pa.total += off; pa.accum += off; pa.dmask += off;
    pa.ray_o += off; pa.ray_d += off; pa.hit += off;
    pa.sh_o += MIRT_SHARED_ORIGIN ? off : koff; pa.sh_d += koff; pa.sh_c += koff; pa.sh_occ += koff;
    pa.ext_queue += off; pa.sh_queue += koff; pa.mf_queue += off;
    pa.ray_stats = a.ray_stats + (size_t) p * 16;
    for (int b = 0; b < CTRL_BLOCKS; b++) {
      pp.ctrl[b] = ctrl + (size_t) (CTRL_BLOCKS * p + b) * CTRL_WORDS;
      pp.tc[b] = closest_args(s, pa.n_slots, o.traversal, p, off, b);
      pp.ts[b] = shadow_args(s, o.traversal, p, koff, b, off, MIRT_SHARED_ORIGIN ? pa.n_slots : 0u);
    }
  }
  uint32_t iters = 0;
  uint64_t n_camera = 0, n_bounced = 0, n_shadow_rays = 0, n_live = 0, n_scanned = 0;
  if (part.n_slots) {
    // Entry const_k of every hit goes to a directional light: its direction records never change, so they are written once,
    // here, and k_logic skips them (RenderArgs::const_k).  Again whenever the buffer, the slot layout or the light changed.
    if (a.const_k != NO_CONST_K) {
      mirt_scene::FillSig sig;
      sig.buf = s->sh_d.p; sig.pixel_slots = part.n_slots; sig.streams = streams; sig.K = K; sig.pipes = (uint32_t) n_pipes; sig.k = a.const_k;
      memcpy(sig.dir, s->const_dir, sizeof sig.dir);
      if (memcmp(&sig, &s->fill_sig, sizeof sig) != 0) {
        for (int p = 0; p < n_pipes; p++)
          HIPCHK(launch_fill4(pipe[p].a.sh_d + (size_t) a.const_k * pipe[p].a.n_slots, pipe[p].a.n_slots,
                              make_float4(s->const_dir[0], s->const_dir[1], s->const_dir[2], s->const_dir[3]), G.stream));
        s->fill_sig = sig;
      }
    }
    if (n_pipes >= 2) {                                                  // the other pipes start after the clears above
      HIPCHK(hipEventRecord(G.pipe_ev[0], G.stream));
      for (int p = 1; p < n_pipes; p++) HIPCHK(hipStreamWaitEvent(G.pstream[p], G.pipe_ev[0], 0));
    }
    // (share pool: a slot renders items until the frame runs out of them - at most (1/streams + 1/shares) of a pixel's samples)
    const uint64_t max_iters = (uint64_t) a.spp * ((uint64_t) a.max_depth + 3) * (pool ? 2u : 1u) + 8;
    bool all_done = false;
    while (!all_done) {
      all_done = true;
      for (int p = 0; p < n_pipes; p++) {
        Pipe &pp = pipe[p];
        if (pp.done) continue;
        all_done = false;
        hipStream_t main = G.pstream[p], side = G.pside[p];
        volatile uint32_t *hcounts = G.h_counts + (size_t) p * RING * HSLOT;
        const uint64_t it = pp.it++;
        if (it >= max_iters + RING) {
          return fail(MIRT_E_DEVICE, "wavefront loop did not terminate");
        }
        const int blk = (int) (it % CTRL_BLOCKS), prev = (int) ((it + CTRL_BLOCKS - 1) % CTRL_BLOCKS),
                  next = (int) ((it + 1) % CTRL_BLOCKS), par = (int) (it & 1);
        const int slot = (int) (it % RING);
        RenderArgs la = pp.a;
        la.qcount = pp.ctrl[blk] + CTRL_QCOUNT;
        la.mf_count = pp.ctrl[blk] + CTRL_MF;
        la.live_out = s->live[par].as<uint32_t>() + pp.off;
        la.live_in = it ? s->live[par ^ 1].as<uint32_t>() + pp.off : nullptr;      // first iteration: no list yet, all slots
        la.live_in_count = pp.ctrl[prev] + CTRL_QCOUNT + QC_LIVE;
        la.clear_next = pp.ctrl[next];
        la.clear_words = CTRL_ERR;                                       // queue counters, work heads, redo counters
        la.first = it == 0 ? 1u : 0u;                                    // the first k_logic initialises the slots (no k_init launch)
        if (++G.seq == 0) G.seq = 1;
        pp.stamp[slot] = G.seq;
        Timed tl{0, nullptr, nullptr};
        if (profile) {
          tl.a = G.pool_next();
          tl.b = G.pool_next();
          if (!tl.a || !tl.b) return fail(MIRT_E_DEVICE, "could not create timing events");
          HIPCHK(hipEventRecord(tl.a, main));
        }
        HIPCHK(launch_logic(la, false, main));
        if (s->has_microfacet) HIPCHK(launch_logic(la, true, main));
        if (profile) {
          HIPCHK(hipEventRecord(tl.b, main));
          timed.push_back(tl);
        }
        // The two traversals of an iteration are independent: the shadow rays go to the side stream, so the tail of one
        // launch (a few long rays, the GPU mostly idle) is filled by the other.  MIRT_SERIAL=1 keeps one stream.
        const bool fork = K != 0 && !G.serial;
        if (fork) {
          HIPCHK(hipEventRecord(G.fork_ev[p][slot], main));
          HIPCHK(hipStreamWaitEvent(side, G.fork_ev[p][slot], 0));
          rc = run_trace(s, pp.ts[blk], true, o.traversal, profile ? &timed : nullptr, counting, side, shadow_blocks);
          if (rc) return rc;
          HIPCHK(hipEventRecord(G.join_ev[p][slot], side));
        }
        TraceArgs tc = pp.tc[blk];
        tc.host_slot = G.d_counts + ((size_t) p * RING + slot) * HSLOT;   // this launch hands the queue counters to the host
        tc.stamp = pp.stamp[slot];
        rc = run_trace(s, tc, false, o.traversal, profile_closest ? &timed : nullptr, counting, main, pipe_blocks);
        if (rc) return rc;
        if (K && !fork) {
          rc = run_trace(s, pp.ts[blk], true, o.traversal, profile ? &timed : nullptr, counting, main, pipe_blocks);
          if (rc) return rc;
        }
        if (fork) HIPCHK(hipStreamWaitEvent(main, G.join_ev[p][slot], 0));
        if (p == 0) iters++;
        // The host runs two iterations ahead of the counters it has seen, so that both pipes always have work queued; in
        // the tail of a frame (few rays left: launches of a few microseconds, nothing to keep fed) it looks at the counters
        // of the iteration it has just issued - they arrive as soon as k_logic has run - and stops without issuing the
        // two or three empty iterations the lag would cost.
        const uint64_t upto = pp.tail ? it + 1 : (it + 1 >= RING - 1 ? it + 1 - (RING - 2) : 0);   // account iterations < upto
        while (pp.accounted < upto && !pp.done) {
          const uint64_t old = pp.accounted++;
          volatile uint32_t *hc = hcounts + (old % RING) * HSLOT;
          // the closest-hit launch wrote the counters, then the stamp (system-scope release): poll it
          const uint32_t want = pp.stamp[old % RING];
          const auto t_wait = std::chrono::steady_clock::now();
          for (uint32_t spin = 0; hc[7] != want; spin++) {
            if ((spin & 0xFFFFu) == 0xFFFFu) {
              if (hipStreamQuery(main) == hipSuccess && hc[7] != want)
                return fail(MIRT_E_DEVICE, "queue counters of an iteration never arrived");
              if (std::chrono::steady_clock::now() - t_wait > std::chrono::seconds(30))
                return fail(MIRT_E_DEVICE, "timed out waiting for the queue counters of an iteration");
            }
            __builtin_ia32_pause();
          }
          std::atomic_thread_fence(std::memory_order_acquire);
          const uint32_t c_front = hc[0], c_back = hc[1], c_shadow = hc[2], c_live = hc[3], c_culled = hc[4];   // QC_* / CTR_STRIDE
          // k_logic's work: iteration 0 runs every slot; iteration i + 1 runs the slots iteration i left rendering, and looks
          // at all of the pipe's slots unless the live-slot list is short enough (the rule of k_logic itself)
          if (old == 0) { n_live += pp.a.n_slots; n_scanned += pp.a.n_slots; }
          if (c_front + c_back + c_shadow != 0) {
            n_live += c_live;
            n_scanned += ((uint64_t) c_live * 16u <= (uint64_t) pp.a.n_slots * G.live_16th) ? c_live : pp.a.n_slots;
          }
          // a slot that is not finished leaves a ray in one of the queues: empty queues = this half is complete (and
          // every later iteration already launched found them empty too)
          n_camera += c_front;
          n_bounced += c_back;
          n_shadow_rays += c_shadow - c_culled;
          const uint32_t in_flight = c_front + c_back + c_shadow;
          if (in_flight == 0) pp.done = true;
          if (in_flight < pp.a.n_slots / 32u) pp.tail = true;
          if (g_debug_iters)
            fprintf(stderr, "[mirt] pipe %d iteration %llu: camera %u bounced %u shadow %u live %u%s\n", p, (unsigned long long) old, c_front,
                    c_back, c_shadow, c_live, pp.tail ? " (tail)" : "");
        }
      }
    }
    for (int p = 1; p < n_pipes; p++) {
      HIPCHK(hipEventRecord(G.pipe_ev[p], G.pstream[p]));
      HIPCHK(hipStreamWaitEvent(G.stream, G.pipe_ev[p], 0));
    }
  }
  if (via_planes && part.n_slots)
    HIPCHK(launch_sum_planes(s->planes.as<float4>(), shares, w * h, (float4 *) accum_dev, accumulate ? 1 : 0, G.stream));
  // error flags and ray statistics ride home behind the frame's last kernel, into pinned memory: no blocking copies after it
  constexpr size_t CTRL_ALL = (size_t) CTRL_BLOCKS * MAX_PIPES * CTRL_WORDS * 4;
  HIPCHK(hipEventRecord(evB, G.stream));
  HIPCHK(hipMemcpyAsync(G.h_tail, ctrl, CTRL_ALL, hipMemcpyDeviceToHost, G.stream));
  HIPCHK(hipMemcpyAsync((char *) G.h_tail + CTRL_ALL, s->ray_stats.p, MAX_PIPES * 16 * 8, hipMemcpyDeviceToHost, G.stream));
  HIPCHK(hipStreamSynchronize(G.stream));
  uint32_t err = 0;
  for (int b = 0; b < CTRL_BLOCKS * MAX_PIPES; b++) err |= G.h_tail[(size_t) b * CTRL_WORDS + CTRL_ERR];
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, evA, evB));
  drain.armed = false;                                                   // G.stream is idle and every other stream had joined it
  if (err) return fail(MIRT_E_LIMIT, "traversal stack overflow on the device (BVH too deep)");
  for (const Timed &t: timed) {
    float tms = 0;
    HIPCHK(hipEventElapsedTime(&tms, t.a, t.b));
    ms_kind[t.kind] += tms;
    n_kind[t.kind]++;
  }
  if (stats) {
    unsigned long long rs[16 * MAX_PIPES] = {0};
    memcpy(rs, (const char *) G.h_tail + CTRL_ALL, sizeof rs);
    for (int p = 1; p < MAX_PIPES; p++)
      for (int k = 0; k < 16; k++) rs[k] += rs[16 * p + k];              // the pipes' counters
    memset(stats, 0, sizeof *stats);
    stats->closest_rays = n_camera + n_bounced;
    stats->shadow_rays = n_shadow_rays;
    stats->paths = n_camera;
    stats->exact_fallback_rays = rs[3];
    stats->iterations = iters;
    stats->tonemap_table_exact = G.table_exact ? 1 : 0;
    stats->render_ms = ms;
    stats->trace_closest_ms = ms_kind[1];
    stats->trace_shadow_ms = ms_kind[2];
    stats->logic_ms = ms_kind[0];
    stats->closest_launches = n_kind[1];
    stats->shadow_launches = n_kind[2];
    stats->closest_node_visits = rs[4];
    stats->closest_tri_tests = rs[5];
    stats->shadow_node_visits = rs[6];
    stats->shadow_tri_tests = rs[7];
    stats->closest_node_steps = rs[8];
    stats->closest_tri_steps = rs[9];
    stats->shadow_node_steps = rs[10];
    stats->shadow_tri_steps = rs[11];
    stats->traversal_used = o.traversal;
    stats->logic_launches = n_kind[0];
    stats->logic_live_slots = n_live;
    stats->logic_scanned_slots = n_scanned;
    stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  }
  return MIRT_OK;
}

}  // namespace

extern "C" {

int mirt_render_accum(mirt_scene *scene, const mirt_camera *camera, const mirt_params *params,
                      const mirt_options *opts, uint32_t width, uint32_t height, void *accum_dev, mirt_stats *stats) {
  return render_impl(scene, camera, params, opts, width, height, accum_dev, stats);
}

int mirt_tonemap(const void *accum_dev, uint32_t width, uint32_t height, int32_t *rgb_out) {
  if (!G.ready) return fail(MIRT_E_STATE, "mirt_init() has not been called");
  if (!accum_dev || !rgb_out || !width || !height) return fail(MIRT_E_ARG, "null / empty argument");
  const size_t n = (size_t) width * height;
  HIPCHK(G.d_rgb.ensure(n * 4));                                         // kept between calls
  if (G.h_film_words < n) {                                              // pinned staging, kept between calls
    if (G.h_film) (void) hipHostFree(G.h_film);
    G.h_film = nullptr;
    G.h_film_words = 0;
    HIPCHK(hipHostMalloc((void **) &G.h_film, n * 4, hipHostMallocDefault));
    G.h_film_words = n;
  }
  // one packed word per pixel crosses PCIe, in a few pieces; host threads widen each piece to three ints as it lands
  constexpr int PIECES = 8;
  HIPCHK(hipEventRecord(G.ev_in, nullptr));                              // after the caller's NULL-stream work on accum_dev
  HIPCHK(hipStreamWaitEvent(G.stream, G.ev_in, 0));
  hipError_t e = launch_tonemap(accum_dev, (uint32_t) n, G.d_table.as<float>(), G.d_rgb.as<uint32_t>(), G.stream);
  size_t cut[PIECES + 1];
  for (int k = 0; k <= PIECES; k++) cut[k] = n * (size_t) k / PIECES;
  for (int k = 0; k < PIECES && e == hipSuccess; k++) {
    if (cut[k + 1] > cut[k])
      e = hipMemcpyAsync(G.h_film + cut[k], G.d_rgb.as<uint32_t>() + cut[k], (cut[k + 1] - cut[k]) * 4, hipMemcpyDeviceToHost,
                         G.stream);
    if (e == hipSuccess) e = hipEventRecord(G.film_ev[k], G.stream);
  }
  if (e != hipSuccess) {
    (void) hipStreamSynchronize(G.stream);
    return fail(MIRT_E_DEVICE, std::string("tonemap: ") + hipGetErrorString(e));
  }
  std::atomic<int> bad{0};
  const hipEvent_t *film_ev = G.film_ev;                                 // (the context is per thread: the workers get copies)
  const uint32_t *h_film = G.h_film;
  // (two workers per piece, each widening half of it: what is left after the last piece has landed is a sixteenth of the film)
  if (!G.film_pool) {
    const unsigned t = host_threads();
    G.film_pool = new WorkerPool(t > 1 ? (t > 2 * PIECES ? 2 * PIECES : t) - 1 : 0);       // + the calling thread
  }
  G.film_pool->run(2 * PIECES, [&](size_t j) {
    const size_t k = j / 2;
    if (hipEventSynchronize(film_ev[k]) != hipSuccess) { bad.store(1); return; }
    const size_t mid = (cut[k] + (cut[k + 1] - cut[k]) / 2) & ~(size_t) 3;
    const size_t lo = (j & 1) ? (mid > cut[k] ? mid : cut[k]) : cut[k], hi = (j & 1) ? cut[k + 1] : (mid > cut[k] ? mid : cut[k]);
    if (hi > lo) widen_film(h_film + lo, hi - lo, rgb_out + 3 * lo);
  });
  e = hipStreamSynchronize(G.stream);
  if (e != hipSuccess || bad.load()) return fail(MIRT_E_DEVICE, std::string("tonemap: ") + hipGetErrorString(e));
  return MIRT_OK;
}

int mirt_render(mirt_scene *scene, const mirt_camera *camera, const mirt_params *params, const mirt_options *opts,
                uint32_t width, uint32_t height, int32_t *rgb_out, mirt_stats *stats) {
  if (!rgb_out) return fail(MIRT_E_ARG, "rgb_out is null");
  if (!scene) return fail(MIRT_E_STATE, "SCENE NOT SET!");
  if (width == 0 || height == 0) return fail(MIRT_E_STATE, "SAMPLE BUFFER EMPTY");
  auto wall0 = std::chrono::steady_clock::now();
  HIPCHK(scene->out_accum.ensure((size_t) width * height * 16));
  int rc = render_impl(scene, camera, params, opts, width, height, scene->out_accum.p, stats);
  if (rc) return rc;
  rc = mirt_tonemap(scene->out_accum.p, width, height, rgb_out);
  if (rc) return rc;
  if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  return MIRT_OK;
}

int mirt_primary_hits(mirt_scene *s, const mirt_camera *camera, const mirt_options *opts_in, uint32_t w, uint32_t h,
                      int32_t *idx_out, uint32_t *tbits_out) {
  if (!G.ready) return fail(MIRT_E_STATE, "mirt_init() has not been called");
  if (!s || !camera || !idx_out || !tbits_out || !w || !h) return fail(MIRT_E_ARG, "null / empty argument");
  mirt_options o;
  memset(&o, 0, sizeof o);
  if (opts_in) o = *opts_in;
  if (o.rng_mode > 1 || o.traversal > 2) return fail(MIRT_E_ARG, "bad rng_mode / traversal");
  if (o.traversal == 0 && !s->fast_ok) o.traversal = 1;
  o.tile_rank = 0;
  o.tile_world = 1;
  const Partition part = make_partition(w, h, o);
  int rc = ensure_work(s, part.n_slots, 0);
  if (rc) return rc;
  const size_t n = (size_t) w * h;
  HIPCHK(s->probe_idx.ensure(n * 4));
  HIPCHK(s->probe_t.ensure(n * 4));
  RenderArgs a;
  memset(&a, 0, sizeof a);
  a.n_slots = part.n_slots; a.width = w; a.height = h;
  a.tiles_x = part.tiles_x; a.tiles_y = part.tiles_y; a.tile_rank = 0; a.tile_world = 1;
  a.fd_pixel_slots = make_fastdiv(part.n_slots); a.fd_tiles_x = make_fastdiv(part.tiles_x);
  a.rng_mode = o.rng_mode; a.seed = o.seed; a.spp = 1; a.batch = 2;
  a.cam = to_device_camera(*camera);
  a.st = s->st.as<uint4>(); a.mask = s->mask.as<rec3>(); a.total = s->total.as<float4>();
  a.accum = s->accum.as<float4>(); a.ray_o = s->ray_o.as<float4>(); a.ray_d = s->ray_d.as<float4>();
  a.hit = s->hit.as<float4>(); a.ext_queue = s->ext_queue.as<uint32_t>();
  uint32_t *ctrl = s->ctrl.as<uint32_t>();
  a.qcount = ctrl + CTRL_QCOUNT;
  HIPCHK(hipMemsetAsync(ctrl, 0, CTRL_WORDS * 4, G.stream));
  HIPCHK(launch_init(a, G.stream));
  HIPCHK(launch_probe_primary(a, G.stream));
  rc = run_trace(s, closest_args(s, part.n_slots, o.traversal), false, o.traversal, nullptr, false, G.stream);
  if (rc) return rc;
  HIPCHK(launch_gather_primary(a, s->probe_idx.as<int32_t>(), s->probe_t.as<uint32_t>(), G.stream));
  HIPCHK(hipMemcpyAsync(idx_out, s->probe_idx.p, n * 4, hipMemcpyDeviceToHost, G.stream));
  HIPCHK(hipMemcpyAsync(tbits_out, s->probe_t.p, n * 4, hipMemcpyDeviceToHost, G.stream));
  HIPCHK(hipStreamSynchronize(G.stream));
  uint32_t err = 0;
  HIPCHK(hipMemcpy(&err, ctrl + CTRL_ERR, 4, hipMemcpyDeviceToHost));
  if (err) return fail(MIRT_E_LIMIT, "traversal stack overflow on the device (BVH too deep)");
  return MIRT_OK;
}

int mirt_trace(mirt_scene *s, const float *rays, uint32_t n, int any_hit, uint32_t traversal, int32_t *idx_out,
               uint32_t *tbits_out, float *bary_out) {
  if (!G.ready) return fail(MIRT_E_STATE, "mirt_init() has not been called");
  if (!s || !idx_out || !tbits_out) return fail(MIRT_E_ARG, "null argument");
  if (traversal > 2) return fail(MIRT_E_ARG, "bad traversal");
  if (traversal == 0 && !s->fast_ok) traversal = 1;
  if (n == 0) return MIRT_OK;
  if (!rays) return fail(MIRT_E_ARG, "rays is null");
  int rc = ensure_work(s, n, 1);
  if (rc) return rc;
  std::vector<float> ro((size_t) n * 4), rd((size_t) n * 4);
  for (uint32_t i = 0; i < n; i++) {
    const float *r = rays + 8 * (size_t) i;
    ro[4 * (size_t) i + 0] = r[0]; ro[4 * (size_t) i + 1] = r[1]; ro[4 * (size_t) i + 2] = r[2]; ro[4 * (size_t) i + 3] = r[6];
    rd[4 * (size_t) i + 0] = r[3]; rd[4 * (size_t) i + 1] = r[4]; rd[4 * (size_t) i + 2] = r[5]; rd[4 * (size_t) i + 3] = r[7];
  }
  uint32_t *ctrl = s->ctrl.as<uint32_t>();
  HIPCHK(hipMemsetAsync(ctrl, 0, CTRL_WORDS * 4, G.stream));
  TraceArgs t;
  if (any_hit) {
    HIPCHK(s->any_t.ensure((size_t) n * 4));
    HIPCHK(hipMemcpyAsync(s->sh_o.p, ro.data(), (size_t) n * 16, hipMemcpyHostToDevice, G.stream));
    HIPCHK(hipMemcpyAsync(s->sh_d.p, rd.data(), (size_t) n * 16, hipMemcpyHostToDevice, G.stream));
    s->fill_sig.buf = nullptr;                               // (the pre-filled direction records of a render are gone)
    HIPCHK(launch_iota(s->sh_queue.as<uint32_t>(), n, G.stream));
    HIPCHK(hipMemcpyAsync(ctrl + CTRL_QCOUNT + QC_SHADOW, &n, 4, hipMemcpyHostToDevice, G.stream));
    t = shadow_args(s, traversal);
    t.any_t = s->any_t.as<float>();
  } else {
    HIPCHK(hipMemcpyAsync(s->ray_o.p, ro.data(), (size_t) n * 16, hipMemcpyHostToDevice, G.stream));
    HIPCHK(hipMemcpyAsync(s->ray_d.p, rd.data(), (size_t) n * 16, hipMemcpyHostToDevice, G.stream));
    HIPCHK(launch_iota(s->ext_queue.as<uint32_t>(), n, G.stream));
    HIPCHK(hipMemcpyAsync(ctrl + CTRL_QCOUNT + QC_EXT_FRONT, &n, 4, hipMemcpyHostToDevice, G.stream));
    t = closest_args(s, n, traversal);
  }
  HIPCHK(hipStreamSynchronize(G.stream));     // the host staging vectors and &n must outlive the copies
  rc = run_trace(s, t, any_hit != 0, traversal, nullptr, false, G.stream);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(G.stream));
  if (any_hit) {
    std::vector<uint32_t> occ(n);
    std::vector<float> tt(n);
    HIPCHK(hipMemcpy(occ.data(), s->sh_occ.p, (size_t) n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tt.data(), s->any_t.p, (size_t) n * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) {
      idx_out[i] = occ[i] ? 1 : -1;
      memcpy(&tbits_out[i], &tt[i], 4);
    }
  } else {
    std::vector<float> hit((size_t) n * 4);
    HIPCHK(hipMemcpy(hit.data(), s->hit.p, (size_t) n * 16, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) {
      memcpy(&tbits_out[i], &hit[4 * (size_t) i], 4);
      memcpy(&idx_out[i], &hit[4 * (size_t) i + 3], 4);
      if (bary_out) {
        bary_out[2 * (size_t) i] = hit[4 * (size_t) i + 1];
        bary_out[2 * (size_t) i + 1] = hit[4 * (size_t) i + 2];
      }
    }
  }
  uint32_t err = 0;
  HIPCHK(hipMemcpy(&err, ctrl + CTRL_ERR, 4, hipMemcpyDeviceToHost));
  if (err) return fail(MIRT_E_LIMIT, "traversal stack overflow on the device (BVH too deep)");
  return MIRT_OK;
}

// mirt_trace for callers that cannot allocate: device in, device out, every intermediate in `scratch`.  No hipMalloc /
// hipFree / hipHostMalloc, no event or stream creation, no host containers: launches, one memset, one 4-byte copy into the
// pinned block mirt_init made, one stream synchronisation.
int mirt_trace_static(mirt_scene *s, const float *rays_dev, uint32_t n, int any_hit, uint32_t traversal, int32_t *idx_dev,
                      uint32_t *tbits_dev) {
  if (!G.ready) return fail(MIRT_E_STATE, "mirt_init() has not been called");
  if (!s || !idx_dev || !tbits_dev) return fail(MIRT_E_ARG, "null argument");
  if (traversal > 2) return fail(MIRT_E_ARG, "bad traversal");
  if (n > STATIC_MAX_RAYS)
    return fail(MIRT_E_LIMIT, "mirt_trace_static: more than MIRT_TRACE_STATIC_MAX_RAYS (65536) rays in one call");
  if (traversal == 0 && !s->fast_ok) traversal = 1;
  unsigned char *base = (G.device >= 0 && G.device < MAX_DEVICES) ? g_scratch_ptr[G.device] : nullptr;
  if (!base) return fail(MIRT_E_STATE, "mirt_trace_static: no scratch address for this device");
  if (s->device != G.device) return fail(MIRT_E_STATE, "mirt_trace_static: the scene lives on another device than this thread's context");
  if (n == 0) return MIRT_OK;
  if (!rays_dev) return fail(MIRT_E_ARG, "rays_dev is null");
  if (((uintptr_t) rays_dev & 15u) || ((uintptr_t) idx_dev & 3u) || ((uintptr_t) tbits_dev & 3u))
    return fail(MIRT_E_ARG, "mirt_trace_static: rays_dev must be 16-byte aligned (it is read as float4), idx_dev / tbits_dev 4-byte aligned");
  uint32_t *ctrl = (uint32_t *) (base + SO_CTRL);
  TraceArgs t = common_args(s, traversal);
  t.ray_o = (const float4 *) (base + SO_RAY_O);
  t.ray_d = (const float4 *) (base + SO_RAY_D);
  t.queue = (const uint32_t *) (base + SO_QUEUE);
  t.qcount = ctrl + CTRL_QCOUNT;
  t.qc_front = any_hit ? QC_SHADOW : QC_EXT_FRONT;
  t.qc_back = -1;
  t.qcap = n;
  t.hit = (float4 *) (base + SO_HIT);
  t.occ = (uint32_t *) (base + SO_OCC);
  t.any_t = any_hit ? (float *) (base + SO_ANY_T) : nullptr;
  t.work_head = ctrl + CTRL_WORK;
  t.ovf = (uint2 *) (base + SO_OVF);
  t.ovf_depth = std::min(s->ovf_depth, STATIC_OVF_DEPTH);               // (MIRT_OVF_CAP may exceed the static region's depth)
  t.redo_queue = (uint32_t *) (base + SO_REDO);
  t.redo_count = ctrl + CTRL_REDO;
  t.error_flag = ctrl + CTRL_ERR;
  t.redo_stat = nullptr;
  t.count_stat = nullptr;
  // the array keeps the last call's contents: the whole control block (counters, work heads, redo count, error flag)
  // starts at zero; rays, queue and results are rewritten for [0, n) before they are read
  HIPCHK(hipEventRecord(G.ev_in, nullptr));                              // after the caller's NULL-stream work on its buffers
  HIPCHK(hipStreamWaitEvent(G.stream, G.ev_in, 0));
  HIPCHK(hipMemsetAsync(ctrl, 0, CTRL_WORDS * 4, G.stream));
  HIPCHK(launch_stage_rays(rays_dev, n, (float4 *) (base + SO_RAY_O), (float4 *) (base + SO_RAY_D), (uint32_t *) (base + SO_QUEUE),
                           ctrl + CTRL_QCOUNT + t.qc_front, G.stream));
  int rc = run_trace(s, t, any_hit != 0, traversal, nullptr, false, G.stream, 0, STATIC_GRID);
  if (rc) { (void) hipStreamSynchronize(G.stream); return rc; }
  HIPCHK(launch_unstage_hits(n, any_hit ? 1 : 0, t.hit, t.occ, t.any_t, idx_dev, tbits_dev, G.stream));
  HIPCHK(hipMemcpyAsync(G.h_counts, ctrl + CTRL_ERR, 4, hipMemcpyDeviceToHost, G.stream));   // pinned: no staging
  HIPCHK(hipStreamSynchronize(G.stream));
  if (G.h_counts[0]) return fail(MIRT_E_LIMIT, "traversal stack overflow on the device (BVH too deep)");
  return MIRT_OK;
}

int mirt_save_bmp(const char *path, const int32_t *rgb, uint32_t width, uint32_t height) {
  if (!path || !rgb || !width || !height) return fail(MIRT_E_ARG, "null / empty argument");
  if (write_bmp(path, rgb, width, height) != 0) return fail(MIRT_E_IO, std::string("could not write ") + path);
  return MIRT_OK;
}

int mirt_bvh_build_host(const float *tri_pos, uint32_t n_tris, uint32_t leaf_size, uint32_t counts[4],
                        uint32_t *nodes2_out, uint32_t *flat_to_orig, uint32_t *nodes4_out) {
  uint32_t c8[8];
  int rc = mirt_bvh_build_host_prims(tri_pos, nullptr, n_tris, leaf_size, c8, nodes2_out, flat_to_orig, nodes4_out, nullptr,
                                     nullptr);
  if (rc == MIRT_OK && counts) memcpy(counts, c8, 4 * sizeof(uint32_t));
  return rc;
}

int mirt_bvh_build_host_prims(const float *prim_pos, const uint8_t *prim_type, uint32_t n_prims, uint32_t leaf_size,
                              uint32_t counts[8], uint32_t *nodes2_out, uint32_t *flat_to_orig, uint32_t *nodes4_out,
                              uint32_t *nodes8q_out, uint32_t *fast_to_flat_out) {
  if (!prim_pos || !counts || n_prims == 0) return fail(MIRT_E_ARG, "null / empty argument");
  HostBVH b;
  try {
    build_reference_bvh2(prim_pos, prim_type, n_prims, leaf_size, b);
    collapse_to_wide(b);
    std::vector<float> flat_pos((size_t) n_prims * 9);
    std::vector<uint8_t> flat_type;
    if (prim_type) flat_type.resize(n_prims);
    for (uint32_t f = 0; f < n_prims; f++) {
      memcpy(&flat_pos[(size_t) f * 9], prim_pos + 9 * (size_t) b.flat_to_orig[f], 9 * sizeof(float));
      if (prim_type) flat_type[f] = prim_type[b.flat_to_orig[f]];
    }
    build_fast_tree(flat_pos.data(), prim_type ? flat_type.data() : nullptr, n_prims, b);
  } catch (const std::exception &ex) {
    return fail(MIRT_E_LIMIT, ex.what());
  }
  counts[0] = (uint32_t) b.nodes2.size();
  counts[1] = b.depth2;
  counts[2] = (uint32_t) b.nodes4.size();
  counts[3] = b.depth4;
  counts[4] = (uint32_t) (b.nodes8q.size() / (sizeof(Node8Q) / 4));
  counts[5] = b.depth8;
  counts[6] = 0;
  counts[7] = 0;
  if (nodes2_out) memcpy(nodes2_out, b.nodes2.data(), b.nodes2.size() * sizeof(Node2));
  if (flat_to_orig) memcpy(flat_to_orig, b.flat_to_orig.data(), b.flat_to_orig.size() * 4);
  if (nodes4_out) memcpy(nodes4_out, b.nodes4.data(), b.nodes4.size() * sizeof(Node4));
  if (nodes8q_out) memcpy(nodes8q_out, b.nodes8q.data(), b.nodes8q.size() * 4);
  if (fast_to_flat_out) memcpy(fast_to_flat_out, b.fast_to_flat.data(), b.fast_to_flat.size() * 4);
  return MIRT_OK;
}

int mirt_tonemap_host(const float *rgb_in, uint32_t n_pixels, int32_t *rgb_out) {
  if (!rgb_in || !rgb_out) return fail(MIRT_E_ARG, "null argument");
  static float table[256];
  static bool have = false;
  if (!have) {
    build_tonemap_table(table);
    have = true;
  }
  for (size_t i = 0; i < (size_t) n_pixels * 3; i++) rgb_out[i] = tonemap_channel_table(table, rgb_in[i]);
  return MIRT_OK;
}

}  // extern "C"
