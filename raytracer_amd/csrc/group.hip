// mirt_group_*: one frame over several GPUs from ONE host process (include/mirt.h).  The reference has no multi-GPU code
// (SURVEY 8e); this is the route by which a C++ caller of the kept host API (PathTracer) reaches N devices without Python:
// one persistent host thread per member, each with its own libmirt context (streams, events, pinned memory are
// thread-local) and its own replica of the scene; a frame = every member renders its partition (spp slices or interleaved
// tiles, the same mirt_options the one-process-per-GPU path of bench.py uses) into its accumulation buffer, then ONE
// sum-reduce onto member 0 and the film.  The reduce is RCCL's ncclReduce over xGMI when every member sits on a GPU of its
// own (librccl is loaded at run time, only then); members that share a GPU (logical partitions: tests on a one-GPU box,
// oversubscription) are added by a kernel in member order instead, so the result does not depend on scheduling.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mirt.h"
#include "kernels.hpp"

namespace mirt {
hipStream_t current_stream();                     // capi.hip: the calling thread's main stream
int set_error(int code, const std::string &msg);  // capi.hip: sets mirt_last_error() of the calling thread
int scene_clone(const mirt_scene *src, mirt_scene **out);   // capi.hip: a replica of a scene on the calling thread's device
}  // namespace mirt

namespace {

// the five RCCL entry points used, bound with dlsym (no link-time dependency: single-GPU users never load librccl)
struct Rccl {
  void *lib = nullptr;
  int (*CommInitAll)(void **, int, const int *) = nullptr;
  int (*Reduce)(const void *, void *, size_t, int, int, int, void *, hipStream_t) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool load(std::string &err) {
    if (lib) return true;
    for (const char *name: {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) { err = std::string("librccl not found: ") + dlerror(); return false; }
    CommInitAll = (decltype(CommInitAll)) dlsym(lib, "ncclCommInitAll");
    Reduce = (decltype(Reduce)) dlsym(lib, "ncclReduce");
    CommDestroy = (decltype(CommDestroy)) dlsym(lib, "ncclCommDestroy");
    GetErrorString = (decltype(GetErrorString)) dlsym(lib, "ncclGetErrorString");
    if (!CommInitAll || !Reduce || !CommDestroy || !GetErrorString) { err = "librccl lacks an expected symbol"; return false; }
    return true;
  }
};
constexpr int NCCL_FLOAT = 7, NCCL_SUM = 0;       // rccl.h: ncclFloat32, ncclSum

struct Member {
  int device = 0, index = 0;
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, quit = false, done = true;
  int rc = 0;
  std::string err;
  mirt_scene *scene = nullptr;
  void *accum = nullptr;                          // w * h float4 on this member's device
  size_t accum_bytes = 0;
  void *comm = nullptr;                           // ncclComm_t
  mirt_stats stats;
};

}  // namespace

struct mirt_group {
  std::vector<Member *> members;
  bool use_rccl = false;
  Rccl rccl;
  void *stage = nullptr;                          // member 0's device: a peer's buffer on its way into the sum
  size_t stage_bytes = 0;

  void worker(Member *mb) {
    int rc = mirt_init(mb->device);               // this thread's context
    {
      std::lock_guard<std::mutex> lk(mb->m);
      mb->rc = rc;
      if (rc) mb->err = mirt_last_error();
      mb->done = true;
    }
    mb->cv.notify_all();
    for (;;) {
      std::function<int()> job;
      {
        std::unique_lock<std::mutex> lk(mb->m);
        mb->cv.wait(lk, [&] { return mb->has_job || mb->quit; });
        if (mb->quit) break;
        job = mb->job;
        mb->has_job = false;
      }
      rc = job();
      {
        std::lock_guard<std::mutex> lk(mb->m);
        mb->rc = rc;
        if (rc) mb->err = mirt_last_error();
        mb->done = true;
      }
      mb->cv.notify_all();
    }
    if (mb->scene) mirt_scene_destroy(mb->scene);
    if (mb->accum) (void) hipFree(mb->accum);
    mirt_shutdown();
  }
  // run job(member) on every member's thread, wait for all; first failure wins
  int run(const std::function<int(Member &)> &job) {
    for (Member *mb: members) {
      std::lock_guard<std::mutex> lk(mb->m);
      mb->job = [mb, &job] { return job(*mb); };
      mb->has_job = true;
      mb->done = false;
      mb->cv.notify_all();
    }
    int rc = 0;
    std::string err;
    for (Member *mb: members) {
      std::unique_lock<std::mutex> lk(mb->m);
      mb->cv.wait(lk, [&] { return mb->done; });
      if (mb->rc && !rc) { rc = mb->rc; err = "member " + std::to_string(mb->index) + " (device " + std::to_string(mb->device) + "): " + mb->err; }
    }
    return rc ? mirt::set_error(rc, err) : MIRT_OK;
  }
};

extern "C" {

int mirt_group_create(const int *devices, int n, mirt_group **out) {
  if (!devices || !out || n < 1 || n > 64) return mirt::set_error(MIRT_E_ARG, "mirt_group_create: 1..64 members");
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return mirt::set_error(MIRT_E_DEVICE, "no HIP device available (libmirt has no CPU fallback)");
  bool distinct = true;
  for (int i = 0; i < n; i++) {
    if (devices[i] < 0 || devices[i] >= n_dev) return mirt::set_error(MIRT_E_ARG, "mirt_group_create: device index out of range");
    for (int j = 0; j < i; j++) distinct = distinct && devices[i] != devices[j];
  }
  auto *g = new mirt_group();
  // (MIRT_GROUP_FORCE_RCCL: also for a single member, where the reduce is a no-op - exercises the RCCL binding on a one-GPU box)
  g->use_rccl = distinct && (n > 1 || getenv("MIRT_GROUP_FORCE_RCCL") != nullptr) && getenv("MIRT_GROUP_NO_RCCL") == nullptr;
  std::vector<void *> comms((size_t) n, nullptr);
  if (g->use_rccl) {
    std::string err;
    if (!g->rccl.load(err)) { delete g; return mirt::set_error(MIRT_E_DEVICE, err); }
    const int rc = g->rccl.CommInitAll(comms.data(), n, devices);
    if (rc != 0) {
      const std::string msg = std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(rc);
      delete g;
      return mirt::set_error(MIRT_E_DEVICE, msg);
    }
  }
  for (int i = 0; i < n; i++) {
    auto *mb = new Member();
    mb->device = devices[i];
    mb->index = i;
    mb->comm = comms[(size_t) i];
    mb->done = false;
    memset(&mb->stats, 0, sizeof mb->stats);
    g->members.push_back(mb);
    mb->th = std::thread([g, mb] { g->worker(mb); });
  }
  int rc = 0;
  std::string err;
  for (Member *mb: g->members) {
    std::unique_lock<std::mutex> lk(mb->m);
    mb->cv.wait(lk, [&] { return mb->done; });
    if (mb->rc && !rc) { rc = mb->rc; err = mb->err; }
  }
  if (rc) {
    mirt_group_destroy(g);
    return mirt::set_error(rc, "mirt_group_create: " + err);
  }
  *out = g;
  return MIRT_OK;
}

void mirt_group_destroy(mirt_group *g) {
  if (!g) return;
  const int root_device = g->members.empty() ? 0 : g->members[0]->device;
  for (Member *mb: g->members) {
    {
      std::lock_guard<std::mutex> lk(mb->m);
      mb->quit = true;
    }
    mb->cv.notify_all();
    if (mb->th.joinable()) mb->th.join();
    if (mb->comm && g->rccl.CommDestroy) (void) g->rccl.CommDestroy(mb->comm);
    delete mb;
  }
  if (g->stage) {
    // (allocated on member 0's device by member 0's thread, which is gone: free from here with that device current)
    int cur = 0;
    (void) hipGetDevice(&cur);
    (void) hipSetDevice(root_device);
    (void) hipFree(g->stage);
    (void) hipSetDevice(cur);
  }
  delete g;
}

int mirt_group_size(const mirt_group *g) { return g ? (int) g->members.size() : 0; }

int mirt_group_set_scene(mirt_group *g, const float *tri_pos, const float *tri_nrm, const uint32_t *tri_mat, const uint8_t *prim_type,
                         uint32_t n_prims, const mirt_bsdf *mats, uint32_t n_mats, const mirt_light *lights, uint32_t n_lights,
                         uint32_t leaf_size, const mirt_build_options *build) {
  if (!g) return mirt::set_error(MIRT_E_ARG, "null group");
  // The first member builds the scene (the reference's BVH2, its fold, the SAH tree: host threads, then the uploads); the others
  // then replicate its device buffers.  (Every member used to run the whole host build at the same time: N times the work on N
  // times oversubscribed cores.)
  int rc = g->run([&](Member &mb) {
    if (mb.scene) mirt_scene_destroy(mb.scene);
    mb.scene = nullptr;
    if (mb.index != 0) return (int) MIRT_OK;
    const int r = mirt_scene_create_opts(tri_pos, tri_nrm, tri_mat, prim_type, n_prims, mats, n_mats, lights, n_lights, leaf_size, build,
                                         &mb.scene);
    if (r == MIRT_OK && hipDeviceSynchronize() != hipSuccess) return mirt::set_error(MIRT_E_DEVICE, "device error after the scene build");
    return r;
  });
  if (rc != MIRT_OK || g->members.size() < 2) return rc;
  const mirt_scene *first = g->members[0]->scene;
  return g->run([&](Member &mb) {
    if (mb.index == 0) return (int) MIRT_OK;
    return mirt::scene_clone(first, &mb.scene);
  });
}

int mirt_group_render(mirt_group *g, const mirt_camera *camera, const mirt_params *params, const mirt_options *opts_in,
                      int partition, uint32_t width, uint32_t height, int32_t *rgb_out, mirt_stats *stats) {
  if (!g) return mirt::set_error(MIRT_E_ARG, "null group");
  if (!camera || !params || !rgb_out) return mirt::set_error(MIRT_E_ARG, "null argument");
  if (partition != MIRT_PARTITION_SPP && partition != MIRT_PARTITION_TILE) return mirt::set_error(MIRT_E_ARG, "partition must be MIRT_PARTITION_SPP or MIRT_PARTITION_TILE");
  if (!width || !height) return mirt::set_error(MIRT_E_STATE, "SAMPLE BUFFER EMPTY");
  const uint32_t n = (uint32_t) g->members.size();
  mirt_options base;
  memset(&base, 0, sizeof base);
  if (opts_in) base = *opts_in;
  if (base.tile_world > 1 || base.sample_count || base.sample_begin)
    return mirt::set_error(MIRT_E_ARG, "mirt_group_render partitions the frame itself: leave tile_* / sample_* zero");
  if (n > 1 && partition == MIRT_PARTITION_SPP) {
    if (params->samples_per_pixel < n) return mirt::set_error(MIRT_E_ARG, "spp partition needs at least one sample per member: use MIRT_PARTITION_TILE");
    base.rng_mode = 1;                            // the xorshift stream of a pixel is sequential: slices need the counter RNG
  }
  const size_t bytes = (size_t) width * height * 16;
  // 1. every member renders its share into its own accumulation buffer (pixels / samples it does not own: zeros)
  int rc = g->run([&](Member &mb) {
    if (!mb.scene) return mirt::set_error(MIRT_E_STATE, "SCENE NOT SET!");
    if (mb.accum_bytes < bytes) {
      if (mb.accum) (void) hipFree(mb.accum);
      mb.accum = nullptr;
      mb.accum_bytes = 0;
      if (hipMalloc(&mb.accum, bytes) != hipSuccess) return mirt::set_error(MIRT_E_DEVICE, "out of device memory (accumulation buffer)");
      mb.accum_bytes = bytes;
    }
    mirt_options o = base;
    if (n > 1 && partition == MIRT_PARTITION_TILE) {
      o.tile_rank = (uint32_t) mb.index;
      o.tile_world = n;
    } else if (n > 1) {
      const uint32_t per = params->samples_per_pixel / n;          // raytracer_amd/dist.py spp_slices: the last takes the rest
      o.sample_begin = (uint32_t) mb.index * per;
      o.sample_count = (uint32_t) mb.index + 1 < n ? per : params->samples_per_pixel - per * (n - 1);
    }
    if (o.shares > 1) {
      // share pool: the shares must divide the samples THIS member renders - take the largest common divisor of what the
      // caller asked for and the member's sample count (a result not above `streams` switches the pool off, mirt.h)
      uint32_t a = o.shares, b = o.sample_count ? o.sample_count : params->samples_per_pixel;
      while (b) { const uint32_t t = a % b; a = b; b = t; }
      o.shares = a;
    }
    return mirt_render_accum(mb.scene, camera, params, &o, width, height, mb.accum, &mb.stats);
  });
  if (rc) return rc;
  // 2. ONE sum onto member 0
  if (g->use_rccl) {
    rc = g->run([&](Member &mb) {
      const int e = g->rccl.Reduce(mb.accum, mb.accum, (size_t) width * height * 4, NCCL_FLOAT, NCCL_SUM, 0, mb.comm, mirt::current_stream());
      if (e != 0) return mirt::set_error(MIRT_E_DEVICE, std::string("ncclReduce: ") + g->rccl.GetErrorString(e));
      if (hipStreamSynchronize(mirt::current_stream()) != hipSuccess) return mirt::set_error(MIRT_E_DEVICE, "ncclReduce: stream failed");
      return MIRT_OK;
    });
    if (rc) return rc;
  } else if (n > 1) {
    Member &root = *g->members[0];
    rc = g->run([&](Member &mb) {
      if (mb.index != 0) return MIRT_OK;
      hipStream_t st = mirt::current_stream();
      for (uint32_t r = 1; r < n; r++) {                           // member order: the sum does not depend on scheduling
        Member &peer = *g->members[r];
        const void *src = peer.accum;
        if (peer.device != root.device) {
          if (g->stage_bytes < bytes) {
            if (g->stage) (void) hipFree(g->stage);
            g->stage = nullptr;
            g->stage_bytes = 0;
            if (hipMalloc(&g->stage, bytes) != hipSuccess) return mirt::set_error(MIRT_E_DEVICE, "out of device memory (staging buffer)");
            g->stage_bytes = bytes;
          }
          if (hipMemcpyPeerAsync(g->stage, root.device, peer.accum, peer.device, bytes, st) != hipSuccess)
            return mirt::set_error(MIRT_E_DEVICE, "peer copy of an accumulation buffer failed");
          src = g->stage;
        }
        if (mirt::launch_sum_planes((const float4 *) src, 1, width * height, (float4 *) root.accum, 1, st) != hipSuccess)
          return mirt::set_error(MIRT_E_DEVICE, "sum of the accumulation buffers failed");
      }
      if (hipStreamSynchronize(st) != hipSuccess) return mirt::set_error(MIRT_E_DEVICE, "sum of the accumulation buffers failed");
      return MIRT_OK;
    });
    if (rc) return rc;
  }
  // 3. film on member 0
  rc = g->run([&](Member &mb) { return mb.index == 0 ? mirt_tonemap(mb.accum, width, height, rgb_out) : MIRT_OK; });
  if (rc) return rc;
  if (stats) {
    memset(stats, 0, sizeof *stats);
    for (Member *mb: g->members) {
      stats->closest_rays += mb->stats.closest_rays;
      stats->shadow_rays += mb->stats.shadow_rays;
      stats->paths += mb->stats.paths;
      stats->exact_fallback_rays += mb->stats.exact_fallback_rays;
      if (mb->stats.iterations > stats->iterations) stats->iterations = mb->stats.iterations;
      if (mb->stats.render_ms > stats->render_ms) stats->render_ms = mb->stats.render_ms;
      if (mb->stats.total_ms > stats->total_ms) stats->total_ms = mb->stats.total_ms;
      // work counters add up over the members, per-kernel times are the slowest member's
      stats->logic_live_slots += mb->stats.logic_live_slots;
      stats->logic_scanned_slots += mb->stats.logic_scanned_slots;
      stats->closest_node_visits += mb->stats.closest_node_visits;
      stats->closest_tri_tests += mb->stats.closest_tri_tests;
      stats->shadow_node_visits += mb->stats.shadow_node_visits;
      stats->shadow_tri_tests += mb->stats.shadow_tri_tests;
      stats->closest_launches += mb->stats.closest_launches;
      stats->shadow_launches += mb->stats.shadow_launches;
      stats->logic_launches += mb->stats.logic_launches;
      if (mb->stats.trace_closest_ms > stats->trace_closest_ms) stats->trace_closest_ms = mb->stats.trace_closest_ms;
      if (mb->stats.trace_shadow_ms > stats->trace_shadow_ms) stats->trace_shadow_ms = mb->stats.trace_shadow_ms;
      if (mb->stats.logic_ms > stats->logic_ms) stats->logic_ms = mb->stats.logic_ms;
    }
    stats->tonemap_table_exact = g->members[0]->stats.tonemap_table_exact;
    stats->traversal_used = g->members[0]->stats.traversal_used;
  }
  return MIRT_OK;
}

}  // extern "C"
