// Device build of the ordered traversal's own tree (build.hip).  Product code; nothing here touches oracle/.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace mirt {

struct DeviceTree {
  void *nodes8q = nullptr;       // n_nodes x 96 bytes (Node8Q, host_bvh.hpp); hipMalloc'ed, the caller owns it
  void *trisf = nullptr;         // n x 48 bytes: the caller's flat-order primitive records in this tree's order
  size_t nodes_bytes = 0, tris_bytes = 0;
  uint32_t n_nodes = 0;
  uint32_t depth8 = 0;           // levels below the root
  uint32_t passes = 0;           // PLOC merge passes
  float ms[4] = {0, 0, 0, 0};    // upload + keys + sort | binary tree + boxes | fold | gather
  std::string error;
};

// pos_flat_host / ptype_flat_host: the primitives in the reference's flat order (host memory; ptype NULL = all triangles);
// tris_dev: their 48-byte intersector records on the device, same order.  Returns 0 on success, 1 when this scene cannot
// have a quantised tree (non-finite vertices, extent beyond float range, fewer than two primitives: the caller falls back
// exactly as for the host builder), -1 on a HIP error.  Synchronises `stream`.
int build_fast_tree_device(const float *pos_flat_host, const uint8_t *ptype_flat_host, const void *tris_dev, uint32_t n,
                           uint32_t max_leaf, int method /* 1 Morton radix tree, 2 PLOC */, hipStream_t stream,
                           DeviceTree &out);

}  // namespace mirt
