#!/usr/bin/env python3
"""Surface-area cost of the host-built 8-wide tree (no GPU needed): expected node visits and primitive tests of a random
long ray = sum over children of area(child box) / area(root box).   python3 tools/sah_cost.py [workload ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from raytracer_amd import capi, scenes  # noqa: E402


def cost(nodes):
    w = nodes
    p = w[:, :3].view(np.float32).astype(np.float64)                       # (N, 3)
    ex = np.stack([(w[:, 3] >> (8 * a)) & 0xFF for a in range(3)], 1).astype(np.float64)
    step = 2.0 ** (ex - 127)
    q = w[:, 12:24].copy().view(np.uint8).reshape(-1, 6, 8).astype(np.float64)   # qlo xyz, qhi xyz
    ext = (q[:, 3:6, :] - q[:, 0:3, :]) * step[:, :, None]                 # (N, 3, 8)
    area = ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]   # (N, 8)
    ref = w[:, 4:12]
    valid = ref != 0xFFFFFFFF
    leaf = valid & ((ref & 0x80000000) != 0)
    inner = valid & ~leaf
    cnt = (ref >> 27) & 15
    root_ext = (q[0, 3:6, :].max(1) - q[0, 0:3, :].min(1)) * step[0]
    root_area = root_ext[0] * root_ext[1] + root_ext[1] * root_ext[2] + root_ext[2] * root_ext[0]
    return 1.0 + (area * inner).sum() / root_area, (area * leaf * cnt).sum() / root_area


for wl in (sys.argv[1:] or ["conference*", "bunny*", "sportsCar*"]):
    sd = scenes.WORKLOADS[wl]()
    t0 = time.time()
    b = capi.bvh_build_host(sd.pos, 4, getattr(sd, "ptype", None))
    n, t = cost(b["nodes8q"])
    print(f"{wl}: {len(b['nodes8q'])} nodes, depth {b['depth8']}, node cost {n:.2f}, primitive cost {t:.2f}, build {time.time() - t0:.2f} s")
