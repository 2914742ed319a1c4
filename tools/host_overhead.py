#!/usr/bin/env python3
"""Where the host time of a small frame goes: wall time per step of the bench loop vs the library's own wall time
(mirt_stats.total_ms) vs the GPU time between the frame's first and last event (render_ms), plus the tonemap call.

    python3 tools/host_overhead.py [spp]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from raytracer_amd import capi, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w, h = 1920, 1080
capi.init(0)
sd = scenes.WORKLOADS["conference*"]()
sd.camera = sd.camera.for_aspect(w, h)
scene = capi.Scene(sd)
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
film = np.zeros((h, w, 3), np.int32)
kw = dict(spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=8, rng_mode=1, traversal=0, streams=4 if spp < 16 else 8, profile_closest=True)
rows = []
for i in range(8):
    t0 = time.perf_counter()
    st = scene.render_accum(w, h, accum.data_ptr(), **kw)
    t1 = time.perf_counter()
    capi.tonemap(accum.data_ptr(), w, h, out=film)
    t2 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, st["total_ms"], st["render_ms"], (t2 - t1) * 1e3))
for r in rows[2:]:
    print("render_accum call %.3f ms | inside library %.3f | GPU first..last event %.3f | tonemap call %.3f" % r)
