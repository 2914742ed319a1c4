#!/usr/bin/env python3
"""The reference app's frame loop (apps/app.cu:24-30: every frame set_scene -> BVH build -> raytrace) on a moving scene:
scene creation + render + destroy per frame, for each builder of the traversal's own tree.

    python3 tools/frame_loop.py [workload] [frames] [spp] [w] [h]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from raytracer_amd import capi, scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "conference*"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
w = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
h = int(sys.argv[5]) if len(sys.argv) > 5 else 1080
capi.init(0)
sd = scenes.WORKLOADS[wl]()
sd.camera = sd.camera.for_aspect(w, h)
base = sd.pos.copy()
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
for builder in ("host", "device", "ploc"):
    t_create = t_render = 0.0
    rays = 0
    for f in range(frames + 1):
        sd.pos = base.copy()
        sd.pos.reshape(-1, 3)[:, 1] *= np.float32(1.0 + 0.01 * np.sin(0.3 * f))        # the scene breathes: new boxes every frame
        t0 = time.perf_counter()
        scene = capi.Scene(sd, builder=builder)
        t1 = time.perf_counter()
        st = scene.render_accum(w, h, accum.data_ptr(), spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=8, rng_mode=1, streams=4)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        scene.close()
        if f:                                                                       # frame 0 warms up
            t_create += t1 - t0
            t_render += t2 - t1
            rays += st["closest_rays"] + st["shadow_rays"]
    print(f"{wl} {sd.n} prims, {w}x{h}, {spp} spp, builder {builder}: create {1e3 * t_create / frames:.1f} ms + render "
          f"{1e3 * t_render / frames:.1f} ms per frame = {frames / (t_create + t_render):.2f} frames/s, "
          f"{rays / (t_create + t_render) / 1e6:.0f} Mrays/s with the build inside the clock")
