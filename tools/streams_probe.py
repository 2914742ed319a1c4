#!/usr/bin/env python3
"""Frame time against mirt_options.streams on the bench workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import capi, scenes
w, h, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 64
capi.init(0)
sd = scenes.WORKLOADS["conference*"]()
sd.camera = sd.camera.for_aspect(w, h)
scene = capi.Scene(sd)
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
kw = dict(spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=8, rng_mode=1, traversal=0)
for S in (1, 2, 4, 8, 16):
    scene.render_accum(w, h, accum.data_ptr(), streams=S, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = scene.render_accum(w, h, accum.data_ptr(), streams=S, **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print(f"streams {S:2d}: {ms:7.1f} ms  {st['iterations']:4d} iterations  {(st['closest_rays'] + st['shadow_rays']) / ms / 1e3:7.0f} Mrays/s", flush=True)
