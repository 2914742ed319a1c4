#!/usr/bin/env python3
"""One rank's share of an 8-GPU frame: pure spp slice (8 spp, all pixels) against tile x spp hybrids (run on one GPU)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from raytracer_amd import capi, scenes  # noqa: E402

capi.init(0)
w, h = 1920, 1080
sd = scenes.WORKLOADS["conference*"]()
sd.camera = sd.camera.for_aspect(w, h)
scene = capi.Scene(sd)
acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
for tiles, spp in ((1, 8), (2, 16), (4, 32), (8, 64)):
    kw = dict(spp=64, spl=1, batch=65, tol=0.0, depth=8, rng_mode=1, streams=4, tile_rank=0, tile_world=tiles,
              sample_begin=0, sample_count=spp)
    scene.render_accum(w, h, acc.data_ptr(), **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        st = scene.render_accum(w, h, acc.data_ptr(), **kw)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print(f"tile groups {tiles} x spp groups {8 // tiles}: share = 1/{tiles} of the tiles at {spp} spp: {ms:.2f} ms, "
          f"{(st['closest_rays'] + st['shadow_rays']) / 1e6:.0f} Mrays")
