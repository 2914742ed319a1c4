#!/usr/bin/env python3
"""One rank's share of an N-rank frame under the TILE partition (all 64 spp on 1/N of the tiles), timed on one GPU, beside the spp
partition's share (64/N spp on every pixel): which decomposition leaves a rank the shorter frame?

    python3 tools/tile_share.py [N=8]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from raytracer_amd import capi, scenes  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w, h, spp = 1920, 1080, 64
capi.init(0)
sd = scenes.WORKLOADS["conference*"]()
sd.camera = sd.camera.for_aspect(w, h)
scene = capi.Scene(sd)
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")


def timed(label, **kw):
    best = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = scene.render_accum(w, h, accum.data_ptr(), spl=1, tol=0.0, depth=8, rng_mode=1, traversal=0, **kw)
        torch.cuda.synchronize()
        best.append(1e3 * (time.perf_counter() - t0))
    rays = st["closest_rays"] + st["shadow_rays"]
    print(f"{label:<58} {sorted(best)[1]:7.2f} ms  {rays / 1e6:7.1f} Mrays  iterations {st['iterations']}", flush=True)


timed("whole frame, 8 slots, pool of 64", spp=spp, batch=spp + 1, streams=8, shares=64)
timed(f"spp partition: {spp // N} spp, 4 slots, pool of {spp // N}", spp=spp // N, batch=spp, streams=4, shares=spp // N)
for streams in (8, 16):                                       # (16 path slots per pixel is the library's limit)
    for r in (0, N // 2):
        timed(f"tile partition: rank {r} of {N}, 64 spp, {streams} slots, pool of 64", spp=spp, batch=spp + 1, streams=streams, shares=64,
              tile_rank=r, tile_world=N)
