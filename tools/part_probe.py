#!/usr/bin/env python3
"""What one rank of an N-GPU frame costs under each partition (run on one GPU): ms for rank 0's share."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import capi, scenes
w, h, spp = 1920, 1080, 64
capi.init(0)
sd = scenes.WORKLOADS["conference*"]()
sd.camera = sd.camera.for_aspect(w, h)
scene = capi.Scene(sd)
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
kw = dict(spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=8, rng_mode=1, traversal=0)
def run(**opt):
    scene.render_accum(w, h, accum.data_ptr(), **kw, **opt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = scene.render_accum(w, h, accum.data_ptr(), **kw, **opt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, st
full, st = run()
print(f"full frame {full:.1f} ms, {st['iterations']} iterations")
for n in (2, 4, 8):
    t_tile, s1 = run(tile_rank=0, tile_world=n)
    per = spp // n
    t_spp, s2 = run(sample_begin=0, sample_count=per)
    print(f"N={n}: tile {t_tile:.1f} ms ({s1['iterations']} it, eff {full / n / t_tile:.2f})   spp {t_spp:.1f} ms ({s2['iterations']} it, eff {full / n / t_spp:.2f})")
