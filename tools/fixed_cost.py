#!/usr/bin/env python3
"""Per-call fixed costs of the boundary: a tiny render, the 1080p tonemap + film read-back."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import capi, scenes
capi.init(0)
sd = scenes.WORKLOADS["conference*"]()
sd.camera = sd.camera.for_aspect(1920, 1080)
scene = capi.Scene(sd)
accum = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda:0")
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
kw = dict(spl=1, tol=0.0, depth=8, rng_mode=1)
print("render 64x64 1 spp      %.3f ms" % t(lambda: scene.render_accum(64, 64, accum.data_ptr(), spp=1, batch=2, **kw)))
print("render 1080p 1 spp      %.3f ms" % t(lambda: scene.render_accum(1920, 1080, accum.data_ptr(), spp=1, batch=2, **kw)))
print("render 1080p 8 spp S=8  %.3f ms" % t(lambda: scene.render_accum(1920, 1080, accum.data_ptr(), spp=8, batch=9, streams=8, **kw)))
print("tonemap 1080p -> host   %.3f ms" % t(lambda: capi.tonemap(accum.data_ptr(), 1920, 1080)))
