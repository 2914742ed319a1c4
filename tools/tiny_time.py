#!/usr/bin/env python3
"""Wall time of a tiny render (64x64, 1 spp, cornell): per-iteration fixed costs dominate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import capi, scenes
capi.init(0)
sd = scenes.cornell_scene(2, "diffuse")
scene = capi.Scene(sd)
accum = torch.zeros((64, 64, 4), dtype=torch.float32, device="cuda:0")
kw = dict(spp=1, spl=1, batch=2, tol=0.0, depth=8, rng_mode=1)
for _ in range(3): st = scene.render_accum(64, 64, accum.data_ptr(), **kw)
t0 = time.perf_counter()
for _ in range(20): st = scene.render_accum(64, 64, accum.data_ptr(), **kw)
print(f"{(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per render, {st['iterations']} iterations, device {st['render_ms']:.3f} ms")
