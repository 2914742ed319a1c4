#!/usr/bin/env python3
"""A 64x64, 1 spp render: every launch is nearly empty, so the kernel trace shows the per-iteration fixed costs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import capi, scenes
capi.init(0)
sd = scenes.cornell_scene(2, "diffuse")
scene = capi.Scene(sd)
accum = torch.zeros((64, 64, 4), dtype=torch.float32, device="cuda:0")
for _ in range(3):
    st = scene.render_accum(64, 64, accum.data_ptr(), spp=1, spl=1, batch=2, tol=0.0, depth=8, rng_mode=1)
print(st["iterations"], st["render_ms"])
