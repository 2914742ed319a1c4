#!/usr/bin/env python3
"""Rank 0's share of an 8-way tile partition, for kernel-trace statistics of small launches."""
import os, sys
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
st = scene.render_accum(w, h, accum.data_ptr(), spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=8, rng_mode=1, traversal=0, tile_rank=0, tile_world=8)
torch.cuda.synchronize()
print(st["render_ms"], st["iterations"])
