#!/usr/bin/env python3
"""Traversal work and SIMD lane utilisation of one counting frame of the bench workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import capi, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w, h = 1920, 1080
capi.init(0)
sd = scenes.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "conference*"]()
sd.camera = sd.camera.for_aspect(w, h)
scene = capi.Scene(sd)
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
st = scene.render_accum(w, h, accum.data_ptr(), spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=8, rng_mode=1, traversal=0, count=True)
for k in ("closest", "shadow"):
    rays = st[f"{k}_rays"]; nv = st[f"{k}_node_visits"]; tt = st[f"{k}_tri_tests"]
    ns = st[f"{k}_node_steps"]; ts = st[f"{k}_tri_steps"]
    print(f"{k}: rays {rays}  nodes/ray {nv / rays:.2f}  tris/ray {tt / rays:.2f}  node-loop util {nv / (64 * ns):.3f}  tri-loop util {tt / (64 * ts):.3f}  "
          f"node steps/ray*64 {64 * ns / rays:.1f}  tri steps/ray*64 {64 * ts / rays:.1f}")
print("fallback rays", st["exact_fallback_rays"], "iterations", st["iterations"])
