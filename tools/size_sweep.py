#!/usr/bin/env python3
"""Trace-kernel throughput against scene size (working set vs the 4 MiB per-XCD L2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import capi, scenes
w, h, spp = 1920, 1080, 8
capi.init(0)
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
for n in [int(x) for x in (sys.argv[1:] or ["5000", "20000", "80000", "331000", "1300000"])]:
    sd = scenes.conference_scene(n)
    sd.camera = sd.camera.for_aspect(w, h)
    scene = capi.Scene(sd)
    kw = dict(spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=8, rng_mode=1, traversal=0)
    scene.render_accum(w, h, accum.data_ptr(), **kw)
    st = scene.render_accum(w, h, accum.data_ptr(), profile=True, **kw)
    ct = scene.render_accum(w, h, accum.data_ptr(), count=True, **kw)
    cr, sr = st["closest_rays"], st["shadow_rays"]
    print(f"tris {sd.n:8d}  frame {st['render_ms']:7.2f} ms  closest {cr / st['trace_closest_ms'] / 1e6:6.2f} Grays/s ({ct['closest_node_visits'] / cr:.2f} nodes {ct['closest_tri_tests'] / cr:.2f} tris)"
          f"  shadow {sr / st['trace_shadow_ms'] / 1e6:6.2f} Grays/s ({ct['shadow_node_visits'] / sr:.2f} nodes {ct['shadow_tri_tests'] / sr:.2f} tris)  logic {st['logic_ms']:.1f} ms", flush=True)
    del scene
