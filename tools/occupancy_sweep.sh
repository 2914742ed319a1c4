#!/bin/bash
# How the traversal kernels' throughput, each running ALONE (MIRT_SERIAL=1: one stream, no overlap), depends on the persistent
# workgroups resident per CU (4 waves each: n workgroups = n waves per SIMD).  The measured basis of the re-binning estimate in
# a scheme that keeps ray state in LDS trades resident waves for lanes that are all busy.
for b in 1 2 3 4 5 6 7; do
  MIRT_SERIAL=1 MIRT_BLOCKS_PER_CU=$b timeout -k 10 200 python3 - <<PY || exit 1
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from raytracer_amd import capi, scenes
capi.init(0)
sd = scenes.WORKLOADS["conference*"]()
w, h = 1920, 1080
sd.camera = sd.camera.for_aspect(w, h)
s = capi.Scene(sd)
acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
kw = dict(spp=64, spl=1, batch=65, tol=0.0, depth=8, rng_mode=1, streams=8)
s.render_accum(w, h, acc.data_ptr(), **kw)
st = s.render_accum(w, h, acc.data_ptr(), profile=True, **kw)
print("workgroups/CU $b: closest %.1f ms (%.2f Grays/s)  shadow %.1f ms (%.2f Grays/s)  logic %.1f ms  frame %.1f ms" % (
    st["trace_closest_ms"], st["closest_rays"] / st["trace_closest_ms"] / 1e6, st["trace_shadow_ms"],
    st["shadow_rays"] / st["trace_shadow_ms"] / 1e6, st["logic_ms"], st["render_ms"]), flush=True)
PY
done
