#!/usr/bin/env python3
"""Lane use of the two loop parts of k_trace_fast from the counting build (mirt_stats, flags bit 1): visits per ray, and
visits / (64 x wave-level steps) = share of the 64 lanes that took part in a step.  python3 tools/lane_stats.py [spp] [streams] [shares]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from raytracer_amd import capi, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 8
shares = int(sys.argv[3]) if len(sys.argv) > 3 else 0
w, h = 1920, 1080
capi.init(0)
sd = scenes.WORKLOADS[os.environ.get("WORKLOAD", "conference*")]()
sd.camera = sd.camera.for_aspect(w, h)
scene = capi.Scene(sd)
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
st = scene.render_accum(w, h, accum.data_ptr(), spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=8, rng_mode=1, traversal=0,
                        streams=streams, shares=shares, count=True)
for kind, rays in (("closest", st["closest_rays"]), ("shadow", st["shadow_rays"])):
    nv, tv, ns, ts = (st[f"{kind}_node_visits"], st[f"{kind}_tri_tests"], st[f"{kind}_node_steps"], st[f"{kind}_tri_steps"])
    print(f"{kind:8s}: {rays / 1e6:7.1f} M rays, {nv / rays:.2f} node visits + {tv / rays:.2f} primitive tests per ray; "
          f"lanes per node step {nv / (64 * ns):.3f}, per primitive step {tv / (64 * ts):.3f}; "
          f"wave steps per 64 rays: node {64 * ns / rays:.2f}, primitive {64 * ts / rays:.2f}")
print("iterations", st["iterations"])
