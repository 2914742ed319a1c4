#!/usr/bin/env python3
"""Ray statistics of a frame in the reference app's regime (dense cloud of small spheres seen from 350 units)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from raytracer_amd import capi  # noqa: E402
from soak_spheres import cloud  # noqa: E402

capi.init(0)
sc = cloud(500000, [0.2], 3, spread=18.0)
s = capi.Scene(sc, builder="ploc")
s.render(800, 600, spp=16, depth=8, batch=64, tol=0.05)
st = s.last
print({k: st[k] for k in ("closest_rays", "shadow_rays", "exact_fallback_rays", "iterations", "render_ms", "closest_node_visits",
                          "closest_tri_tests")})
print("fallback share of all rays %.3f" % (st["exact_fallback_rays"] / (st["closest_rays"] + st["shadow_rays"])))
