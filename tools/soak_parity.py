#!/usr/bin/env python3
"""Parity soak: millions of arbitrary rays per scene and builder, ordered traversal (0) against the reference walk (2, itself
checked against the CPU oracle by tests/test_gpu_parity.py).  Any difference in hit index, t bits or occlusion is a bug.

    python3 tools/soak_parity.py [rays_per_case]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from gpu_common import random_rays  # noqa: E402
from raytracer_amd import capi, scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
capi.init(0)
cases = {
    "cornell_mixed": lambda: scenes.cornell_scene(2, "mixed"),
    "conf20k": lambda: scenes.conference_scene(20000),
    "soup500": lambda: scenes.random_soup(500, 1),
    "soup5000": lambda: scenes.random_soup(5000, 7),
    "sphere_soup": lambda: scenes.sphere_soup(300, 60, 3),
    "blob64": lambda: scenes.blob_scene(64, 32),
    "conference*": scenes.WORKLOADS["conference*"],
    "sportsCar*": scenes.WORKLOADS["sportsCar*"],
}
bad = 0
for name, make in cases.items():
    sc = make()
    for builder in ("host", "device", "ploc"):
        s = capi.Scene(sc, builder=builder)
        diff = 0
        for unit in (False, True):
            rays = random_rays(sc, n // 2, 100 + unit, unit)
            i2, t2 = s.trace(rays, traversal=2)
            i0, t0 = s.trace(rays, traversal=0)
            o2, _ = s.trace(rays, anyhit=True, traversal=2)
            o0, _ = s.trace(rays, anyhit=True, traversal=0)
            diff += int((i0 != i2).sum() + (t0 != t2).sum() + (o0 != o2).sum())
        print(f"{name:14s} {builder:6s} {sc.n:7d} prims, {n} rays: {diff} differences, hit rate {(i2 >= 0).mean():.2f}", flush=True)
        bad += diff
        s.close()
print("TOTAL differences:", bad)
sys.exit(1 if bad else 0)
