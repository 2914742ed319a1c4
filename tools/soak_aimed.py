#!/usr/bin/env python3
"""Aimed-ray soak: rays aimed exactly at vertices and edge midpoints (tests/gpu_common.aimed_rays), many seeds, ordered
traversal (0) against the reference walk (2, itself checked against the CPU oracle by tests/test_gpu_configs.py), closest hit
and occlusion.  Any difference is a bug (or the residual hole DESIGN.md 4 names).

    python3 tools/soak_aimed.py [rays_per_seed [seeds [aimed|grazing]]]

`grazing`: tests/gpu_common.grazing_rays -- rays almost in the plane of a triangle, the regime of the known class of exceptions.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from gpu_common import aimed_rays, grazing_rays  # noqa: E402
from raytracer_amd import capi, scenes  # noqa: E402
from test_gpu_configs import _open_meshes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
make = grazing_rays if (len(sys.argv) > 3 and sys.argv[3] == "grazing") else aimed_rays
capi.init(0)
cases = dict(_open_meshes())
cases["blob64"] = scenes.blob_scene(64, 32)
cases["sphere_soup"] = scenes.sphere_soup(300, 60, 3)
cases["conference*"] = scenes.WORKLOADS["conference*"]()
cases["sportsCar*"] = scenes.WORKLOADS["sportsCar*"]()
bad = 0
for name, sc in cases.items():
    s = capi.Scene(sc)
    diff = tot = hits = 0
    for seed in range(seeds):
        rays = make(sc, n, 1000 + seed)
        ri, rt = s.trace(rays, traversal=2)
        gi, gt = s.trace(rays, traversal=0)
        ra, _ = s.trace(rays, anyhit=True, traversal=2)
        ga, _ = s.trace(rays, anyhit=True, traversal=0)
        diff += int((ri != gi).sum() + (rt != gt).sum() + (ra != ga).sum())
        for j in np.nonzero((ri != gi) | (rt != gt) | (ra != ga))[0][:4]:
            print(f"   seed {1000 + seed} ray {j}: trav0 {gi[j]} {gt[j].view(np.float32) if hasattr(gt[j], 'view') else gt[j]} any {ga[j]} | "
                  f"reference {ri[j]} {rt[j]} any {ra[j]} | {rays[j].tolist()}", flush=True)
        tot += 2 * len(rays)
        hits += int((ri >= 0).sum())
    bad += diff
    print(f"{name:14s} {sc.n:8d} prims, {tot:9d} aimed queries: {diff} differences, hit rate {hits / (tot / 2):.2f}", flush=True)
    s.close()
print("TOTAL differences:", bad)
sys.exit(1 if bad else 0)
