#!/usr/bin/env python3
"""Share pool against sample streams at full size on every BASELINE workload: the float accumulation buffer of
`streams = S, shares = W` must equal that of `streams = W` bit for bit (same shares, same order of summation), with the same
ray counts, whatever the home-group size.   python3 tools/soak_pool.py  > profiles/r3b_pool_soak.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from raytracer_amd import capi, scenes  # noqa: E402

capi.init(0)
w, h = 1920, 1080
bad = 0
for name in ("bunny*", "buddha*", "conference*", "sportsCar*"):
    sd = scenes.WORKLOADS[name]()
    sd.camera = sd.camera.for_aspect(w, h)
    scene = capi.Scene(sd)
    for spp, W in ((16, 16), (12, 4)):
        kw = dict(spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=8, rng_mode=1, seed=5, traversal=0)
        ref = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
        st0 = dict(scene.render_accum(w, h, ref.data_ptr(), streams=W, **kw))
        for S, group in ((1, 0), (2, 256), (4, 0), (8, 4096)):
            if S >= W:
                continue
            acc = torch.zeros_like(ref)
            st = dict(scene.render_accum(w, h, acc.data_ptr(), streams=S, shares=W, pool_group=group, **kw))
            diff = int((acc != ref).any(dim=-1).sum().item())
            same_rays = all(st[k] == st0[k] for k in ("closest_rays", "shadow_rays", "paths"))
            bad += diff + (0 if same_rays else 1)
            print(f"{name:12s} {spp:3d} spp, {W:2d} shares, {S} slots per pixel, group {group or 64:5d}: {diff} pixels differ, ray counts "
                  f"{'equal' if same_rays else 'DIFFER'} ({st['closest_rays'] + st['shadow_rays']} rays), iterations {st['iterations']} "
                  f"(streams = {W}: {st0['iterations']})", flush=True)
    scene.close()
print("TOTAL differences:", bad)
sys.exit(1 if bad else 0)
