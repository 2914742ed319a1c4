"""Diagnostic: the aimed rays of tools/soak_aimed.py on the `cut` scene where traversal 0 and the reference walk disagree."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_common import aimed_rays
from raytracer_amd import capi
from test_gpu_configs import _open_meshes
np.set_printoptions(precision=9, linewidth=220, floatmode="unique")
capi.init(0)
sc = _open_meshes()[os.environ.get("KEY", "cut")]
s = capi.Scene(sc)
nodes, flat = s.bvh2()
f32 = lambda u: np.array([u], np.uint32).view(np.float32)[0]
for seed in range(1000, 1004):
    rays = aimed_rays(sc, 1000000, seed)
    ri, rt, rb = s.trace(rays, traversal=2, bary=True)
    gi, gt, gb = s.trace(rays, traversal=0, bary=True)
    g1, t1 = s.trace(rays, traversal=1)
    ra, _ = s.trace(rays, anyhit=True, traversal=2)
    ga, _ = s.trace(rays, anyhit=True, traversal=0)
    bad = np.nonzero((ri != gi) | (rt != gt) | (ra != ga))[0]
    print("seed", seed, "bad", bad)
    for i in bad:
        print("  ray", i, rays[i].tolist())
        print("   trav0 idx", gi[i], "t", f32(gt[i]), "bary", gb[i], "| trav2 idx", ri[i], "t", f32(rt[i]), "bary", rb[i], "| trav1 idx", g1[i], "t", f32(t1[i]), "| any t0/t2", ga[i], ra[i])
        for nm, idx in (("trav0", gi[i]), ("trav2", ri[i])):
            if idx >= 0:
                tri = sc.pos[flat[idx]].reshape(3, 3)
                print("   ", nm, "tri", tri.tolist())
s.close()
