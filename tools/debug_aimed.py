"""Diagnostic: the rays of tests/test_gpu_configs.py::test_rays_aimed_at_vertices_and_edges on which traversal 0 and the
oracle disagree, with what the brute-force walk (one leaf holding every primitive) says."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cpulibs import CpuTracer
from gpu_common import aimed_rays
from raytracer_amd import capi
from test_gpu_configs import _open_meshes
np.set_printoptions(precision=9, linewidth=200)
capi.init(0)
o, b = CpuTracer("oracle"), CpuTracer("oracle")
for key, sc in _open_meshes().items():
    o.load(sc); b.load(sc, leaf_size=10 ** 7)
    s = capi.Scene(sc)
    rays = aimed_rays(sc, int(os.environ.get("N", 120000)), int(os.environ.get("SEED", 17)))
    ci, ct = o.trace(rays); bi, bt = b.trace(rays)
    gi, gt, gb = s.trace(rays, traversal=0, bary=True)
    bad = np.nonzero((gi != ci) | (gt != ct))[0]
    ai, _ = o.trace(rays, anyhit=True); hi, _ = s.trace(rays, anyhit=True, traversal=0)
    print(key, "closest differ", len(bad), "any differ", int((hi != ai).sum()), "oracle-vs-brute differ", int((ct != bt).sum()))
    nodes, flat = s.bvh2()
    for i in bad[:5]:
        print("  ray", i, rays[i])
        print("   gpu idx", gi[i], "t", gt[i].view(np.float32) if hasattr(gt[i], 'view') else gt[i], "bary", gb[i], " oracle idx", ci[i], "t", np.array([ct[i]], np.uint32).view(np.float32)[0],
              " brute t", np.array([bt[i]], np.uint32).view(np.float32)[0], "gpu t", np.array([gt[i]], np.uint32).view(np.float32)[0])
        if gi[i] >= 0:
            tri = sc.pos[flat[gi[i]]].reshape(3, 3)
            print("   gpu tri", tri.tolist())
            hp = rays[i, :3] + rays[i, 3:6] * np.array([gt[i]], np.uint32).view(np.float32)[0]
            print("   hit point", hp, "dist to verts", np.linalg.norm(tri - hp, axis=1))
        if ci[i] >= 0:
            print("   oracle tri", sc.pos[flat[ci[i]]].reshape(3, 3).tolist())
    s.close()
