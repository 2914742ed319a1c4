"""Diagnostic: many overlapping emissive spheres (the particle app's early frames): films and hits per builder vs the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cpulibs import CpuTracer
from raytracer_amd import capi, scenes
capi.init(0)
rng = np.random.default_rng(5)
n = int(os.environ.get("N", 20000))
pos = np.zeros((n + 2, 9), np.float32)
d = rng.normal(size=(n, 3)); d[:, 1] = np.abs(d[:, 1]); d /= np.linalg.norm(d, axis=1, keepdims=True)
pos[:n, :3] = d * rng.uniform(0, 18, (n, 1))
pos[:n, 3] = 0.2
pos[n] = [-5000, 0, -5000, 5000, 0, -5000, 5000, 0, 5000]
pos[n + 1] = [-5000, 0, -5000, -5000, 0, 5000, 5000, 0, 5000]
nrm = np.zeros((n + 2, 9), np.float32); nrm[n:] = [0, 1, 0] * 3
mats = np.concatenate([[scenes.WHITE], [scenes.emission(*rng.uniform(0.2, 1, 3)) for _ in range(n)]])
mat = np.concatenate([np.arange(1, n + 1), [0, 0]]).astype(np.uint32)
sc = scenes.SceneData("p", pos, nrm, mat, mats, np.array([scenes.directional_light((.2, .2, .2), (0, -1, 0))]),
                      scenes.Camera((0, 175, -300), (0, 20, 0), hfov=42, vfov=31),
                      ptype=np.concatenate([np.full(n, 1, np.uint8), np.full(2, 2, np.uint8)])).finalize()
o = CpuTracer("oracle").load(sc)
w, h = 200, 150
oi, ot = o.primary_hits(w, h)
ref = o.render(w, h, spp=8, depth=8, batch=64, tol=0.05, nthreads=8)
for b in ("host", "device", "ploc"):
    s = capi.Scene(sc, builder=b)
    for trav in (0, 1):
        gi, gt = s.primary_hits(w, h, traversal=trav)
        img = s.render(w, h, spp=8, depth=8, batch=64, tol=0.05, traversal=trav)
        print(b, "trav", trav, "hits differ", int((gi != oi).sum()), int((gt != ot).sum()), "film same", float((img == ref).mean()),
              "rays", s.last["closest_rays"], s.last["shadow_rays"], "oracle", o.last["closest"], o.last["shadow"], "redo", s.last["exact_fallback_rays"])
    s.close()
