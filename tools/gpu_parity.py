"""Diagnostic: HIP path vs the CPU oracle on a handful of scenes (run on the GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cpulibs import CpuTracer
from raytracer_amd import scenes, capi

capi.init(0)
O = CpuTracer("oracle")
O.set_glass_float_abs(1)
rng = np.random.default_rng(0)

def random_rays(sc, n):
    lo = sc.pos.reshape(-1, 3).min(0) - 0.5; hi = sc.pos.reshape(-1, 3).max(0) + 0.5
    o = rng.uniform(lo, hi, (n, 3)); d = rng.normal(size=(n, 3))
    r = np.zeros((n, 8), np.float32); r[:, :3] = o; r[:, 3:6] = d; r[:, 6] = 1e-5; r[:, 7] = np.inf
    r[: n // 8, 7] = rng.uniform(0.1, 3.0, n // 8)          # finite max_t
    r[n // 8: n // 4, 3 + 0] = 0.0                           # zero direction components -> reference-walk fallback
    r[n // 4: n // 4 + 50, 3:6] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 50)]   # axis aligned
    half = n // 2                                            # second half: unit directions, as the integrator makes them
    d = r[half:, 3:6].astype(np.float64)
    r[half:, 3:6] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return r

cases = [("cornell", scenes.cornell_scene(2), (96, 72), dict(spp=4, depth=6, spl=2)),
         ("soup", scenes.random_soup(2000, 1), (64, 64), dict(spp=2, depth=3)),
         ("blob", scenes.blob_scene(64, 32), (128, 128), dict(spp=4, depth=4, batch=5, tol=0)),
         ("conf", scenes.conference_scene(20000), (160, 90), dict(spp=4, depth=8))]
if len(sys.argv) > 1: cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, sc, (w, h), kw in cases:
    O.load(sc); G = capi.Scene(sc)
    print(f"== {name}: {sc.n} tris, {G.info()}")
    i0, t0 = O.primary_hits(w, h)
    for trav in (2, 1, 0):
        i1, t1 = G.primary_hits(w, h, traversal=trav)
        print(f"   primary hits trav={trav}: idx mismatches {(i0 != i1).sum()}  t mismatches {(t0 != t1).sum()} of {w*h}")
    rays = random_rays(sc, 20000)
    ci, ct = O.trace(rays); ai, at = O.trace(rays, anyhit=True)
    for trav in (2, 1, 0):
        gi, gt = G.trace(rays, traversal=trav); hi_, ht = G.trace(rays, anyhit=True, traversal=trav)
        print(f"   trace trav={trav}: closest idx {(ci != gi).sum()} t {(ct != gt).sum()} | any {(ai != hi_).sum()} (hits {(ci>=0).sum()}, occl {(ai>0).sum()})")
    t = time.time(); ref = O.render(w, h, nthreads=8, **kw); tc = time.time() - t
    for trav in (2, 0):
        img = G.render(w, h, traversal=trav, **kw)
        d = np.abs(img - ref)
        print(f"   render trav={trav}: exact {100*(d==0).mean():.3f}% <=1 {100*(d<=1).mean():.3f}% max {d.max()} mean {img.mean():.3f} vs {ref.mean():.3f}; rays gpu {G.last['closest_rays']}+{G.last['shadow_rays']} cpu {O.last['closest']}+{O.last['shadow']}; iters {G.last['iterations']} fallback {G.last['exact_fallback_rays']} gpu {G.last['render_ms']:.1f} ms cpu {tc*1e3:.0f} ms")
    G.close()
