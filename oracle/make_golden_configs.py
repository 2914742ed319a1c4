#!/usr/bin/env python3
"""Golden vectors for the BASELINE configurations, dumped from the REFERENCE ITSELF (oracle/_ref/libref*.so = the reference's
own headers compiled with g++ where /root/reference is mounted) into tests/golden/configs.npz.  Data only.

    make -C oracle ref && python3 oracle/make_golden_configs.py

  cfg1        BASELINE configs[0] at its own size: bunny* (69 172 triangles) 256x256, 4 spp, depth 4, diffuse -- the one
              configuration small enough for a full-size film from the reference (0.9 s of CPU): film + primary hits
  sportscar   configs[4] material set (all 37 materials of objects/sportsCar.mtl: glass ior 1, mirror, microfacet, emitters)
              on a 6 000-triangle cut of the stand-in, 96x54, 4 spp, depth 8: film (both abs() variants) + primary hits
  thinlens    scene/camera.cuh:46-60 on the all-BSDF Cornell scene, aperture 0.05: film + primary hits
  mixed_a/b   float radiance of the all-BSDF Cornell scene, two independent reference runs at 256 spp (rRMSE bar)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cpulibs import GOLDEN, CpuTracer  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

R = CpuTracer("ref")
RF = CpuTracer("ref", "_fabs")
out = {}


def film(key, sc, w, h, both=True, **kw):
    R.load(sc)
    RF.load(sc)
    idx, tb = R.primary_hits(w, h)
    out[key + "_hit_idx"] = idx
    out[key + "_hit_tbits"] = tb
    out[key + "_rgb_fabs"] = RF.render(w, h, **kw).astype(np.uint8)      # film values are 0..255
    if both:
        out[key + "_rgb"] = R.render(w, h, **kw).astype(np.uint8)
    out[key + "_params"] = np.array([w, h, kw.get("spp", 4), kw.get("spl", 1), kw.get("batch", 64), kw.get("depth", 4)], np.int64)
    out[key + "_tol"] = np.float32(kw.get("tol", 0.05))
    print(key, sc.n, "primitives,", (idx >= 0).mean(), "of the primary rays hit")


bunny = scenes.blob_scene(264, 132)
bunny.camera = bunny.camera.for_aspect(256, 256)
film("cfg1", bunny, 256, 256, both=False, spp=4, depth=4, batch=5, tol=0.0)

car = scenes.sportscar_scene(6000)
car.camera = car.camera.for_aspect(96, 54)
film("sportscar", car, 96, 54, spp=4, depth=8, batch=5, tol=0.0)

lens = scenes.cornell_scene(2, "mixed")
lens.camera = scenes.Camera((0, 1.5, -1.95), (0, 1.2, 0), hfov=70, vfov=55, aperture=0.05, focal=0.0)
film("thinlens", lens, 64, 48, spp=4, depth=5, batch=5, tol=0.0)

mixed = scenes.cornell_scene(2, "mixed")
RF.load(mixed)
out["mixed_a"] = RF.radiance(48, 36, spp=256, depth=6, seed_xor=0, nthreads=8)
out["mixed_b"] = RF.radiance(48, 36, spp=256, depth=6, seed_xor=0x5bd1e995, nthreads=8)
out["mixed_params"] = np.array([48, 36, 256, 6], np.int64)

np.savez_compressed(os.path.join(GOLDEN, "configs.npz"), **out)
print({k: getattr(v, "shape", v) for k, v in out.items()})
