import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from raytracer_amd import capi, scenes
capi.init(0)
sd = scenes.WORKLOADS["conference*"]()
sd.camera = sd.camera.for_aspect(1920, 1080)
for b in ("host", "ploc"):
    sc = capi.Scene(sd, builder=b)
    acc = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda:0")
    st = sc.render_accum(1920, 1080, acc.data_ptr(), spp=8, spl=1, batch=9, tol=0.0, depth=8, rng_mode=1, streams=4)
    print(b, os.environ.get("MIRT_SAH_PEEL"), "fallback rays", st["exact_fallback_rays"], "of", st["closest_rays"] + st["shadow_rays"], "ms", round(st["render_ms"], 2))
    sc.close()
