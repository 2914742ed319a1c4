#!/usr/bin/env python3
"""One frame of the bench workload at reduced spp, with no per-launch events: the program rocprofv3 wraps for PMC passes.

    rocprofv3 --pmc <counters> -d gpurun_out/pmc_x -o x --output-format csv -- python3 tools/prof_frame.py [spp] [w] [h] [streams] [shares]     (WORKLOAD=bunny* ... picks another scene)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from raytracer_amd import capi, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
streams = int(sys.argv[4]) if len(sys.argv) > 4 else 1
shares = int(sys.argv[5]) if len(sys.argv) > 5 else 0
capi.init(0)
sd = scenes.WORKLOADS[os.environ.get("WORKLOAD", "conference*")]()
sd.camera = sd.camera.for_aspect(w, h)
scene = capi.Scene(sd)
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
st = scene.render_accum(w, h, accum.data_ptr(), spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=8, rng_mode=1, traversal=0,
                         streams=streams, shares=shares)
torch.cuda.synchronize()
print({k: st[k] for k in ("closest_rays", "shadow_rays", "iterations", "render_ms", "exact_fallback_rays")})
