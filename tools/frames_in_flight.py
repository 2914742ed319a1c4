#!/usr/bin/env python3
"""Frames in flight: F host threads, each with its own library context (thread-local: own streams, own control blocks) and
its own scene, render alternate frames of a frame loop on ONE GPU, so that the shrinking launches at the end of one frame
run beside the full launches of the next.  Throughput of K frames (film hand-over included) against the plain loop.

    python3 tools/frames_in_flight.py [spp=64] [frames=12]
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from raytracer_amd import capi, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
w, h = 1920, 1080
streams = 8 if spp > 32 else 4
shares = spp if spp <= 64 else 0
sd = scenes.WORKLOADS["conference*"]()
sd.camera = sd.camera.for_aspect(w, h)


class Worker:
    def __init__(self):
        self.ready = threading.Event()
        self.go = threading.Event()
        self.done = threading.Event()
        self.frames = 0
        self.rays = 0
        self.th = threading.Thread(target=self.run, daemon=True)
        self.th.start()

    def run(self):
        torch.cuda.set_device(0)
        capi.init(0)                                   # this thread's own library context
        scene = capi.Scene(sd)
        accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
        film = np.zeros((h, w, 3), np.int32)
        kw = dict(spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=8, rng_mode=1, traversal=0, streams=streams, shares=shares)
        scene.render_accum(w, h, accum.data_ptr(), **kw)          # warm-up
        capi.tonemap(accum.data_ptr(), w, h, out=film)
        while True:
            self.ready.set()
            self.go.wait()
            self.go.clear()
            if self.frames < 0:
                return
            rays = 0
            for _ in range(self.frames):
                st = scene.render_accum(w, h, accum.data_ptr(), **kw)
                capi.tonemap(accum.data_ptr(), w, h, out=film)
                rays += st["closest_rays"] + st["shadow_rays"]
            self.rays = rays
            self.checksum = int(film.sum())
            self.done.set()


def loop(workers, frames_each):
    for wk in workers:
        wk.ready.wait()
        wk.ready.clear()
        wk.frames = frames_each
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for wk in workers:
        wk.go.set()
    for wk in workers:
        wk.done.wait()
        wk.done.clear()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = frames_each * len(workers)
    rays = sum(wk.rays for wk in workers)
    print(f"spp {spp:3d}: {len(workers)} in flight, {n} frames: {1e3 * dt / n:7.2f} ms per frame, {rays / dt / 1e6:7.0f} Mrays/s, "
          f"film checksums {sorted(set(wk.checksum for wk in workers))}", flush=True)


ws = [Worker() for _ in range(3)]
for rep in range(2):
    loop(ws[:1], K)
    loop(ws[:2], K // 2)
    loop(ws[:3], K // 3)
