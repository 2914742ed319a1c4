#!/usr/bin/env python3
"""Sphere-regime soak (the reference app: spheres of radius 0.2 seen from hundreds of units away, where the reference's sphere
test cancels catastrophically): ordered traversal (0) against the reference walk (2) on camera-like rays from far away, rays
aimed at sphere centres and rims, and rays starting on spheres; closest hit and occlusion.  Mixed radii too.

    python3 tools/soak_spheres.py [rays_per_case]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from raytracer_amd import capi, scenes  # noqa: E402



def cloud(n, radii, seed, spread=18.0):
    rng = np.random.default_rng(seed)
    pos = np.zeros((n + 2, 9), np.float32)
    d = rng.normal(size=(n, 3))
    d[:, 1] = np.abs(d[:, 1])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pos[:n, :3] = d * rng.uniform(0, spread, (n, 1))
    pos[:n, 3] = rng.choice(radii, n)
    pos[n] = [-5000, 0, -5000, 5000, 0, -5000, 5000, 0, 5000]
    pos[n + 1] = [-5000, 0, -5000, -5000, 0, 5000, 5000, 0, 5000]
    nrm = np.zeros((n + 2, 9), np.float32)
    nrm[n:] = [0, 1, 0] * 3
    mats = np.array([scenes.WHITE])
    return scenes.SceneData("cloud", pos, nrm, np.zeros(n + 2, np.uint32), mats,
                            np.array([scenes.directional_light((.2, .2, .2), (0, -1, 0))]),
                            scenes.Camera((0, 175, -300), (0, 20, 0), hfov=42, vfov=31),
                            ptype=np.concatenate([np.full(n, 1, np.uint8), np.full(2, 2, np.uint8)])).finalize()


def rays_for(sc, n, seed, dist):
    rng = np.random.default_rng(seed)
    f = np.float32
    ns = sc.n - 2
    c = sc.pos[:ns, :3]
    rad = sc.pos[:ns, 3]
    r = np.zeros((n, 8), f)
    k = rng.integers(0, ns, n)
    # origins: a shell `dist` away (camera-like), or on the surface of another sphere (bounce-like)
    o = rng.normal(size=(n, 3))
    o /= np.linalg.norm(o, axis=1, keepdims=True)
    o[:, 1] = np.abs(o[:, 1]) * 0.6 + 0.05
    o = (o * dist * rng.uniform(0.5, 1.5, (n, 1))).astype(f)
    j = rng.integers(0, ns, n)
    on = rng.normal(size=(n, 3))
    on /= np.linalg.norm(on, axis=1, keepdims=True)
    surf = (c[j] + on * (rad[j, None] * 1.01)).astype(f)
    use_surf = rng.random(n) < 0.3
    o[use_surf] = surf[use_surf]
    # targets: centre, a point on the rim (tangent-ish), or a random point near the sphere
    off = rng.normal(size=(n, 3))
    off /= np.linalg.norm(off, axis=1, keepdims=True)
    scale = rng.choice([0.0, 0.5, 0.98, 1.0, 1.02, 1.5], n)[:, None] * rad[k, None]
    tgt = (c[k] + off * scale).astype(f)
    d = (tgt - o).astype(f)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(f)
    r[:, :3] = o
    r[:, 3:6] = d.astype(f)
    r[:, 6] = 1e-5
    r[:, 7] = np.inf
    return r


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    capi.init(0)
    bad = 0
    for name, sc, dist in (("r=0.2 from 350", cloud(20000, [0.2], 5), 350.0), ("r=0.2 from 3000", cloud(20000, [0.2], 6), 3000.0),
                           ("mixed radii 0.01-2 from 350", cloud(20000, [0.01, 0.05, 0.2, 2.0], 7), 350.0),
                           ("200k spheres from 350", cloud(200000, [0.2], 8, spread=40.0), 350.0),
                           ("r=0.2 from 20", cloud(20000, [0.2], 9), 20.0)):
        for builder in ("host", "ploc"):
            s = capi.Scene(sc, builder=builder)
            diff = tot = hits = 0
            for seed in range(max(1, N // 500000)):
                rays = rays_for(sc, min(N, 500000), 100 + seed, dist)
                ri, rt = s.trace(rays, traversal=2)
                gi, gt = s.trace(rays, traversal=0)
                ra, _ = s.trace(rays, anyhit=True, traversal=2)
                ga, _ = s.trace(rays, anyhit=True, traversal=0)
                diff += int(((ri != gi) | (rt != gt)).sum() + (ra != ga).sum())
                tot += 2 * len(rays)
                hits += int((ri >= 0).sum())
            bad += diff
            print(f"{name:30s} {builder:6s} {tot:9d} queries: {diff} differences, hit rate {hits / (tot / 2):.2f}", flush=True)
            s.close()
    print("TOTAL differences:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
