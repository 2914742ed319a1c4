"""Scenes and ray sets shared by the GPU parity tests (tests/test_gpu_*.py)."""
import numpy as np

from raytracer_amd import scenes

SCENES = {
    "cornell_diffuse": lambda: scenes.cornell_scene(2, "diffuse"),
    "cornell_mixed": lambda: scenes.cornell_scene(2, "mixed"),
    "blob64": lambda: scenes.blob_scene(64, 32),
    "soup500": lambda: scenes.random_soup(500, 1),
    "conf20k": lambda: scenes.conference_scene(20000),
    "adaptive": lambda: scenes.cornell_scene(1, "diffuse"),
}
SPHERE_SCENES = {"sphere_soup": lambda: scenes.sphere_soup(300, 60, 3), "known_answer": scenes.known_answer_scene}


def film_delta(name, img, want, log="gpurun_out/r3_film_delta.txt"):
    """How a film differs from the reference's: fraction of differing bytes, how many of them are off by exactly one level
    (the device libm's sinf / cosf / acosf round the last ulp differently: a radiance an ulp away lands on the other side of
    a tonemap threshold), how many by more (a bounce direction an ulp away crossed a silhouette: another path, any delta), the
    largest delta.  Appends one line per call to `log` (copied to profiles/ by hand) and returns the numbers."""
    import os

    d = np.abs(np.asarray(img, np.int64) - np.asarray(want, np.int64))
    n = d.size
    st = {"bytes": n, "differ": int((d > 0).sum()), "by_one": int((d == 1).sum()), "by_more": int((d > 1).sum()), "max": int(d.max()),
          "pixels_by_more": int((d.reshape(-1, 3) > 1).any(1).sum())}
    try:
        os.makedirs(os.path.dirname(log), exist_ok=True)
        with open(log, "a") as fh:
            fh.write(f"{name:44s} {n:9d} bytes: {st['differ']:6d} differ ({st['differ'] / n:.2e}), {st['by_one']:6d} by one level, "
                     f"{st['by_more']:5d} by more ({st['by_more'] / n:.2e}; {st['pixels_by_more']} pixels), max {st['max']}\n")
    except OSError:
        pass
    return st


def random_rays(sc, n, seed, unit):
    rng = np.random.default_rng(seed)
    lo = sc.pos.reshape(-1, 3).min(0) - 0.5
    hi = sc.pos.reshape(-1, 3).max(0) + 0.5
    r = np.zeros((n, 8), np.float32)
    r[:, :3] = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    if unit:
        d /= np.linalg.norm(d, axis=1, keepdims=True)
    r[:, 3:6] = d
    r[:, 6] = 1e-5
    r[:, 7] = np.inf
    r[: n // 8, 7] = rng.uniform(0.1, 3.0, n // 8)                  # finite max_t
    r[n // 8: n // 4, 3] = 0.0                                       # zero direction component: 0 * inf slabs
    r[n // 4: n // 4 + 60, 3:6] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 60)] * rng.choice([-1, 1], (60, 1))
    return r


def aimed_rays(sc, n, seed):
    """Rays aimed exactly at triangle vertices and edge midpoints (targets and directions computed in float32), from
    axis-aligned and generic origins, unit directions.  A hit on a vertex or on an axis-parallel edge can sit on an EDGE of
    one of the reference's node boxes, where its slab test (bbox.cuh:67-79) has entry ~ exit and decides in the last ulp."""
    rng = np.random.default_rng(seed)
    f = np.float32
    tri = sc.pos.reshape(-1, 3, 3)
    if sc.ptype is not None:
        tri = tri[sc.ptype == 2]
    pick = rng.integers(0, len(tri), n)
    which = rng.integers(0, 6, n)                      # 0..2 vertex, 3..5 midpoint of the edge starting there
    a = tri[pick, which % 3]
    b = tri[pick, (which + 1) % 3]
    tgt = np.where((which >= 3)[:, None], (a + b) * f(0.5), a).astype(np.float32)
    kind = rng.integers(0, 3, n)                       # 0 axis-aligned origin, 1 generic origin, 2 generic, far away
    axis = np.eye(3, dtype=np.float32)[rng.integers(0, 3, n)] * rng.choice([-1, 1], (n, 1)).astype(np.float32)
    gen = rng.normal(size=(n, 3)).astype(np.float32)
    gen /= np.linalg.norm(gen, axis=1, keepdims=True).astype(np.float32)
    dist = np.where(kind == 2, rng.uniform(5, 40, n), rng.uniform(0.3, 3, n)).astype(np.float32)[:, None]
    org = (tgt - np.where((kind == 0)[:, None], axis, gen) * dist).astype(np.float32)
    d = (tgt - org).astype(np.float32)
    ln = np.sqrt((d * d).sum(1, dtype=np.float32), dtype=np.float32)[:, None]
    d = (d / ln).astype(np.float32)
    r = np.zeros((n, 8), np.float32)
    r[:, :3] = org
    r[:, 3:6] = d
    r[:, 6] = 1e-5
    r[:, 7] = np.inf
    return r


def grazing_rays(sc, n, seed):
    """Rays that run almost IN the plane of a triangle (1e-6 ... 1e-2 rad off it) through a point of that triangle: the regime
    in which Moeller-Trumbore is ill-conditioned (|det| = 2 area |cos incidence| -> 0) and reports hits from outside the
    triangle, at distances that can lie in front of the triangle's own box -- the one class of rays on which the ordered
    traversal is known to differ from the reference.  Targets: interior points, edge midpoints and vertices;
    near and far origins; float32 arithmetic, unit directions."""
    rng = np.random.default_rng(seed)
    tri = sc.pos.reshape(-1, 3, 3).astype(np.float64)
    if sc.ptype is not None:
        tri = tri[sc.ptype == 2]
    pick = rng.integers(0, len(tri), n)
    a, b, c = tri[pick, 0], tri[pick, 1], tri[pick, 2]
    kind = rng.integers(0, 3, n)                       # 0 interior point, 1 edge midpoint, 2 vertex
    u, v = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    flip = u + v > 1
    u, v = np.where(flip, 1 - u, u), np.where(flip, 1 - v, v)
    tgt = a + u[:, None] * (b - a) + v[:, None] * (c - a)
    tgt = np.where((kind == 1)[:, None], 0.5 * (a + b), tgt)
    tgt = np.where((kind == 2)[:, None], a, tgt)
    nrm = np.cross(b - a, c - a)
    nl = np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = nrm / np.where(nl > 0, nl, 1)
    g = rng.normal(size=(n, 3))
    inplane = g - (g * nrm).sum(1, keepdims=True) * nrm
    inplane /= np.maximum(np.linalg.norm(inplane, axis=1, keepdims=True), 1e-30)
    eps = 10.0 ** rng.uniform(-6, -2, n) * rng.choice([-1, 1], n)
    d = inplane + eps[:, None] * nrm
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = np.where(rng.integers(0, 2, n) == 1, rng.uniform(5, 40, n), rng.uniform(0.3, 3, n))[:, None]
    org = (tgt - d * dist).astype(np.float32)
    d32 = (tgt.astype(np.float32) - org).astype(np.float32)
    ln = np.sqrt((d32 * d32).sum(1, dtype=np.float32), dtype=np.float32)[:, None]
    d32 = (d32 / ln).astype(np.float32)
    r = np.zeros((n, 8), np.float32)
    r[:, :3] = org
    r[:, 3:6] = d32
    r[:, 6] = 1e-5
    r[:, 7] = np.inf
    return r
