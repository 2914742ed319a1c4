#!/usr/bin/env python3
"""Dumps golden vectors from the REFERENCE ITSELF (oracle/_ref/libref*.so = the reference's own headers compiled with g++ in
the container where /root/reference is mounted) into tests/golden/*.npz.  Fixtures are data only (inputs + expected
outputs); they let the GPU box and fresh checkouts pin the C oracle and the HIP path without the reference present.

    make -C oracle ref && python3 oracle/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cpulibs import GOLDEN, CpuTracer  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

os.makedirs(GOLDEN, exist_ok=True)
R = CpuTracer("ref")
RF = CpuTracer("ref", "_fabs")
rng = np.random.default_rng(1234)


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def save(name, **kw):
    np.savez_compressed(os.path.join(GOLDEN, name), **kw)
    print(name, {k: getattr(v, "shape", v) for k, v in kw.items()})


# ---- 1. the survey's known-answer scene: sum of all channels must be 950192
sc = scenes.known_answer_scene()
R.load(sc)
img = R.render(64, 48, spp=4, spl=1, batch=64, tol=0.05, depth=4)
assert int(img.sum()) == 950192, img.sum()
bmp_path = os.path.join(GOLDEN, "_tmp.bmp")
R.lib.ref_save_bmp(bmp_path.encode(), img.ctypes.data_as(__import__("ctypes").c_void_p), 64, 48)
bmp = np.fromfile(bmp_path, np.uint8)
os.remove(bmp_path)
idx, tb = R.primary_hits(64, 48)
save("known_answer.npz", rgb=img, checksum=np.int64(img.sum()), bmp=bmp, hit_idx=idx, hit_tbits=tb)

# ---- 2. BVH structure: full dump for small scenes, hashes for larger ones
out = {}
for key, s in [("cornell2", scenes.cornell_scene(2)), ("soup500", scenes.random_soup(500, 1)),
               ("blob64", scenes.blob_scene(64, 32)), ("conf20k", scenes.conference_scene(20000)),
               ("bunny", scenes.blob_scene(264, 132))]:
    R.load(s)
    nodes, flat = R.bvh()
    n_nodes, depth = R.info()
    out[key + "_info"] = np.array([s.n, n_nodes, depth], np.int64)
    out[key + "_sha"] = np.array(sha(nodes, flat))
    if s.n <= 4000:
        out[key + "_nodes"] = nodes
        out[key + "_flat"] = flat
save("bvh.npz", **out)

# ---- 3. primary hits (sample 0, reference seed procedure) and full renders
out = {}
for key, s, (w, h), kw in [
    ("cornell_diffuse", scenes.cornell_scene(2, "diffuse"), (64, 48), dict(spp=4, depth=4)),
    ("cornell_mixed", scenes.cornell_scene(2, "mixed"), (64, 48), dict(spp=8, depth=6, spl=2)),
    ("blob64", scenes.blob_scene(64, 32), (96, 96), dict(spp=4, depth=4, batch=5, tol=0.0)),
    ("soup500", scenes.random_soup(500, 1), (64, 64), dict(spp=2, depth=3)),
    ("conf20k", scenes.conference_scene(20000), (80, 45), dict(spp=4, depth=8)),
    ("adaptive", scenes.cornell_scene(1, "diffuse"), (40, 30), dict(spp=32, depth=3, batch=4, tol=0.6)),
]:
    R.load(s)
    RF.load(s)
    idx, tb = R.primary_hits(w, h)
    out[key + "_hit_idx"] = idx
    out[key + "_hit_tbits"] = tb
    out[key + "_rgb"] = R.render(w, h, **kw)
    out[key + "_rgb_fabs"] = RF.render(w, h, **kw)
    out[key + "_params"] = np.array([w, h, kw.get("spp", 4), kw.get("spl", 1), kw.get("batch", 64), kw.get("depth", 4)], np.int64)
    out[key + "_tol"] = np.float32(kw.get("tol", 0.05))
save("renders.npz", **out)

# ---- 4. float radiance (mean of estimate_global_lighting) for rRMSE tests: two independent reference runs
s = scenes.cornell_scene(2, "diffuse")
RF.load(s)
a = RF.radiance(48, 36, spp=256, depth=4, seed_xor=0, nthreads=8)
b = RF.radiance(48, 36, spp=256, depth=4, seed_xor=0x5bd1e995, nthreads=8)
save("radiance.npz", cornell_a=a, cornell_b=b, params=np.array([48, 36, 256, 4], np.int64))

# ---- 5. unit vectors
u = {}
seeds = np.array([0, 1, 5, 42, 0x42ffffff, 0xdeadbeef], np.uint32)
seq = []
for s0 in seeds:
    st = int(s0)
    row = []
    for _ in range(8):
        f, st = R.random_float(st)
        row.append(f)
    seq.append(row)
u["rand_seeds"] = seeds
u["rand_seq"] = np.array(seq, np.float32)

# slab test incl. NaN cases
boxes, rays, res = [], [], []
for _ in range(300):
    lo = rng.uniform(-1, 0.5, 3)
    hi = lo + rng.uniform(0, 1.5, 3)
    o = rng.uniform(-2, 2, 3)
    d = rng.normal(size=3)
    if rng.random() < 0.3:
        d[rng.integers(0, 3)] = 0.0
    if rng.random() < 0.3:
        k = rng.integers(0, 3)
        o[k] = lo[k] if rng.random() < 0.5 else hi[k]
    mx = np.inf if rng.random() < 0.5 else rng.uniform(0.1, 3)
    boxes.append(np.concatenate([lo, hi]))
    rays.append(np.concatenate([o, d, [1e-5, mx]]))
boxes.append([1, -1, -1, 1, 1, 1]); rays.append([1, .5, 0, 0, 1, 0, 1e-5, np.inf])      # NaN on x flushed on y -> hit
boxes.append([-1, -1, 0, 1, 1, 0]); rays.append([0, -2, 0, 0, 1, 0, 1e-5, np.inf])      # NaN on z survives -> miss
boxes = np.array(boxes, np.float32)
rays = np.array(rays, np.float32)
res = np.array([R.bbox_intersect(b[:3], b[3:], r) for b, r in zip(boxes, rays)], np.int32)
u["bbox_boxes"], u["bbox_rays"], u["bbox_hit"] = boxes, rays, res

# Moeller-Trumbore incl. det == 0 and ties
tris, rays, hits, tbits, nrms = [], [], [], [], []
for i in range(300):
    p = rng.uniform(-1, 1, (3, 3))
    n = rng.normal(size=(3, 3))
    c = p.mean(0)
    o = rng.uniform(-2, 2, 3)
    tgt = c + rng.normal(scale=0.4, size=3)
    d = tgt - o
    if i % 7 == 0:
        d = d / np.linalg.norm(d)
    mx = np.inf if i % 3 else np.linalg.norm(tgt - o) * rng.uniform(0.5, 1.5)
    tris.append(np.concatenate([p.reshape(9), n.reshape(9)]))
    rays.append(np.concatenate([o, d, [1e-5, mx]]))
wall = np.array([1, 0, -1, 1, 2, -1, 1, 2, 1, 1, 0, 0, 1, 0, 0, 1, 0, 0], np.float64)
tris.append(wall); rays.append([0.5, 0.1, 0, 0, 1, 0, 1e-5, np.inf])    # parallel, origin off the plane: rejected
tris.append(wall); rays.append([1.0, 0.1, 0, 0, 1, 0, 1e-5, np.inf])    # parallel, origin IN the plane: accepted NaN hit
tris = np.array(tris, np.float32)
rays = np.array(rays, np.float32)
for any_hit in (0, 1):
    hh, tt, nn = [], [], []
    for t, r in zip(tris, rays):
        h, tb_, nr = R.tri_intersect(t, r, anyhit=bool(any_hit))
        hh.append(h); tt.append(tb_); nn.append(nr)
    u[f"tri_hit{any_hit}"] = np.array(hh, np.int32)
    u[f"tri_tbits{any_hit}"] = np.array(tt, np.uint32)
    u[f"tri_nrm{any_hit}"] = np.array(nn, np.float32)
u["tri_tris"], u["tri_rays"] = tris, rays

# shading frame
ns = rng.normal(size=(200, 3)).astype(np.float32)
ns[:6] = [[0, 1, 0], [1, 0, 0], [0, 0, 1], [0, -1, 0], [1, 1, 0], [.5, .5, .5]]
u["frame_n"] = ns
u["frame_o2w"] = np.array([R.make_coord_space(n) for n in ns], np.float32)
ir = rng.normal(size=(100, 8)).astype(np.float32)
ir[:, 6] = 1e-5
ir[:, 7] = np.inf
it = rng.uniform(0.1, 5, 100).astype(np.float32)
inn = rng.normal(size=(100, 3)).astype(np.float32)
u["isect_rays"], u["isect_t"], u["isect_n"] = ir, it, inn
u["isect_out"] = np.array([R.isect_compute(r, t, n) for r, t, n in zip(ir, it, inn)], np.float32)

# BSDFs: f and sample for every type; glass with both abs() variants
mats = np.array([scenes.WHITE, scenes.RED, scenes.emission(3, 2, 1), scenes.MIRROR, scenes.REFRACT, scenes.GLASS_PRESET,
                 scenes.glass(1.0, 0, (.7, .7, .7), (.7, .7, .7)), scenes.GOLD, scenes.ALUMINUM, scenes.BLUE_METAL], np.float32)
wo = rng.normal(size=(40, 3))
wo /= np.linalg.norm(wo, axis=1, keepdims=True)
wo[:, 2] = np.abs(wo[:, 2]) * np.where(np.arange(40) % 4 == 0, -1, 1)
wi = rng.normal(size=(40, 3))
wi /= np.linalg.norm(wi, axis=1, keepdims=True)
wo, wi = wo.astype(np.float32), wi.astype(np.float32)
u["bsdf_mats"], u["bsdf_wo"], u["bsdf_wi"] = mats, wo, wi
u["bsdf_f"] = np.array([[R.bsdf_f(m, a, b) for a, b in zip(wo, wi)] for m in mats], np.float32)
for tag, lib in (("", R), ("_fabs", RF)):
    outs, seeds_out = [], []
    for m in mats:
        ro, so = [], []
        for k, a in enumerate(wo):
            o7, s2, _ = lib.bsdf_sample(m, a, 1000 + 17 * k)
            ro.append(o7); so.append(s2)
        outs.append(ro); seeds_out.append(so)
    u["bsdf_sample" + tag] = np.array(outs, np.float32)
    u["bsdf_seed" + tag] = np.array(seeds_out, np.uint32)

# lights
lights = np.array([scenes.area_light((12, 11, 10), (0, 3, 0), (0.2, -1, 0.1), (1, 0, 0.3), (0, 0.1, 1)),
                   scenes.point_light((5, 5, 5), (1, 2, 3)), scenes.directional_light((.5, .4, .3), (-1, -1, 1))], np.float32)
pts = rng.uniform(-2, 2, (50, 3)).astype(np.float32)
u["light_args"], u["light_pts"] = lights, pts
u["light_records"] = np.array([R.light_record(l) for l in lights], np.uint32)
lo_, ls_ = [], []
for l in lights:
    ro, so = [], []
    for k, p in enumerate(pts):
        o8, s2 = R.light_sample(l, p, 77 + k)
        ro.append(o8); so.append(s2)
    lo_.append(ro); ls_.append(so)
u["light_out"], u["light_seed"] = np.array(lo_, np.float32), np.array(ls_, np.uint32)

# film
cs = np.concatenate([rng.uniform(0, 1.2, (2000, 3)), rng.uniform(0, 0.02, (500, 3)),
                     [[-1, 0, np.nan], [np.inf, -0.0, 1e-30], [0.5, 1.0, 2.0], [0.70710678, 0.7071068, 0.7071067]]]).astype(np.float32)
u["film_in"] = cs
u["film_out"] = np.array([R.update_pixel(c) for c in cs], np.int32)

# camera: record + rays (pinhole and thin lens)
cams = [scenes.Camera((0, 2, -6), (0, 1, 0), hfov=42, vfov=31.5),
        scenes.Camera((3.6, 2.4, -7.4), (-0.6, 0.8, 1.0), up=(0.1, 1, 0), hfov=60, vfov=33.75, nclip=0.1, fclip=100.0),
        scenes.Camera((1, 1, 1), (0, 0, 0), hfov=50, vfov=40, aperture=0.05, focal=0.0)]
R.load(scenes.cornell_scene(1))
recs, rr = [], []
xy = rng.uniform(0, 1, (30, 2)).astype(np.float32)
for c in cams:
    R.set_camera(c)
    recs.append(R.camera_record())
    rr.append([R.generate_ray(x, y, seed=99 + k)[0] for k, (x, y) in enumerate(xy)])
u["cam_args"] = np.array([[*c.look_from, *c.look_at, *c.up, c.hfov, c.vfov, c.nclip, c.fclip, c.aperture, c.focal] for c in cams], np.float32)
u["cam_records"] = np.array(recs, np.float32)
u["cam_xy"], u["cam_rays"] = xy, np.array(rr, np.float32)
save("units.npz", **u)

# ---- 6. struct sizes of the reference (measured) -- pins the C-ABI record layouts
sz = np.zeros(10, np.uint32)
R.lib.ref_sizes(sz.ctypes.data_as(__import__("ctypes").c_void_p))
save("sizes.npz", sizes=sz, names=np.array(["Vector3f", "Matrix3f", "Ray", "Intersection", "Primitive", "BSDF", "BVHNodeOpt",
                                            "SceneLight", "Camera", "Parameters"]))
print("done")
