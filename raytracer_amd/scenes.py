"""Scene descriptions as flat numpy arrays, in the layout the C-ABI (include/mirt.h) takes.

The four meshes BASELINE.json names (bunny / buddha / conference / sportsCar .obj) are absent from the
reference mount (``/root/reference/.MISSING_LARGE_BLOBS``), so the bench and the tests use the
deterministic procedural stand-ins SURVEY.md 8(d) defines.  If the real ``.obj`` files are dropped into
``objects/``, :func:`load_obj_scene` reads them instead (through the C++ host mirror's ``Generator::from_obj``, i.e. the
parser a caller of the kept host API gets: include/raytracer_amd/pathtracer.hpp, dumped by raytracer_amd/obj_info).

Array conventions (identical for the HIP path, the C oracle and the reference harness):
  pos    (N, 9) f32   p1 p2 p3 per triangle            (reference: primitives.cuh:23-26)
  nrm    (N, 9) f32   n1 n2 n3 per triangle
  mat    (N,)   u32   material index per triangle
  mats   (M, 9) f32   type, then the BSDF union's floats (bsdf.cuh:44-67)
  lights (L, 16) f32  type, then the light factory's arguments (lights.cuh:123-146)
"""
from __future__ import annotations

import dataclasses
import math
import os

import numpy as np

# BSDF / light type tags: the reference's enum order (bsdf.cuh:33-42, lights.cuh:11-16)
DIFFUSE, EMISSION, REFLECTION, REFRACTION, GLASS, MICROFACET = range(6)
AREA, POINT, DIRECTIONAL = range(3)

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def diffuse(r, g, b):
    return [DIFFUSE, r, g, b, 0, 0, 0, 0, 0]


def emission(r, g, b):
    return [EMISSION, r, g, b, 0, 0, 0, 0, 0]


def reflection(roughness, r, g, b):
    return [REFLECTION, roughness, r, g, b, 0, 0, 0, 0]


def refraction(ior, roughness, r, g, b):
    return [REFRACTION, ior, roughness, r, g, b, 0, 0, 0]


def glass(ior, roughness, refl, trans):
    return [GLASS, ior, roughness, *refl, *trans]


def microfacet(alpha, eta, k):
    return [MICROFACET, alpha, *eta, *k, 0]


def area_light(radiance, position, direction, dim_x, dim_y):
    return [AREA, *radiance, *position, *direction, *dim_x, *dim_y]


def point_light(radiance, position):
    return [POINT, *radiance, *position] + [0] * 9


def directional_light(radiance, direction):
    """direction = the direction the light TRAVELS (lights.cuh:140-145)."""
    return [DIRECTIONAL, *radiance, *direction] + [0] * 9


# the reference's preset BSDFs (bsdf.cuh:358-388)
WHITE = diffuse(.6, .6, .6)
RED = diffuse(.6, .2, .2)
GREEN = diffuse(.2, .6, .2)
BLUE = diffuse(.2, .2, .6)
BLACK = diffuse(0, 0, 0)
MIRROR = reflection(0., .9, .9, .9)
REFRACT = refraction(1.45, 0., 1, 1, 1)
GLASS_PRESET = glass(1.45, 0., (1, 1, 1), (1, 1, 1))
GOLD = microfacet(0.05, (0.18299, 0.42108, 1.37340), (3.42420, 2.34590, 1.77040))
ALUMINUM = microfacet(0.5, (1.34560, 0.96521, 0.61722), (7.47460, 6.39950, 5.30310))
BLUE_METAL = microfacet(0.5, (2.5, .2, .05), (0.5, 0.5, 0.5))


@dataclasses.dataclass
class Camera:
    look_from: tuple
    look_at: tuple
    up: tuple = (0., 1., 0.)
    hfov: float = 42.0
    vfov: float = 31.5
    nclip: float = 0.0
    fclip: float = float("inf")
    aperture: float = 0.0
    focal: float = 0.0

    def for_aspect(self, w, h):
        """vFov = hFov * h / w, the rule SURVEY 8(d) fixes for the stand-in scenes."""
        return dataclasses.replace(self, vfov=self.hfov * h / w)


@dataclasses.dataclass
class SceneData:
    name: str
    pos: np.ndarray
    nrm: np.ndarray
    mat: np.ndarray
    mats: np.ndarray
    lights: np.ndarray
    camera: Camera
    ptype: np.ndarray | None = None      # the reference's enum per primitive: 1 SPHERE (pos = centre, radius), 2 TRIANGLE

    @property
    def n(self):
        return int(self.pos.shape[0])

    def finalize(self):
        self.pos = np.ascontiguousarray(self.pos, np.float32).reshape(-1, 9)
        self.nrm = np.ascontiguousarray(self.nrm, np.float32).reshape(-1, 9)
        self.mat = np.ascontiguousarray(self.mat, np.uint32).reshape(-1)
        self.mats = np.ascontiguousarray(self.mats, np.float32).reshape(-1, 9)
        self.lights = np.ascontiguousarray(self.lights, np.float32).reshape(-1, 16)
        assert self.pos.shape[0] == self.nrm.shape[0] == self.mat.shape[0]
        if self.ptype is not None:
            self.ptype = np.ascontiguousarray(self.ptype, np.uint8)
        return self


# --------------------------------------------------------------------------------------- mesh helpers
def room_triangles(xs, ys, zs, walls=("floor", "ceiling", "left", "right", "back", "front")):
    """Generator::room (geometry/generator.cuh:10-96): two triangles per wall, same vertex order."""
    x, y, z = xs / 2, ys, zs / 2
    t = {
        "floor": ([(-x, 0, -z), (x, 0, -z), (x, 0, z)], [(-x, 0, -z), (-x, 0, z), (x, 0, z)], (0, 1, 0)),
        "ceiling": ([(-x, y, -z), (x, y, -z), (x, y, z)], [(-x, y, -z), (-x, y, z), (x, y, z)], (0, -1, 0)),
        "left": ([(-x, 0, -z), (-x, y, -z), (-x, y, z)], [(-x, 0, -z), (-x, 0, z), (-x, y, z)], (1, 0, 0)),
        "right": ([(x, 0, -z), (x, y, -z), (x, y, z)], [(x, 0, -z), (x, 0, z), (x, y, z)], (-1, 0, 0)),
        "back": ([(x, 0, z), (-x, y, z), (x, y, z)], [(x, 0, z), (-x, y, z), (-x, 0, z)], (0, 0, -1)),
        "front": ([(x, 0, -z), (-x, y, -z), (x, y, -z)], [(x, 0, -z), (-x, y, -z), (-x, 0, -z)], (0, 0, 1)),
    }
    pos, nrm, wall = [], [], []
    for k, name in enumerate(("floor", "ceiling", "left", "right", "back", "front")):
        if name not in walls:
            continue
        a, b, n = t[name]
        for tri in (a, b):
            pos.append(np.array(tri, np.float32).reshape(9))
            nrm.append(np.array([n, n, n], np.float32).reshape(9))
            wall.append(k)
    return np.array(pos, np.float32), np.array(nrm, np.float32), np.array(wall, np.uint32)


def uv_blob(nu, nv, center=(0, 0, 0), radius=1.0, amp=0.08, fu=7, fv=5, squash=(1, 1, 1)):
    """Displaced UV sphere R = radius*(1 + amp sin(fu phi) sin(fv theta)), 2*nu*(nv-1) triangles, smooth normals."""
    th = (np.arange(nv + 1, dtype=np.float64) / nv) * math.pi
    ph = (np.arange(nu, dtype=np.float64) / nu) * 2 * math.pi
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    R = radius * (1 + amp * np.sin(fu * PH) * np.sin(fv * TH))
    P = np.stack([R * np.sin(TH) * np.cos(PH), R * np.cos(TH), R * np.sin(TH) * np.sin(PH)], -1)
    P = P * np.array(squash, np.float64)
    # analytic-ish normals by finite differences of the parametrisation (good enough; they are data)
    d = 1e-4

    def pt(t, p):
        r = radius * (1 + amp * np.sin(fu * p) * np.sin(fv * t))
        return np.stack([r * np.sin(t) * np.cos(p), r * np.cos(t), r * np.sin(t) * np.sin(p)], -1) * np.array(squash)

    dT = pt(TH + d, PH) - pt(TH - d, PH)
    dP = pt(TH, PH + d) - pt(TH, PH - d)
    N = np.cross(dP, dT)
    ln = np.linalg.norm(N, axis=-1, keepdims=True)
    Nr = np.where(ln > 1e-12, N / np.maximum(ln, 1e-30), P / np.maximum(np.linalg.norm(P, axis=-1, keepdims=True), 1e-30))
    P = P + np.array(center, np.float64)
    i = np.arange(nv)[:, None]
    j = np.arange(nu)[None, :]
    jn = (j + 1) % nu
    a = (i, j + 0 * i)
    b = (i + 1, j + 0 * i)
    c = (i + 1, jn + 0 * i)
    e = (i, jn + 0 * i)

    def g(A, idx):
        return A[idx[0], idx[1]]

    t1 = np.stack([g(P, a), g(P, b), g(P, c)], -2)      # (nv, nu, 3, 3)
    t2 = np.stack([g(P, a), g(P, c), g(P, e)], -2)
    n1 = np.stack([g(Nr, a), g(Nr, b), g(Nr, c)], -2)
    n2 = np.stack([g(Nr, a), g(Nr, c), g(Nr, e)], -2)
    # drop the degenerate triangles at the poles: row 0 -> t2 degenerate (a==e), last row -> t1 degenerate (b==c)
    pos = np.concatenate([t1[:-1].reshape(-1, 9), t2[1:].reshape(-1, 9)], 0)
    nrm = np.concatenate([n1[:-1].reshape(-1, 9), n2[1:].reshape(-1, 9)], 0)
    # interleave deterministically by source row so that neighbouring triangles stay near each other in the input
    order = np.argsort(np.concatenate([np.arange(t1[:-1].reshape(-1, 9).shape[0]) * 2,
                                       np.arange(t2[1:].reshape(-1, 9).shape[0]) * 2 + 2 * nu + 1]), kind="stable")
    return pos[order].astype(np.float32), nrm[order].astype(np.float32)


def box_triangles(lo, hi):
    """Axis-aligned box, 12 triangles, outward flat normals."""
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    quads = [
        ([(x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (x1, y0, z0)], (0, 0, -1)),
        ([(x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)], (0, 0, 1)),
        ([(x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)], (-1, 0, 0)),
        ([(x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1)], (1, 0, 0)),
        ([(x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)], (0, -1, 0)),
        ([(x0, y1, z0), (x0, y1, z1), (x1, y1, z1), (x1, y1, z0)], (0, 1, 0)),
    ]
    pos, nrm = [], []
    for q, n in quads:
        for tri in ((q[0], q[1], q[2]), (q[0], q[2], q[3])):
            pos.append(np.array(tri, np.float32).reshape(9))
            nrm.append(np.array([n] * 3, np.float32).reshape(9))
    return np.array(pos, np.float32), np.array(nrm, np.float32)


def icosphere(subdiv, center=(0, 0, 0), radius=1.0):
    t = (1 + 5 ** 0.5) / 2
    v = np.array([(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
                  (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
                  (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11),
                  (6, 2, 10), (8, 6, 7), (9, 8, 1)])
    tri = v[f]
    for _ in range(subdiv):
        a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
        ab, bc, ca = (a + b) / 2, (b + c) / 2, (c + a) / 2
        for m in (ab, bc, ca):
            m /= np.linalg.norm(m, axis=1, keepdims=True)
        tri = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1),
                              np.stack([ab, bc, ca], 1)], 0)
    nrm = tri.copy()
    pos = tri * radius + np.array(center)
    return pos.reshape(-1, 9).astype(np.float32), nrm.reshape(-1, 9).astype(np.float32)


# --------------------------------------------------------------------------------------- MTL rule
def parse_mtl(path):
    """Minimal MTL reader -> list of dicts (name, Kd, Ks, Ke, Tf, Ni, Ns, illum)."""
    out, cur = [], None
    with open(path) as fh:
        for line in fh:
            tok = line.split()
            if not tok or tok[0].startswith("#"):
                continue
            if tok[0] == "newmtl":
                cur = dict(name=tok[1], Kd=(1., 1., 1.), Ks=(0., 0., 0.), Ke=(0., 0., 0.), Tf=(1., 1., 1.), Ni=1.0, Ns=10.0,
                           illum=0)
                out.append(cur)
            elif cur is not None and tok[0] in ("Kd", "Ks", "Ke", "Tf"):
                vals = [float(t) for t in tok[1:4]]
                while len(vals) < 3:
                    vals.append(vals[-1])
                cur[tok[0]] = tuple(vals)
            elif cur is not None and tok[0] in ("Ni", "Ns"):
                cur[tok[0]] = float(tok[1])
            elif cur is not None and tok[0] == "illum":
                cur["illum"] = int(float(tok[1]))
    return out


def mtl_to_bsdf(m):
    """The reference's MTL -> BSDF rule (geometry/generator.cuh:126-148)."""
    if m["Ke"][0] > 0:
        return emission(*m["Ke"])
    if m["illum"] == 3:
        return MIRROR
    if m["illum"] == 4:
        return glass(m["Ni"], 0, m["Tf"], m["Tf"])
    if m["illum"] == 11:
        return microfacet(np.float32(m["Ns"]) / np.float32(100), m["Kd"], m["Ks"])
    return diffuse(*m["Kd"])


def mtl_materials(name):
    return [mtl_to_bsdf(m) for m in parse_mtl(os.path.join(_ROOT, "objects", name))]


# --------------------------------------------------------------------------------------- scenes
def known_answer_scene():
    """SURVEY.md 8(c) known-answer scene (sphere + floor, 64x48, checksum 950192)."""
    pos = np.zeros((3, 9), np.float32)
    nrm = np.zeros((3, 9), np.float32)
    pos[0, :4] = [0, 1, 0, 1]
    pos[1] = [-5, 0, -5, 5, 0, -5, 5, 0, 5]
    pos[2] = [-5, 0, -5, -5, 0, 5, 5, 0, 5]
    nrm[1] = [0, 1, 0] * 3
    nrm[2] = [0, 1, 0] * 3
    return SceneData("known_answer", pos, nrm, np.array([0, 1, 1]), np.array([RED, WHITE]),
                     np.array([directional_light((.2, .2, .2), (0, -1, 0)), point_light((50, 50, 50), (0, 3, 0))]),
                     Camera((0, 2, -6), (0, 1, 0), hfov=42, vfov=31.5),
                     ptype=np.array([1, 2, 2], np.uint8)).finalize()


def blob_scene(nu=264, nv=132, name="bunny*"):
    """bunny* / buddha* stand-in: displaced UV sphere, centre (0,1.1,0), white, + red 2-triangle floor,
    one directional light radiance .5 travelling (-1,-1,1); camera (0,2,-4) -> (0,1,0), hFov 42.
    nu=264, nv=132 -> 69 170 triangles; nu=1024, nv=512 -> 1 046 530."""
    bp, bn = uv_blob(nu, nv, center=(0, 1.1, 0))
    fp = np.array([[-5, 0, -5, 5, 0, -5, 5, 0, 5], [-5, 0, -5, -5, 0, 5, 5, 0, 5]], np.float32)
    fn = np.array([[0, 1, 0] * 3] * 2, np.float32)
    pos = np.concatenate([bp, fp])
    nrm = np.concatenate([bn, fn])
    mat = np.concatenate([np.zeros(len(bp), np.uint32), np.ones(2, np.uint32)])
    return SceneData(name, pos, nrm, mat, np.array([WHITE, RED]),
                     np.array([directional_light((.5, .5, .5), (-1, -1, 1))]),
                     Camera((0, 2, -4), (0, 1, 0), hfov=42, vfov=42)).finalize()


def cornell_scene(subdiv=3, mats="mixed"):
    """Small closed room with an area light, an icosphere and a box: exercises every BSDF type the GPU path has."""
    rp, rn, wall = room_triangles(4, 3, 4)
    wall_mats = {0: 0, 1: 0, 2: 1, 3: 2, 4: 0, 5: 0}
    rm = np.array([wall_mats[int(k)] for k in wall], np.uint32)
    sp, sn = icosphere(subdiv, center=(-0.7, 0.7, 0.3), radius=0.7)
    bp, bn = box_triangles((0.4, 0.0, -0.9), (1.3, 1.2, 0.0))
    lp = np.array([[-.5, 2.999, -.5, .5, 2.999, -.5, .5, 2.999, .5], [-.5, 2.999, -.5, -.5, 2.999, .5, .5, 2.999, .5]],
                  np.float32)
    ln = np.array([[0, -1, 0] * 3] * 2, np.float32)
    if mats == "diffuse":
        table = [WHITE, RED, GREEN, BLUE, WHITE, emission(4, 4, 4)]
    else:
        table = [WHITE, RED, GREEN, GLASS_PRESET, GOLD, emission(4, 4, 4), MIRROR, REFRACT]
    pos = np.concatenate([rp, sp, bp, lp])
    nrm = np.concatenate([rn, sn, bn, ln])
    mat = np.concatenate([rm, np.full(len(sp), 3, np.uint32), np.full(len(bp), 4, np.uint32), np.full(2, 5, np.uint32)])
    if mats != "diffuse":
        mat[len(rp) + len(sp): len(rp) + len(sp) + 4] = 6          # two mirror faces on the box
        mat[len(rp): len(rp) + len(sp) // 4] = 7                   # a refractive cap on the sphere
    lights = np.array([area_light((12, 12, 12), (0, 2.99, 0), (0, -1, 0), (1, 0, 0), (0, 0, 1)),
                       point_light((1.5, 1.5, 1.5), (1.5, 2.5, -1.5))])
    return SceneData("cornell", pos, nrm, mat, np.array(table), lights,
                     Camera((0, 1.5, -1.95), (0, 1.2, 0), hfov=70, vfov=55)).finalize()


def conference_scene(target_tris=331_000, name="conference*"):
    """conference* stand-in: closed room (Generator::room, all six walls) + a long table + a grid of
    furniture blobs, the 33 diffuse materials of objects/conference.mtl, one directional light radiance .5 travelling
    (-1,-1,1) and one area light under the ceiling.  ~331 k triangles by default."""
    mats = mtl_materials("conference.mtl")
    nm = len(mats)
    parts_p, parts_n, parts_m = [], [], []
    rp, rn, wall = room_triangles(10, 4, 16)
    parts_p.append(rp); parts_n.append(rn); parts_m.append((wall % nm).astype(np.uint32) + 1)
    tp, tn = box_triangles((-1.2, 0.72, -5.0), (1.2, 0.80, 5.0))
    parts_p.append(tp); parts_n.append(tn); parts_m.append(np.full(len(tp), 8 % nm, np.uint32))
    for lx in (-1.0, 1.0):
        for lz in (-4.5, 0.0, 4.5):
            p, n = box_triangles((lx - .06, 0, lz - .06), (lx + .06, .72, lz + .06))
            parts_p.append(p); parts_n.append(n); parts_m.append(np.full(len(p), 9 % nm, np.uint32))
    # chairs: 2 rows x 12 along the table + 8 along the walls; each a squashed blob seat + a blob back
    seats = [(sx * 1.9, 0.45, -4.6 + 0.8 * k) for sx in (-1, 1) for k in range(12)]
    seats += [(sx * 4.3, 0.45, -6 + 1.7 * k) for sx in (-1, 1) for k in range(8)]
    n_blobs = 2 * len(seats) + 6
    per = max(64, (target_tris - 200) // n_blobs)
    nv = max(4, int(round(math.sqrt(per / 4.0))))
    nu = 2 * nv
    k = 0
    for (cx, cy, cz) in seats:
        p, n = uv_blob(nu, nv, center=(cx, cy, cz), radius=0.33, amp=0.05, fu=5, fv=3, squash=(1, 0.35, 1))
        parts_p.append(p); parts_n.append(n); parts_m.append(np.full(len(p), (10 + k) % nm, np.uint32)); k += 1
        bx = cx + (0.3 if cx > 0 else -0.3)
        p, n = uv_blob(nu, nv, center=(bx, cy + 0.45, cz), radius=0.36, amp=0.04, fu=3, fv=4, squash=(0.25, 1, 0.9))
        parts_p.append(p); parts_n.append(n); parts_m.append(np.full(len(p), (10 + k) % nm, np.uint32)); k += 1
    for i in range(6):       # objects on the table
        p, n = uv_blob(nu, nv, center=(-0.6 + 0.25 * i, 0.98, -3.5 + 1.4 * i), radius=0.18, amp=0.12, fu=6, fv=4)
        parts_p.append(p); parts_n.append(n); parts_m.append(np.full(len(p), (3 + 5 * i) % nm, np.uint32))
    pos = np.concatenate(parts_p)
    nrm = np.concatenate(parts_n)
    mat = np.concatenate(parts_m) % nm
    lights = np.array([directional_light((.5, .5, .5), (-1, -1, 1)),
                       area_light((18, 18, 18), (0, 3.98, 0), (0, -1, 0), (3.0, 0, 0), (0, 0, 8.0))])
    return SceneData(name, pos, nrm, mat, np.array(mats), lights,
                     Camera((3.6, 2.4, -7.4), (-0.6, 0.8, 1.0), hfov=60, vfov=60 * 9 / 16)).finalize()


def sportscar_scene(target_tris=300_000, name="sportsCar*"):
    """sportsCar* stand-in: a blob set on a floor using the 38 materials of objects/sportsCar.mtl
    (emissive, glass ior 1.0, mirror, microfacet, diffuse), directional + area light."""
    mats = mtl_materials("sportsCar.mtl")
    nm = len(mats)
    parts_p, parts_n, parts_m = [], [], []
    fp = np.array([[-12, 0, -12, 12, 0, -12, 12, 0, 12], [-12, 0, -12, -12, 0, 12, 12, 0, 12]], np.float32)
    parts_p.append(fp); parts_n.append(np.array([[0, 1, 0] * 3] * 2, np.float32)); parts_m.append(np.full(2, 3 % nm, np.uint32))
    n_blobs = nm
    per = max(64, target_tris // n_blobs)
    nv = max(4, int(round(math.sqrt(per / 4.0))))
    nu = 2 * nv
    for k in range(n_blobs):
        ang = 2 * math.pi * k / n_blobs
        ring = 1.2 + 0.9 * (k % 3)
        c = (ring * math.cos(ang), 0.55 + 0.25 * (k % 4), ring * math.sin(ang))
        p, n = uv_blob(nu, nv, center=c, radius=0.42, amp=0.07, fu=3 + k % 5, fv=2 + k % 3, squash=(1.3, 0.7, 0.9))
        parts_p.append(p); parts_n.append(n); parts_m.append(np.full(len(p), k % nm, np.uint32))
    lights = np.array([directional_light((.5, .5, .5), (-1, -1, 1)),
                       area_light((10, 10, 10), (0, 6, 0), (0, -1, 0), (4, 0, 0), (0, 0, 4))])
    return SceneData(name, np.concatenate(parts_p), np.concatenate(parts_n), np.concatenate(parts_m), np.array(mats), lights,
                     Camera((0, 3.2, -6.5), (0, 0.6, 0), hfov=50, vfov=50 * 9 / 16)).finalize()


def random_soup(n, seed=0, extent=1.0, tri_size=0.15, n_mats=3):
    """Random triangle soup for traversal parity tests (includes exact duplicates and axis-aligned triangles)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (n, 1, 3))
    pos = (c + rng.uniform(-tri_size, tri_size, (n, 3, 3))).astype(np.float32)
    if n >= 16:
        pos[1] = pos[0]                      # exact duplicate: tie rule
        pos[5, :, 0] = pos[5, 0, 0]          # triangle in a plane x = const
        pos[7, :, 1] = 0.0                   # triangle in the plane y = 0
    nrm = rng.normal(size=(n, 3, 3)).astype(np.float32)
    mats = np.array([WHITE, RED, GREEN][:n_mats])
    return SceneData(f"soup{n}", pos.reshape(n, 9), nrm.reshape(n, 9), rng.integers(0, n_mats, n), mats,
                     np.array([point_light((3, 3, 3), (0, 3, 0))]),
                     Camera((0, 0.5, -3.5), (0, 0, 0), hfov=42, vfov=42)).finalize()


def sphere_soup(n_tri=300, n_sph=60, seed=3):
    """Triangle soup + spheres (the reference's second primitive type): overlapping spheres, an exact duplicate (tie rule),
    spheres around the camera-side origins of the parity rays."""
    rng = np.random.default_rng(seed)
    t = random_soup(n_tri, seed)
    sp = np.zeros((n_sph, 9), np.float32)
    sp[:, :3] = rng.uniform(-1.0, 1.0, (n_sph, 3))
    sp[:, 3] = rng.uniform(0.03, 0.35, n_sph)
    sp[1] = sp[0]
    sp[2, :3] = sp[3, :3]                      # concentric
    pos = np.concatenate([t.pos, sp])
    nrm = np.concatenate([t.nrm, np.zeros((n_sph, 9), np.float32)])
    mat = np.concatenate([t.mat, rng.integers(0, len(t.mats), n_sph).astype(np.uint32)])
    ptype = np.concatenate([np.full(n_tri, 2, np.uint8), np.full(n_sph, 1, np.uint8)])
    perm = rng.permutation(n_tri + n_sph)
    return SceneData(f"sphere_soup{n_tri}_{n_sph}", pos[perm], nrm[perm], mat[perm], t.mats, t.lights, t.camera,
                     ptype=ptype[perm]).finalize()


def load_obj_scene(path, name=None, default_bsdf=None, scale=(1, 1, 1), translate=(0, 0, 0)):
    """A real OBJ/MTL pair as a SceneData, read by the C++ host mirror's Generator::from_obj (reference:
    geometry/generator.cuh:98-180; rapidobj parity unpinned).  The reference records no camera or lights for
    its meshes (from_obj has no caller there): the camera is fitted to the bounding box (three quarters view from
    outside), lights = the directional light of SURVEY 8(d) + an area light above the mesh."""
    import subprocess
    import tempfile

    tool = os.path.join(_ROOT, "raytracer_amd", "obj_info")
    if not os.path.exists(tool):
        raise FileNotFoundError(f"{tool} is missing: run `make -C raytracer_amd/csrc all`")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "mesh.bin")
        args = [tool, "--bin", out, path] + [repr(float(v)) for v in (*scale, *translate)]
        r = subprocess.run(args, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"obj_info failed on {path}: {r.stdout}{r.stderr}")
        raw = np.fromfile(out, np.uint32)
    n = int(raw[0])
    tri = raw[1:1 + 18 * n].view(np.float32).reshape(n, 18)
    rec = raw[1 + 18 * n:1 + 27 * n].reshape(n, 9)
    uniq, mat = np.unique(rec, axis=0, return_inverse=True)         # one material per distinct BSDF record
    mats = np.concatenate([uniq[:, :1].astype(np.float32), uniq[:, 1:].view(np.float32)], axis=1)
    if default_bsdf is not None:
        mats, mat = np.array([default_bsdf], np.float32), np.zeros(n, np.uint32)
    pts = tri[:, :9].reshape(-1, 3)
    lo, hi = pts.min(0).astype(np.float64), pts.max(0).astype(np.float64)
    c, ext = (lo + hi) / 2, float(np.linalg.norm(hi - lo))
    cam = Camera(tuple(c + ext * np.array([0.45, 0.35, -0.95])), tuple(c), hfov=42, vfov=42 * 9 / 16)
    top = hi[1] + 0.05 * ext
    lights = np.array([directional_light((.5, .5, .5), (-1, -1, 1)),
                       area_light((12, 12, 12), (c[0], top, c[2]), (0, -1, 0), (0.3 * ext, 0, 0), (0, 0, 0.3 * ext))])
    return SceneData(name or os.path.basename(path), tri[:, :9].copy(), tri[:, 9:].copy(), mat.astype(np.uint32), mats, lights,
                     cam).finalize()


def workload(name):
    """(SceneData, "real" | "synthetic"): `objects/<name>.obj` when somebody has dropped the real mesh there (the reference
    mount lacks the four meshes: .MISSING_LARGE_BLOBS), else the procedural stand-in of SURVEY 8(d)."""
    base = name.rstrip("*")
    path = os.path.join(_ROOT, "objects", base + ".obj")
    if os.path.exists(path):
        return load_obj_scene(path, base, default_bsdf=WHITE if base in ("bunny", "buddha") else None), "real"
    return WORKLOADS[name if name in WORKLOADS else base + "*"](), "synthetic"


WORKLOADS = {
    "bunny*": lambda: blob_scene(264, 132, "bunny*"),
    "buddha*": lambda: blob_scene(1024, 512, "buddha*"),
    "conference*": conference_scene,
    "sportsCar*": sportscar_scene,
}
