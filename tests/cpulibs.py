"""ctypes wrappers around the two CPU checkers (test infrastructure only):

* ``oracle/liboracle.so``     -- the plain-C restatement (prefix ``orc_``), always available once built;
* ``oracle/_ref/libref*.so``  -- the reference's own headers behind oracle/ref_harness.cpp (prefix ``ref_``), built only
  in the container where /root/reference is mounted (the prebuilt files travel to the GPU box).

Both export the same entry points, so one class drives either.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")

_c_f = C.c_float
_vp = C.c_void_p


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def build_oracle():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def ref_available(variant=""):
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", f"libref{variant}.so"))


class CpuTracer:
    def __init__(self, kind="oracle", variant=""):
        if kind == "oracle":
            self.lib = C.CDLL(build_oracle())
            self.px = "orc_"
        else:
            self.lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", f"libref{variant}.so"))
            self.px = "ref_"
        self.kind = kind
        L = self.lib
        self._f("scene_create").restype = _vp
        self._f("render").restype = C.c_double
        self._f("random_float").restype = _c_f
        if kind == "oracle":
            L.orc_ctr_key.restype = C.c_uint32
            L.orc_ctr_u32.restype = C.c_uint32
        self.h = None

    def _f(self, name):
        return getattr(self.lib, self.px + name)

    # ------------------------------------------------------------------ scene
    def load(self, sc, leaf_size=4):
        self.close()
        self.n = sc.n
        self.h = _vp(self._f("scene_create")(_p(sc.pos), _p(sc.nrm), _p(sc.mat), _p(sc.ptype), C.c_uint32(sc.n),
                                             _p(sc.mats), C.c_uint32(len(sc.mats)), _p(sc.lights),
                                             C.c_uint32(len(sc.lights)), C.c_uint32(leaf_size)))
        self.set_camera(sc.camera)
        return self

    def close(self):
        if self.h:
            self._f("scene_destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_camera(self, cam):
        v = lambda t: (_c_f * 3)(*t)
        self._f("set_camera")(self.h, v(cam.look_from), v(cam.look_at), v(cam.up), _c_f(cam.hfov), _c_f(cam.vfov),
                              _c_f(cam.nclip), _c_f(cam.fclip), _c_f(cam.aperture), _c_f(cam.focal))

    def camera_record(self):
        out = np.zeros(18, np.float32)
        self._f("get_camera")(self.h, _p(out))
        return out

    def set_rng(self, mode, seed=0):
        assert self.kind == "oracle"
        self.lib.orc_set_rng(self.h, C.c_int(mode), C.c_uint32(seed))

    def set_glass_float_abs(self, v):
        assert self.kind == "oracle"
        self.lib.orc_set_glass_float_abs(C.c_int(int(v)))

    def info(self):
        a, b = C.c_uint32(), C.c_uint32()
        self._f("scene_info")(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def bvh(self):
        n_nodes, _ = self.info()
        nodes = np.zeros((n_nodes, 8), np.uint32)
        flat = np.zeros(self.n, np.uint32)
        self._f("scene_bvh")(self.h, _p(nodes), _p(flat))
        return nodes, flat

    # ------------------------------------------------------------------ path
    def render(self, w, h, spp=4, spl=1, batch=64, tol=0.05, depth=4, leaf=4, nthreads=1, accum=False, sample_begin=0):
        prm = np.array([spp, spl, batch, depth, leaf], np.uint32)
        rgb = np.zeros((h, w, 3), np.int32)
        if self.kind == "oracle":
            acc = np.zeros((h, w, 4), np.float32) if accum else None
            cnt = np.zeros(4, np.uint64)
            t = self.lib.orc_render(self.h, _p(prm), _c_f(tol), C.c_uint32(w), C.c_uint32(h), _p(rgb), _p(acc),
                                    C.c_uint32(sample_begin), C.c_int(nthreads), _p(cnt))
            self.last = dict(seconds=t, closest=int(cnt[0]), shadow=int(cnt[1]), box=int(cnt[2]), tri=int(cnt[3]))
            return (rgb, acc) if accum else rgb
        t = self.lib.ref_render(self.h, _p(prm), _c_f(tol), C.c_uint32(w), C.c_uint32(h), _p(rgb), C.c_int(nthreads))
        self.last = dict(seconds=t)
        return rgb

    def primary_hits(self, w, h, rays=False):
        idx = np.zeros((h, w), np.int32)
        tb = np.zeros((h, w), np.uint32)
        r = np.zeros((h, w, 11), np.float32) if rays else None
        self._f("primary_hits")(self.h, C.c_uint32(w), C.c_uint32(h), _p(idx), _p(tb), _p(r))
        return (idx, tb, r) if rays else (idx, tb)

    def trace(self, rays8, anyhit=False, normals=False):
        rays8 = np.ascontiguousarray(rays8, np.float32).reshape(-1, 8)
        n = len(rays8)
        idx = np.zeros(n, np.int32)
        tb = np.zeros(n, np.uint32)
        nr = np.zeros((n, 3), np.float32) if normals else None
        self._f("trace")(self.h, _p(rays8), C.c_uint32(n), C.c_int(int(anyhit)), _p(idx), _p(tb), _p(nr))
        return (idx, tb, nr) if normals else (idx, tb)

    def radiance(self, w, h, spp, spl=1, depth=4, leaf=4, seed_xor=0, nthreads=1):
        prm = np.array([spp, spl, spp + 1, depth, leaf], np.uint32)
        out = np.zeros((h, w, 3), np.float32)
        self._f("radiance")(self.h, _p(prm), C.c_uint32(w), C.c_uint32(h), C.c_uint32(seed_xor), _p(out), C.c_int(nthreads))
        return out

    # ------------------------------------------------------------------ unit probes
    def random_float(self, seed):
        s = C.c_uint32(seed)
        f = self._f("random_float")(C.byref(s))
        return np.float32(f), s.value

    def bbox_intersect(self, minp, maxp, ray8):
        a, b, r = (np.asarray(x, np.float32) for x in (minp, maxp, ray8))
        return int(self._f("bbox_intersect")(_p(a), _p(b), _p(r)))

    def tri_intersect(self, tri18, ray8, anyhit=False):
        t, r = np.asarray(tri18, np.float32), np.asarray(ray8, np.float32)
        tb = C.c_uint32()
        nr = np.zeros(3, np.float32)
        hit = int(self._f("tri_intersect")(_p(t), _p(r), C.c_int(int(anyhit)), C.byref(tb), _p(nr)))
        return hit, tb.value, nr

    def make_coord_space(self, n):
        n = np.asarray(n, np.float32)
        out = np.zeros(9, np.float32)
        self._f("make_coord_space")(_p(n), _p(out))
        return out

    def isect_compute(self, ray8, max_t, normal):
        r, n = np.asarray(ray8, np.float32), np.asarray(normal, np.float32)
        out = np.zeros(27, np.float32)
        self._f("isect_compute")(_p(r), _c_f(max_t), _p(n), _p(out))
        return out

    def bsdf_f(self, mat9, wo, wi):
        m, a, b = (np.asarray(x, np.float32) for x in (mat9, wo, wi))
        out = np.zeros(3, np.float32)
        self._f("bsdf_f")(_p(m), _p(a), _p(b), _p(out))
        return out

    def bsdf_sample(self, mat9, wo, seed):
        m, a = np.asarray(mat9, np.float32), np.asarray(wo, np.float32)
        s = C.c_uint32(seed)
        out = np.zeros(7, np.float32)
        d = int(self._f("bsdf_sample")(_p(m), _p(a), C.byref(s), _p(out)))
        return out, s.value, d

    def light_sample(self, light16, p, seed):
        l, q = np.asarray(light16, np.float32), np.asarray(p, np.float32)
        s = C.c_uint32(seed)
        out = np.zeros(8, np.float32)
        self._f("light_sample")(_p(l), _p(q), C.byref(s), _p(out))
        return out, s.value

    def light_record(self, light16):
        l = np.asarray(light16, np.float32)
        out = np.zeros(17, np.uint32)
        self._f("light_record")(_p(l), _p(out))
        return out

    def update_pixel(self, c):
        c = np.asarray(c, np.float32)
        out = np.zeros(3, np.int32)
        self._f("update_pixel")(_p(c), _p(out))
        return out

    def generate_ray(self, x, y, seed=1):
        s = C.c_uint32(seed)
        out = np.zeros(11, np.float32)
        self._f("generate_ray")(self.h, _c_f(x), _c_f(y), C.byref(s), _p(out))
        return out, s.value
