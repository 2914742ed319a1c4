"""ctypes binding of libmirt.so (include/mirt.h).  Plumbing only: every compute call goes to the HIP library; if the
library is missing or no GPU is usable the call raises -- there is no Python or CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIRT_LIB") or os.path.join(_HERE, "libmirt.so")   # MIRT_LIB: tuning builds (tools/build_variant.sh)

SYMBOLS = [
    "mirt_init", "mirt_shutdown", "mirt_last_error", "mirt_version", "mirt_scene_create", "mirt_scene_create_prims",
    "mirt_scene_create_opts", "mirt_scene_destroy", "mirt_scene_build_times", "mirt_scene_own_tree",
    "mirt_scene_info", "mirt_scene_bvh2", "mirt_camera_look_at", "mirt_render", "mirt_render_accum", "mirt_tonemap",
    "mirt_primary_hits", "mirt_trace", "mirt_save_bmp", "mirt_tonemap_host", "mirt_bvh_build_host",
    "mirt_bvh_build_host_prims", "mirt_trace_static", "mirt_group_create", "mirt_group_destroy", "mirt_group_size",
    "mirt_group_set_scene", "mirt_group_render",
]


class MirtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmirt error {code}: {msg}")
        self.code = code
        self.msg = msg


class Bsdf(C.Structure):
    _fields_ = [("type", C.c_int32), ("p", C.c_float * 8)]


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("p", C.c_float * 16)]


class CameraRec(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("c2w", C.c_float * 9), ("hFov", C.c_float), ("vFov", C.c_float),
                ("nClip", C.c_float), ("fClip", C.c_float), ("aperture", C.c_float), ("focal_distance", C.c_float)]


class Params(C.Structure):
    _fields_ = [("samples_per_pixel", C.c_uint32), ("samples_per_light", C.c_uint32), ("samples_per_batch", C.c_uint32),
                ("max_tolerance", C.c_float), ("max_ray_depth", C.c_uint32), ("bvh_node_size", C.c_uint32)]


class Options(C.Structure):
    _fields_ = [("rng_mode", C.c_uint32), ("seed", C.c_uint32), ("traversal", C.c_uint32), ("tile_rank", C.c_uint32),
                ("tile_world", C.c_uint32), ("sample_begin", C.c_uint32), ("sample_count", C.c_uint32),
                ("glass_int_abs", C.c_uint32), ("flags", C.c_uint32), ("streams", C.c_uint32),
                ("shares", C.c_uint32), ("pool_group", C.c_uint32), ("reserved", C.c_uint32 * 4)]


class BuildOptions(C.Structure):
    _fields_ = [("own_tree_builder", C.c_uint32), ("reserved", C.c_uint32 * 7)]


class Stats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("paths", C.c_uint64),
                ("exact_fallback_rays", C.c_uint64), ("iterations", C.c_uint32), ("tonemap_table_exact", C.c_uint32),
                ("render_ms", C.c_double), ("trace_closest_ms", C.c_double), ("trace_shadow_ms", C.c_double),
                ("logic_ms", C.c_double), ("total_ms", C.c_double), ("closest_launches", C.c_uint32),
                ("shadow_launches", C.c_uint32), ("closest_node_visits", C.c_uint64), ("closest_tri_tests", C.c_uint64),
                ("shadow_node_visits", C.c_uint64), ("shadow_tri_tests", C.c_uint64),
                ("closest_node_steps", C.c_uint64), ("closest_tri_steps", C.c_uint64),
                ("shadow_node_steps", C.c_uint64), ("shadow_tri_steps", C.c_uint64),
                ("traversal_used", C.c_uint32), ("logic_launches", C.c_uint32),
                ("logic_live_slots", C.c_uint64), ("logic_scanned_slots", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


assert C.sizeof(Bsdf) == 36 and C.sizeof(Light) == 68 and C.sizeof(CameraRec) == 72 and C.sizeof(Params) == 24

_lib = None


def load():
    """Load libmirt.so (built in-tree by raytracer_amd/csrc/Makefile).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: run `make -C raytracer_amd/csrc` (hipcc, gfx950)")
        lib = C.CDLL(LIB_PATH)
        lib.mirt_last_error.restype = C.c_char_p
        lib.mirt_version.restype = C.c_char_p
        lib.mirt_scene_destroy.restype = None
        lib.mirt_group_destroy.restype = None
        lib.mirt_shutdown.restype = None
        _lib = lib
    return _lib


def _chk(rc):
    if rc != 0:
        raise MirtError(rc, load().mirt_last_error().decode())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def init(device=0):
    _chk(load().mirt_init(C.c_int(device)))


def bsdf_records(mats9):
    """(M, 9) float rows [type, union floats...] -> array of mirt_bsdf."""
    mats9 = np.asarray(mats9, np.float32).reshape(-1, 9)
    rec = np.zeros((len(mats9), 9), np.uint32)
    rec[:, 0] = mats9[:, 0].astype(np.uint32)
    rec[:, 1:] = mats9[:, 1:].view(np.uint32)
    return rec


def light_records(lights16):
    """(L, 16) factory-argument rows -> (L, 17) stored mirt_light records, built as the reference's factories do
    (scene/lights.cuh:123-146): area direction normalised + area = |dim_x| |dim_y|, directional negated + normalised.
    Same float32 operation order as linalg/Vector3f.cuh (norm = sqrt(x*x + y*y + z*z), unit = v * float(1/norm))."""
    lights16 = np.asarray(lights16, np.float32).reshape(-1, 16)
    f = np.float32

    def norm(v):
        return np.sqrt(f(f(f(v[0] * v[0]) + f(v[1] * v[1])) + f(v[2] * v[2])), dtype=np.float32)

    def unit(v):
        rn = f(np.float64(1.0) / np.float64(norm(v)))
        return np.array([f(v[0] * rn), f(v[1] * rn), f(v[2] * rn)], np.float32)

    out = np.zeros((len(lights16), 17), np.uint32)
    for i, l in enumerate(lights16):
        t = int(l[0])
        p = np.zeros(16, np.float32)
        a = l[1:]
        if t == 0:
            p[0:6] = a[0:6]
            p[6:9] = unit(a[6:9])
            p[9:15] = a[9:15]
            p[15] = f(norm(a[9:12]) * norm(a[12:15]))
        elif t == 1:
            p[0:6] = a[0:6]
        else:
            t = 2
            p[0:3] = a[0:3]
            p[3:6] = unit(-a[3:6])
        out[i, 0] = t
        out[i, 1:] = p.view(np.uint32)
    return out


class Scene:
    """Owns a mirt_scene handle.  Mirrors what PathTracer::set_scene + the BVH part of raytrace() do."""

    def __init__(self, sc, leaf_size=4, builder=None):
        """builder: None = library default (or MIRT_BUILDER); "host", "device" (= "lbvh": Morton radix tree on the GPU) or "ploc"
        (locally-ordered clustering on the GPU) = mirt_build_options.own_tree_builder 0 / 1 / 2."""
        lib = load()
        self.n = sc.n
        self.mats = bsdf_records(sc.mats)
        self.lights = light_records(sc.lights)
        h = C.c_void_p()
        ptype = getattr(sc, "ptype", None)
        if builder is not None:
            bo = BuildOptions()
            bo.own_tree_builder = {"host": 0, "device": 1, "lbvh": 1, "ploc": 2}[builder]
            _chk(lib.mirt_scene_create_opts(_p(sc.pos), _p(sc.nrm), _p(sc.mat), _p(ptype), C.c_uint32(sc.n), _p(self.mats),
                                            C.c_uint32(len(self.mats)), _p(self.lights) if len(self.lights) else None,
                                            C.c_uint32(len(self.lights)), C.c_uint32(leaf_size), C.byref(bo), C.byref(h)))
        elif ptype is None:
            _chk(lib.mirt_scene_create(_p(sc.pos), _p(sc.nrm), _p(sc.mat), C.c_uint32(sc.n), _p(self.mats),
                                       C.c_uint32(len(self.mats)), _p(self.lights) if len(self.lights) else None,
                                       C.c_uint32(len(self.lights)), C.c_uint32(leaf_size), C.byref(h)))
        else:                                   # spheres: the reference's enum per primitive (1 SPHERE, 2 TRIANGLE)
            _chk(lib.mirt_scene_create_prims(_p(sc.pos), _p(sc.nrm), _p(sc.mat), _p(ptype), C.c_uint32(sc.n), _p(self.mats),
                                             C.c_uint32(len(self.mats)), _p(self.lights) if len(self.lights) else None,
                                             C.c_uint32(len(self.lights)), C.c_uint32(leaf_size), C.byref(h)))
        self.h = h
        self.camera = None
        if sc.camera is not None:
            self.set_camera(sc.camera)

    def close(self):
        if getattr(self, "h", None):
            load().mirt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        a = (C.c_uint32 * 8)()
        _chk(load().mirt_scene_info(self.h, a))
        keys = ["bvh2_nodes", "bvh2_depth", "wide_nodes", "wide_depth", "n_tris", "n_mats", "n_lights", "device_kib"]
        return dict(zip(keys, list(a)))

    def build_times(self):
        a = (C.c_float * 8)()
        _chk(load().mirt_scene_build_times(self.h, a))
        keys = ["reference_bvh2_ms", "fold4_ms", "own_tree_ms", "dev_sort_ms", "dev_tree_ms", "dev_fold_ms", "dev_gather_ms",
                "upload_ms"]
        return dict(zip(keys, [float(x) for x in a]))

    def own_tree(self):
        """(counts dict, nodes8q (N, 24) uint32, prims (n, 12) uint32) as they sit on the device."""
        c = (C.c_uint32 * 4)()
        _chk(load().mirt_scene_own_tree(self.h, c, None, None))
        nodes = np.zeros((c[0], 24), np.uint32)
        prims = np.zeros((c[2], 12), np.uint32)
        if c[0]:
            _chk(load().mirt_scene_own_tree(self.h, c, _p(nodes), _p(prims)))
        return {"nodes": c[0], "depth": c[1], "prims": c[2], "builder": c[3]}, nodes, prims

    def bvh2(self):
        n_nodes = self.info()["bvh2_nodes"]
        nodes = np.zeros((n_nodes, 8), np.uint32)
        flat = np.zeros(self.n, np.uint32)
        _chk(load().mirt_scene_bvh2(self.h, _p(nodes), _p(flat)))
        return nodes, flat

    def set_camera(self, cam):
        rec = CameraRec()
        v = lambda t: (C.c_float * 3)(*t)
        _chk(load().mirt_camera_look_at(v(cam.look_from), v(cam.look_at), v(cam.up), C.c_float(cam.hfov),
                                        C.c_float(cam.vfov), C.c_float(cam.nclip), C.c_float(cam.fclip),
                                        C.c_float(cam.aperture), C.c_float(cam.focal), C.byref(rec)))
        self.camera = rec
        return rec

    def camera_record(self):
        return np.frombuffer(bytes(self.camera), np.float32).copy()

    @staticmethod
    def _params(spp, spl, batch, tol, depth, leaf):
        return Params(spp, spl, batch, tol, depth, leaf)

    @staticmethod
    def options(rng_mode=0, seed=0, traversal=0, tile_rank=0, tile_world=1, sample_begin=0, sample_count=0,
                glass_int_abs=0, profile=False, count=False, streams=0, accumulate=False, profile_closest=False,
                cull_zero=False, shares=0, pool_group=0):
        o = Options()
        o.rng_mode, o.seed, o.traversal = rng_mode, seed, traversal
        o.tile_rank, o.tile_world = tile_rank, tile_world
        o.sample_begin, o.sample_count, o.glass_int_abs = sample_begin, sample_count, glass_int_abs
        o.flags = (1 if profile else 0) | (2 if count else 0) | (4 if accumulate else 0) | (8 if profile_closest else 0) | \
            (16 if cull_zero else 0)
        o.streams = streams
        o.shares = shares
        o.pool_group = pool_group
        return o

    def render(self, w, h, spp=4, spl=1, batch=64, tol=0.05, depth=4, leaf=4, **opt):
        prm = self._params(spp, spl, batch, tol, depth, leaf)
        o = self.options(**opt)
        st = Stats()
        rgb = np.zeros((h, w, 3), np.int32)
        _chk(load().mirt_render(self.h, C.byref(self.camera), C.byref(prm), C.byref(o), C.c_uint32(w), C.c_uint32(h),
                                _p(rgb), C.byref(st)))
        self.last = st.as_dict()
        return rgb

    def render_accum(self, w, h, accum_ptr, spp=4, spl=1, batch=64, tol=0.05, depth=4, leaf=4, **opt):
        """accum_ptr: integer device address of a w*h*4 float32 buffer (e.g. torch tensor .data_ptr())."""
        prm = self._params(spp, spl, batch, tol, depth, leaf)
        o = self.options(**opt)
        st = Stats()
        _chk(load().mirt_render_accum(self.h, C.byref(self.camera), C.byref(prm), C.byref(o), C.c_uint32(w),
                                      C.c_uint32(h), C.c_void_p(accum_ptr), C.byref(st)))
        self.last = st.as_dict()
        return self.last

    def primary_hits(self, w, h, **opt):
        o = self.options(**opt)
        idx = np.zeros((h, w), np.int32)
        tb = np.zeros((h, w), np.uint32)
        _chk(load().mirt_primary_hits(self.h, C.byref(self.camera), C.byref(o), C.c_uint32(w), C.c_uint32(h), _p(idx), _p(tb)))
        return idx, tb

    def trace(self, rays8, anyhit=False, traversal=0, bary=False):
        rays8 = np.ascontiguousarray(rays8, np.float32).reshape(-1, 8)
        n = len(rays8)
        idx = np.zeros(n, np.int32)
        tb = np.zeros(n, np.uint32)
        b = np.zeros((n, 2), np.float32) if bary else None
        _chk(load().mirt_trace(self.h, _p(rays8), C.c_uint32(n), C.c_int(int(anyhit)), C.c_uint32(traversal), _p(idx),
                               _p(tb), _p(b)))
        return (idx, tb, b) if bary else (idx, tb)


class Group:
    """mirt_group_*: one frame over several GPUs (or logical partitions of one) from this process."""

    def __init__(self, devices):
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _chk(load().mirt_group_create(arr, C.c_int(len(devices)), C.byref(h)))
        self.h = h
        self.camera = CameraRec()

    def close(self):
        if getattr(self, "h", None):
            load().mirt_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, sc, leaf_size=4):
        mats, lights = bsdf_records(sc.mats), light_records(sc.lights)
        _chk(load().mirt_group_set_scene(self.h, _p(sc.pos), _p(sc.nrm), _p(sc.mat), _p(getattr(sc, "ptype", None)), C.c_uint32(sc.n),
                                         _p(mats), C.c_uint32(len(mats)), _p(lights) if len(lights) else None,
                                         C.c_uint32(len(lights)), C.c_uint32(leaf_size), None))
        rec = CameraRec()
        cam = sc.camera
        v = lambda t: (C.c_float * 3)(*t)
        _chk(load().mirt_camera_look_at(v(cam.look_from), v(cam.look_at), v(cam.up), C.c_float(cam.hfov), C.c_float(cam.vfov),
                                        C.c_float(cam.nclip), C.c_float(cam.fclip), C.c_float(cam.aperture), C.c_float(cam.focal),
                                        C.byref(rec)))
        self.camera = rec

    def render(self, w, h, partition="spp", spp=4, spl=1, batch=64, tol=0.05, depth=4, leaf=4, **opt):
        prm = Params(spp, spl, batch, tol, depth, leaf)
        o = Scene.options(**opt)
        st = Stats()
        rgb = np.zeros((h, w, 3), np.int32)
        _chk(load().mirt_group_render(self.h, C.byref(self.camera), C.byref(prm), C.byref(o), C.c_int(1 if partition == "tile" else 0),
                                      C.c_uint32(w), C.c_uint32(h), _p(rgb), C.byref(st)))
        self.last = st.as_dict()
        return rgb


TRACE_STATIC_MAX_RAYS = 65536


def trace_static(scene, rays_ptr, n, idx_ptr, tbits_ptr, anyhit=False, traversal=0):
    """mirt_trace_static: device pointers in and out (integers, e.g. torch tensor .data_ptr()), no allocation inside."""
    _chk(load().mirt_trace_static(scene.h, C.c_void_p(rays_ptr), C.c_uint32(n), C.c_int(int(anyhit)), C.c_uint32(traversal),
                                  C.c_void_p(idx_ptr), C.c_void_p(tbits_ptr)))


def bvh_build_host(pos, leaf_size=4, ptype=None):
    """Host-only structure probe (no GPU needed): the reference-layout BVH2, the flat order, the wide nodes, and the ordered
    traversal's own quantised tree."""
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 9)
    n = len(pos)
    pt = None if ptype is None else np.ascontiguousarray(ptype, np.uint8)
    cnt = (C.c_uint32 * 8)()
    f = load().mirt_bvh_build_host_prims
    _chk(f(_p(pos), _p(pt), C.c_uint32(n), C.c_uint32(leaf_size), cnt, None, None, None, None, None))
    nodes2 = np.zeros((cnt[0], 8), np.uint32)
    flat = np.zeros(n, np.uint32)
    nodes4 = np.zeros((cnt[2], 32), np.uint32)
    nodes8q = np.zeros((cnt[4], 24), np.uint32)
    fast_to_flat = np.zeros(n, np.uint32)
    _chk(f(_p(pos), _p(pt), C.c_uint32(n), C.c_uint32(leaf_size), cnt, _p(nodes2), _p(flat), _p(nodes4), _p(nodes8q),
           _p(fast_to_flat)))
    return dict(nodes2=nodes2, flat=flat, nodes4=nodes4, depth2=cnt[1], depth4=cnt[3], nodes8q=nodes8q, depth8=cnt[5],
                fast_to_flat=fast_to_flat)


def tonemap(accum_ptr, w, h, out=None):
    """Film of an accumulation buffer as (h, w, 3) int32 in host memory.  `out`: a buffer to fill instead of a new array (the
    reference's sample_buffer lives across frames too, renderer.cuh:61; a fresh 25 MB array costs its page faults per frame)."""
    if out is not None and (out.shape != (h, w, 3) or out.dtype != np.int32 or not out.flags["C_CONTIGUOUS"]):
        raise ValueError("out must be a C-contiguous (h, w, 3) int32 array")
    rgb = np.zeros((h, w, 3), np.int32) if out is None else out
    _chk(load().mirt_tonemap(C.c_void_p(accum_ptr), C.c_uint32(w), C.c_uint32(h), _p(rgb)))
    return rgb


def tonemap_host(rgb_float):
    a = np.ascontiguousarray(rgb_float, np.float32).reshape(-1, 3)
    out = np.zeros(a.shape, np.int32)
    _chk(load().mirt_tonemap_host(_p(a), C.c_uint32(len(a)), _p(out)))
    return out


def save_bmp(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.int32)
    h, w, _ = rgb.shape
    _chk(load().mirt_save_bmp(path.encode(), _p(rgb), C.c_uint32(w), C.c_uint32(h)))
