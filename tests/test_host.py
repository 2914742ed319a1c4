"""Host side of libmirt without a GPU: the library loads, exports every symbol include/mirt.h declares, and its host-only
pieces (BVH build, light/camera records, tonemap table, BMP writer, error paths) match the golden vectors."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

from raytracer_amd import capi, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    lib = capi.load()
    text = open(os.path.join(ROOT, "include", "mirt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mirt_[a-z0-9_]+)\s*\(", text)))
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(lib, name), f"libmirt.so does not export {name}"
    assert sorted(capi.SYMBOLS) == declared


def test_record_layouts_match_reference_structs(golden):
    g = golden("sizes")
    sizes = dict(zip([str(n) for n in g["names"]], [int(s) for s in g["sizes"]]))
    assert C.sizeof(capi.Bsdf) == sizes["BSDF"] == 36
    assert C.sizeof(capi.Light) == sizes["SceneLight"] == 68
    assert C.sizeof(capi.CameraRec) == sizes["Camera"] == 72
    assert C.sizeof(capi.Params) == sizes["Parameters"] == 24
    assert sizes["BVHNodeOpt"] == 32 and sizes["Primitive"] == 88 and sizes["Ray"] == 44


@pytest.mark.parametrize("key,make", [("cornell2", lambda: scenes.cornell_scene(2)), ("soup500", lambda: scenes.random_soup(500, 1)),
                                      ("blob64", lambda: scenes.blob_scene(64, 32)),
                                      ("conf20k", lambda: scenes.conference_scene(20000)),
                                      ("bunny", lambda: scenes.blob_scene(264, 132))])
def test_reference_bvh_rebuilt_bit_for_bit(golden, key, make):
    g = golden("bvh")
    sc = make()
    b = capi.bvh_build_host(sc.pos, 4)
    assert [sc.n, len(b["nodes2"]), b["depth2"]] == list(g[key + "_info"])
    h = hashlib.sha256()
    h.update(b["nodes2"].tobytes())
    h.update(b["flat"].tobytes())
    assert h.hexdigest() == str(g[key + "_sha"])


def _walk_wide(nodes4, node, out, boxes, parent_box):
    f = nodes4[node].view(np.float32)
    ch = nodes4[node][24:28]
    for k in range(4):
        if ch[k] == 0xFFFFFFFF:
            continue
        lo = np.array([f[k], f[4 + k], f[8 + k]])
        hi = np.array([f[12 + k], f[16 + k], f[20 + k]])
        if parent_box is not None:
            assert (lo >= parent_box[0]).all() and (hi <= parent_box[1]).all()
        if ch[k] & 0x80000000:
            out.append((int(ch[k] & ((1 << 27) - 1)), int((ch[k] >> 27) & 15)))
            boxes.append((lo, hi))
        else:
            _walk_wide(nodes4, int(ch[k]), out, boxes, (lo, hi))


@pytest.mark.parametrize("make", [lambda: scenes.cornell_scene(2), lambda: scenes.conference_scene(20000),
                                  lambda: scenes.random_soup(37, 2)])
def test_wide_tree_keeps_reference_leaf_order(make):
    sc = make()
    b = capi.bvh_build_host(sc.pos, 4)
    leaves, boxes = [], []
    _walk_wide(b["nodes4"], 0, leaves, boxes, None)
    n2 = b["nodes2"]
    ref_leaves = [(int(n[7]), int(n[6] & 0x7fffffff)) for n in n2 if n[6] & 0x80000000]     # node order = visiting order
    ref_leaves.sort()
    # inline leaves carry (start, count); oversized reference leaves carry count 0 and an index into the side table
    got = [(s, c) for s, c in leaves if c]
    big = [rl for rl in ref_leaves if rl[1] > 15]
    assert len([1 for s, c in leaves if c == 0]) == len(big)
    assert got == [rl for rl in ref_leaves if rl[1] <= 15] or sorted(got) == [rl for rl in ref_leaves if rl[1] <= 15]
    # DFS order of the wide tree visits primitives in increasing flat index (the reference's order)
    starts = [s for s, c in leaves if c]
    assert starts == sorted(starts)
    # every triangle lies inside its leaf box
    pos = sc.pos[b["flat"]].reshape(-1, 3, 3)
    for (s, c), (lo, hi) in zip(leaves, boxes):
        if c:
            p = pos[s:s + c].reshape(-1, 3)
            assert (p >= lo).all() and (p <= hi).all()


def test_sphere_scenes_build_the_reference_tree(oracle):
    """Spheres (the reference's second primitive type): bbox = origin -/+ r (primitives.cuh:166-169); the rebuilt BVH2 and
    the flat order equal the C oracle's, which equals the reference's (tests/test_oracle_golden.py, known-answer scene)."""
    for sc in (scenes.known_answer_scene(), scenes.sphere_soup(300, 60, 3)):
        oracle.load(sc)
        on, of = oracle.bvh()
        b = capi.bvh_build_host(sc.pos, 4, sc.ptype)
        assert (b["nodes2"] == on).all() and (b["flat"] == of).all()


@pytest.mark.parametrize("make", [lambda: scenes.cornell_scene(2), lambda: scenes.conference_scene(20000),
                                  lambda: scenes.random_soup(37, 2), lambda: scenes.sphere_soup(300, 60, 3),
                                  lambda: scenes.random_soup(3, 1)])
def test_quantised_tree_boxes_contain_their_primitives(make):
    """The ordered traversal's 8-wide tree (host builder): every child box, decoded exactly as the kernel does, contains the
    child's primitives; every primitive is in exactly one leaf; leaves hold 1..3 primitives (tests/treecheck.py)."""
    from treecheck import check_own_tree
    sc = make()
    b = capi.bvh_build_host(sc.pos, 4, sc.ptype)
    flat_sph = (sc.ptype[b["flat"]] == 1) if sc.ptype is not None else np.zeros(sc.n, bool)
    depth = check_own_tree(b["nodes8q"], b["fast_to_flat"], sc.pos[b["flat"]], flat_sph, sc.n)
    assert depth == b["depth8"]


def _leaf_sizes(nodes8q):
    refs = np.asarray(nodes8q).reshape(-1, 24)[:, 4:12].ravel()
    leaves = refs[(refs != 0xFFFFFFFF) & ((refs & 0x80000000) != 0)]
    return np.bincount((leaves >> 27) & 15, minlength=4)


def test_would_be_leaves_are_divided_where_it_pays(monkeypatch):
    """host_bvh.cpp, SahBuilder::build: n <= 3 primitives stay one leaf only if n <= 0.5 + the area-weighted count of their
    best division (MIRT_SAH_LEAF_CT; 1e30 = the fixed rule of rounds 1 - 2).  Same primitives either way, valid trees, fewer
    primitives per leaf with the rule; on two triangles far apart the division always pays, on two coincident ones never."""
    from treecheck import check_own_tree
    sc = scenes.conference_scene(20000)
    sizes = {}
    for ct in ("1e30", "0.5"):
        monkeypatch.setenv("MIRT_SAH_LEAF_CT", ct)
        b = capi.bvh_build_host(sc.pos, 4, sc.ptype)
        check_own_tree(b["nodes8q"], b["fast_to_flat"], sc.pos[b["flat"]], np.zeros(sc.n, bool), sc.n)
        h = _leaf_sizes(b["nodes8q"])
        assert (h * np.arange(len(h))).sum() == sc.n
        sizes[ct] = (h * np.arange(len(h))).sum() / h.sum()
    assert sizes["0.5"] < sizes["1e30"]
    monkeypatch.delenv("MIRT_SAH_LEAF_CT")
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    apart = np.stack([tri, tri + np.float32(100.0)])
    same = np.stack([tri, tri])
    assert _leaf_sizes(capi.bvh_build_host(apart, 4)["nodes8q"])[1] == 2          # two leaves of one
    assert _leaf_sizes(capi.bvh_build_host(same, 4)["nodes8q"])[2] == 1           # one leaf of two


def test_single_leaf_scene():
    sc = scenes.random_soup(3, 1)
    b = capi.bvh_build_host(sc.pos, 4)
    assert len(b["nodes2"]) == 1 and b["depth2"] == 0 and len(b["nodes4"]) == 1


def test_light_and_camera_records(golden):
    u = golden("units")
    rec = capi.light_records(u["light_args"])
    for l, got, want in zip(u["light_args"], rec, u["light_records"]):
        used = 17 if int(l[0]) == 0 else 7
        assert (got[:used] == want[:used]).all()
    lib = capi.load()
    for a, want in zip(u["cam_args"], u["cam_records"]):
        out = capi.CameraRec()
        v = lambda t: (C.c_float * 3)(*[float(x) for x in t])
        f = [C.c_float(float(x)) for x in a[9:15]]
        assert lib.mirt_camera_look_at(v(a[0:3]), v(a[3:6]), v(a[6:9]), *f, C.byref(out)) == 0
        got = np.frombuffer(bytes(out), np.float32)
        assert (got.view(np.uint32) == want.view(np.uint32)).all()


def test_tonemap_table_is_exact(golden):
    u = golden("units")
    assert (capi.tonemap_host(u["film_in"]) == u["film_out"]).all()
    # dense sweep against the direct formula through the C oracle (same host libm)
    from cpulibs import CpuTracer

    o = CpuTracer("oracle")
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(0, 0.75, 30000), np.linspace(0, 0.72, 30000)]).astype(np.float32)
    x = np.stack([x, x[::-1], np.sqrt(x)], 1).astype(np.float32)
    want = np.array([o.update_pixel(c) for c in x[:6000]])
    assert (capi.tonemap_host(x[:6000]) == want).all()


def test_bmp_bytes(golden, tmp_path):
    g = golden("known_answer")
    p = str(tmp_path / "film.bmp")
    capi.save_bmp(p, g["rgb"])
    assert (np.fromfile(p, np.uint8) == g["bmp"]).all()          # PathTracer::save_to_file on the same film
    odd = np.random.default_rng(0).integers(0, 256, (5, 7, 3)).astype(np.int32)   # row padding path
    capi.save_bmp(p, odd)
    raw = np.fromfile(p, np.uint8)
    assert len(raw) == 54 + 5 * 24 and raw[0] == 66 and raw[1] == 77
    assert (raw[54:54 + 21].reshape(7, 3)[:, ::-1] == odd[4]).all()   # bottom-up, BGR


def test_errors_are_loud_without_gpu_or_init():
    import torch

    lib = capi.load()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    with pytest.raises(capi.MirtError) as e:
        capi.init(0)
    assert e.value.code == -2 and "no CPU fallback" in e.value.msg      # MIRT_E_DEVICE
    with pytest.raises(capi.MirtError) as e:
        capi.Scene(scenes.random_soup(8, 1))
    assert e.value.code == -3                                             # MIRT_E_STATE: not initialised
    assert lib.mirt_save_bmp(None, None, 0, 0) == -1
    assert lib.mirt_bvh_build_host(None, 0, 4, None, None, None, None) == -1


def test_threaded_builders_do_not_depend_on_the_thread_count(monkeypatch):
    """The host builders hand the halves of large ranges to threads (host_bvh.cpp): the reference's BVH2, its flat order and
    the traversal's own tree must come out the same on 1, 3 and 16 threads."""
    sc = scenes.blob_scene(192, 96)                     # 36 k triangles: above the size at which ranges are forked
    assert sc.n >= 16384
    out = []
    for t in ("1", "3", "16"):
        monkeypatch.setenv("MIRT_BUILD_THREADS", t)
        out.append(capi.bvh_build_host(sc.pos, 4, sc.ptype))
    for b in out[1:]:
        for key in ("nodes2", "flat", "nodes4", "nodes8q", "fast_to_flat"):
            assert (b[key] == out[0][key]).all(), key


# ---- the kept host API's math types and presets (include/raytracer_amd/linalg.hpp, pathtracer.hpp, physics.hpp)
def _gxx(args, **kw):
    import subprocess

    return subprocess.run(["g++", "-std=c++17", "-ffp-contract=off", "-w", "-I", os.path.join(ROOT, "include")] + args,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, **kw)


@pytest.mark.parametrize("with_particle", [False, True])
def test_linalg_mirror_equals_the_references_types_bitwise(tmp_path, with_particle):
    """oracle/linalg_probe.cpp built against the mirror's Vector3f / Vector2f / Matrix3f prints the same bit patterns as the
    build against the reference's own headers (tests/golden/linalg_probe.txt, written by oracle/Makefile): every operator, and
    40 steps of the particle demo's update for 50 particles - with_particle: through physics.hpp's Particle::step, whose
    reference counterpart cannot be compiled under g++ (particle.cuh:77), so its expressions are what the golden holds."""
    exe = str(tmp_path / "probe")
    r = _gxx(["-O2"] + (["-DWITH_PARTICLE"] if with_particle else []) + ["-o", exe, os.path.join(ROOT, "oracle", "linalg_probe.cpp")])
    assert r.returncode == 0, r.stderr[-2000:]
    import subprocess

    got = subprocess.run([exe], stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines()
    want = open(os.path.join(ROOT, "tests", "golden", "linalg_probe.txt")).read().splitlines()
    assert len(got) == len(want) > 3000
    bad = [(g, w) for g, w in zip(got, want) if g != w]
    assert not bad, bad[:5]


def test_every_preset_and_operator_of_the_kept_api_compiles(tmp_path):
    """A caller's translation unit that names all 19 preset BSDFs of scene/bsdf.cuh:358-388 and the Vector3f operators a
    scene description reaches (missing #1)."""
    src = tmp_path / "caller.cpp"
    src.write_text('''#include "raytracer_amd/pathtracer.hpp"
BSDF *all_presets[] = {&mirror_bsdf, &refract_bsdf, &glass_bsdf, &aluminum_bsdf, &brass_bsdf, &copper_bsdf, &gold_bsdf, &iron_bsdf,
                       &lead_bsdf, &mercury_bsdf, &platinum_bsdf, &silver_bsdf, &titanium_bsdf, &blue_metal_bsdf, &black_bsdf,
                       &white_bsdf, &red_bsdf, &green_bsdf, &blue_bsdf};
int main() {
  Vector3f a(1, 2, 3), b(.5f);
  a /= 2; a *= 3; a -= b; a += b;
  Vector3f c = 2.f * a + a * 2.f - a / 3.f + 1.f / a;
  float d = a.dot(b) + a.norm2() + c.illum() + c[0] + c.r;
  Vector2f p(1, 2); Matrix3f M; Vector3f e = M.T() * c;
  std::vector<Primitive *> prims;
  for (BSDF *m: all_presets) prims.push_back(new Primitive(PrimitiveFactory::createSphere(e, d, m)));
  static_assert(sizeof(Vector3f) == 12 && sizeof(Vector2f) == 8 && sizeof(Matrix3f) == 36, "layouts of the reference (SURVEY 8)");
  return gold_bsdf.rec.type == BSDF::MICROFACET && gold_bsdf.rec.p[0] == 0.05f && (a == a) ? 0 : 1;
}
''')
    r = _gxx(["-fsyntax-only", str(src)])
    assert r.returncode == 0, r.stderr[-2000:]


def test_worker_pool_and_film_widening_native(tmp_path):
    """Host helpers of libmirt that need no GPU (tests/native/host_pool_check.cpp against the library's own sources): the
    persistent worker pool runs every job exactly once, run after run; the vectorised film widening equals the scalar rule
    `r | g << 8 | b << 16 -> three ints` (util/image.cuh:9-48 layout) for every length and alignment."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "raytracer_amd", "csrc")
    exe = str(tmp_path / "host_pool_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", csrc, "-o", exe,
                           os.path.join(root, "tests", "native", "host_pool_check.cpp"),
                           os.path.join(csrc, "host_bvh.cpp"), os.path.join(csrc, "host_film.cpp")])
    # several instances at once: on oversubscribed cores workers wake up late, after their run() has returned - the case a first
    # version got wrong (a late worker paired the old job count with the next run's claim counter: 596 crashes in 720 runs)
    for _ in range(3):
        procs = [subprocess.Popen([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(12)]
        for p in procs:
            out, err = p.communicate(timeout=300)
            assert p.returncode == 0 and out.strip() == "ok", (p.returncode, out + err)
