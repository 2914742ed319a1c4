"""Device builder of the ordered traversal's own tree (raytracer_amd/csrc/build.hip) against the same
bars as the host builder: structure (conservative boxes, every primitive once), hit parity with the reference through the
C oracle, films identical to the host-built scene's.  Need a real MI355X: `pytest -m gpu`."""
import numpy as np
import pytest

from raytracer_amd import scenes
from gpu_common import SCENES, SPHERE_SCENES, random_rays
from treecheck import check_own_tree

pytestmark = pytest.mark.gpu

BUILD_SCENES = {
    "cornell_mixed": lambda: scenes.cornell_scene(2, "mixed"),
    "soup37": lambda: scenes.random_soup(37, 2),
    "soup3": lambda: scenes.random_soup(3, 1),
    "soup2": lambda: scenes.random_soup(2, 4),
    "conf20k": lambda: scenes.conference_scene(20000),
    "sphere_soup": lambda: scenes.sphere_soup(300, 60, 3),
    "blob64": lambda: scenes.blob_scene(64, 32),
}


DEVICE_BUILDERS = {"device": 1, "ploc": 2}


@pytest.mark.parametrize("builder", list(DEVICE_BUILDERS))
@pytest.mark.parametrize("key", list(BUILD_SCENES))
def test_device_tree_structure(mirt, key, builder):
    sc = BUILD_SCENES[key]()
    s = mirt.Scene(sc, builder=builder)
    counts, nodes, prims = s.own_tree()
    assert counts["builder"] == DEVICE_BUILDERS[builder] and counts["prims"] == sc.n and counts["nodes"] >= 1
    _, flat = s.bvh2()
    flat_sph = (sc.ptype[flat] == 1) if sc.ptype is not None else np.zeros(sc.n, bool)
    f2f = prims[:, 9]
    depth = check_own_tree(nodes, f2f, sc.pos[flat], flat_sph, sc.n)
    assert depth == counts["depth"]
    # the gathered records are the flat-order records (host-built scene: the same records in SAH order)
    h = mirt.Scene(sc, builder="host")
    hc, _, hprims = h.own_tree()
    assert hc["builder"] == 0
    assert (prims[np.argsort(f2f)] == hprims[np.argsort(hprims[:, 9])]).all()
    s.close()
    h.close()


@pytest.mark.parametrize("builder", list(DEVICE_BUILDERS))
@pytest.mark.parametrize("key", ["cornell_mixed", "soup500", "conf20k", "sphere_soup", "known_answer"])
def test_device_tree_traces_like_the_reference(mirt, oracle, key, builder):
    """Traversal 0 over the device-built tree: closest hits (index and t bits) and occlusion bit-exact with the reference."""
    sc = (SCENES.get(key) or SPHERE_SCENES[key])()
    oracle.load(sc)
    s = mirt.Scene(sc, builder=builder)
    for unit in (False, True):
        rays = random_rays(sc, 30000, 15 + unit, unit)
        ci, ct, _ = oracle.trace(rays, normals=True)
        ai, _ = oracle.trace(rays, anyhit=True)
        gi, gt, _ = s.trace(rays, traversal=0, bary=True)
        assert (gi == ci).all() and (gt == ct).all(), f"closest, unit={unit}: {(gi != ci).sum()} differ"
        hi, _ = s.trace(rays, anyhit=True, traversal=0)
        assert (hi == ai).all()
    s.close()


def test_device_and_host_built_scenes_render_the_same_film(mirt):
    """The tree is invisible in the result: same accumulators, bit for bit, whoever built it."""
    import torch
    sc = scenes.conference_scene(20000)
    w, h = 320, 180
    out = []
    for builder in ("host", "device", "ploc"):
        s = mirt.Scene(sc, builder=builder)
        acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        st = s.render_accum(w, h, acc.data_ptr(), spp=8, depth=6, rng_mode=1)
        assert st["traversal_used"] == 0
        out.append((acc.cpu(), st["closest_rays"], st["shadow_rays"]))
        s.close()
    for o in out[1:]:
        assert out[0][1:] == o[1:]
        assert torch.equal(out[0][0], o[0])


@pytest.mark.parametrize("builder", list(DEVICE_BUILDERS))
def test_device_tree_one_million_triangles(mirt, builder):
    """buddha* (1 046 530 triangles) at 1080p: 2 M primary hits of the ordered traversal over the device-built tree equal the
    reference walk's; build times of both builders are printed (pytest -s) for DESIGN.md."""
    sc = scenes.blob_scene(1024, 512, "buddha*")
    sc.camera = sc.camera.for_aspect(1920, 1080)
    s = mirt.Scene(sc, builder=builder)
    bt = s.build_times()
    counts, _, _ = s.own_tree()
    print(builder, "builder:", counts, {k: round(v, 2) for k, v in bt.items()})
    i2, t2 = s.primary_hits(1920, 1080, traversal=2)
    i0, t0 = s.primary_hits(1920, 1080, traversal=0)
    assert (i0 == i2).all() and (t0 == t2).all(), f"{(i0 != i2).sum()} differ"
    assert bt["own_tree_ms"] < 500 and bt["dev_fold_ms"] > 0
    s.close()


def test_non_finite_scene_falls_back_like_the_host_builder(mirt):
    sc = scenes.random_soup(50, 3)
    sc.pos = sc.pos.copy()
    sc.pos[7, 4] = np.inf
    s = mirt.Scene(sc, builder="device")
    counts, _, _ = s.own_tree()
    assert counts["nodes"] == 0                       # no own tree: traversal 0 runs as 1, as with the host builder
    idx, _ = s.primary_hits(32, 32, traversal=0)
    idx1, _ = s.primary_hits(32, 32, traversal=1)
    assert (idx == idx1).all()
    s.close()


@pytest.mark.parametrize("builder", list(DEVICE_BUILDERS))
def test_identical_and_clustered_primitives(mirt, builder):
    """Degenerate input: 3 000 copies of one triangle (equal Morton keys, equal merge costs everywhere: the radix tree splits
    by index, the clustering pairs up (2k, 2k + 1)) next to 1 000 ordinary ones.  The reference puts the copies into one
    oversized leaf (bvh.cuh:216-220); hits must still be the reference walk's, ties included (last copy wins)."""
    sc = scenes.random_soup(4000, 9)
    sc.pos = sc.pos.copy()
    sc.pos[:3000] = sc.pos[0]
    s = mirt.Scene(sc, builder=builder)
    counts, nodes, prims = s.own_tree()
    assert counts["builder"] == DEVICE_BUILDERS[builder] and counts["nodes"] > 0
    _, flat = s.bvh2()
    check_own_tree(nodes, prims[:, 9], sc.pos[flat], np.zeros(sc.n, bool), sc.n)
    rays = random_rays(sc, 20000, 3, True)
    c = sc.pos[0].reshape(3, 3).mean(0)
    rays[:5000, 3:6] = c - rays[:5000, :3]                          # aimed at the stack of copies
    rays[:5000, 3:6] /= np.linalg.norm(rays[:5000, 3:6], axis=1, keepdims=True)
    rays[:5000, 7] = np.inf
    i2, t2 = s.trace(rays, traversal=2)
    i0, t0 = s.trace(rays, traversal=0)
    assert (i0 == i2).all() and (t0 == t2).all()
    assert (i2[:5000] >= 0).mean() > 0.5
    s.close()
