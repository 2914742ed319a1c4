"""Parity tests proper: the HIP path (through the C ABI, raytracer_amd/capi.py -> libmirt.so) against the CPU oracle and the
golden vectors from the reference.  Need a real MI355X: `pytest -m gpu`.

Bars (written where they are applied):
  * hit indices, t bit patterns, ray counts, BVH arrays, tonemap, BMP bytes: bit-exact;
  * film bytes in reference-RNG mode: >= 99.9 % of the channels identical to the reference's film (the rest differ because
    the device libm's sinf/cosf/acosf round differently in the last ulp, which moves a bounce direction by one ulp);
  * float radiance in counter-RNG mode: relative RMSE vs the reference <= 1.5 x the reference-vs-reference rRMSE at equal
    spp and mean-luminance bias <= 1 %.
"""
import numpy as np
import pytest

from gpu_common import SCENES, SPHERE_SCENES, film_delta, random_rays
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("key", ["cornell_diffuse", "cornell_mixed", "blob64", "soup500", "conf20k"])
def test_primary_hits_bit_exact_vs_reference_golden(mirt, golden, key):
    """BASELINE north_star: primary-ray hit indices bit-exact with the reference -- in every traversal mode."""
    g = golden("renders")
    w, h = int(g[key + "_params"][0]), int(g[key + "_params"][1])
    s = mirt.Scene(SCENES[key]())
    for trav in (0, 1, 2):
        idx, tb = s.primary_hits(w, h, traversal=trav)
        assert (idx == g[key + "_hit_idx"]).all(), f"traversal {trav}: {(idx != g[key + '_hit_idx']).sum()} indices differ"
        assert (tb == g[key + "_hit_tbits"]).all(), f"traversal {trav}: t bits differ"
    s.close()


def test_primary_hits_full_size_vs_oracle(mirt, oracle):
    """bunny* at 1920x1080 (BASELINE configs[1] geometry): 2 M primary rays, all three traversals vs the C oracle."""
    sc = scenes.blob_scene(264, 132)
    sc.camera = sc.camera.for_aspect(1920, 1080)
    oracle.load(sc)
    s = mirt.Scene(sc)
    w, h = 1920, 1080
    idx2, tb2 = s.primary_hits(w, h, traversal=2)
    # the oracle is single-threaded here: check a 480 x 270 window-equivalent render fully, and the full frame across modes
    oi, ot = oracle.primary_hits(480, 270)
    gi, gt = s.primary_hits(480, 270, traversal=0)
    assert (oi == gi).all() and (ot == gt).all()
    for trav in (0, 1):
        idx, tb = s.primary_hits(w, h, traversal=trav)
        assert (idx == idx2).all() and (tb == tb2).all(), f"traversal {trav} vs reference walk: {(idx != idx2).sum()} differ"
    assert (idx2 >= 0).mean() > 0.3
    s.close()


@pytest.mark.parametrize("key", ["cornell_mixed", "soup500", "conf20k", "sphere_soup", "known_answer"])
def test_trace_arbitrary_rays(mirt, oracle, key):
    sc = (SCENES.get(key) or SPHERE_SCENES[key])()
    oracle.load(sc)
    s = mirt.Scene(sc)
    for unit in (False, True):
        rays = random_rays(sc, 30000, 5 + unit, unit)
        ci, ct, cn = oracle.trace(rays, normals=True)
        ai, at = oracle.trace(rays, anyhit=True)
        for trav in (2, 1, 0):
            gi, gt, gb = s.trace(rays, traversal=trav, bary=True)
            assert (gi == ci).all() and (gt == ct).all(), f"closest, traversal {trav}, unit={unit}: {(gi != ci).sum()} differ"
            hi, ht = s.trace(rays, anyhit=True, traversal=trav)
            assert (hi == ai).all(), f"any-hit, traversal {trav}"
            if trav != 0:
                assert (ht == at).all()          # which primitive ends an any-hit walk is order dependent in mode 0
        assert (ci >= 0).sum() > 1000 and (ci < 0).sum() > 100
    s.close()


def test_known_answer_scene_with_a_sphere(mirt, oracle, golden):
    """sphere + floor, Parameters{4,1,64,0.05,4,4}, 64x48; the reference's film sums to 950192.  Sphere
    primitives (primitives.cuh:64-85, :111-136) through the same kernels: hits bit-exact, film within the libm bar."""
    g = golden("known_answer")
    sc = scenes.known_answer_scene()
    s = mirt.Scene(sc)
    for trav in (0, 1, 2):
        idx, tb = s.primary_hits(64, 48, traversal=trav)
        assert (idx == g["hit_idx"]).all() and (tb == g["hit_tbits"]).all(), f"traversal {trav}"
    for trav in (0, 1):
        img = s.render(64, 48, spp=4, spl=1, batch=64, tol=0.05, depth=4, traversal=trav)
        assert (img == g["rgb"]).mean() >= 0.999, f"only {(img == g['rgb']).mean():.5f} of the film bytes are identical"
        assert abs(int(img.sum()) - int(g["checksum"])) <= 0.001 * int(g["checksum"])
    s.close()
    # a mixed scene: primary hits against the oracle in every traversal
    sc = scenes.sphere_soup(300, 60, 3)
    oracle.load(sc)
    oi, ot = oracle.primary_hits(160, 120)
    s = mirt.Scene(sc)
    for trav in (0, 1, 2):
        gi, gt = s.primary_hits(160, 120, traversal=trav)
        assert (gi == oi).all() and (gt == ot).all(), f"traversal {trav}: {(gi != oi).sum()} differ"
    assert len(np.unique(oi)) > 50
    img = s.render(160, 120, spp=4, spl=1, batch=64, tol=0.0, depth=4)
    ref = oracle.render(160, 120, spp=4, spl=1, batch=64, tol=0.0, depth=4)
    assert (img == ref).mean() >= 0.999
    s.close()


def test_overlapping_geometry_overflows_the_lds_stack(mirt, oracle):
    """2000 large triangles in a small volume: every box overlaps every ray, so the 8-wide traversal pushes far more than
    the 12 entries a lane keeps in LDS (global overflow region), and 64 coincident triangles tie on every hit."""
    sc = scenes.random_soup(2000, 4, extent=0.3, tri_size=1.0)
    sc.pos[100:164] = sc.pos[100]                              # 64 coincident triangles
    sc = sc.finalize()
    oracle.load(sc)
    s = mirt.Scene(sc)
    rays = random_rays(sc, 20000, 11, True)
    ci, ct = oracle.trace(rays)
    ai, _ = oracle.trace(rays, anyhit=True)
    for trav in (0, 1, 2):
        gi, gt = s.trace(rays, traversal=trav)
        assert (gi == ci).all() and (gt == ct).all(), f"closest, traversal {trav}: {(gi != ci).sum()} differ"
        hi, _ = s.trace(rays, anyhit=True, traversal=trav)
        assert (hi == ai).all(), f"any-hit, traversal {trav}"
    assert (ci >= 100).sum() > 0 and ((ci >= 100) & (ci < 164)).sum() >= 0
    s.close()


def test_capped_overflow_region_hands_deep_rays_to_the_reference_walk(mirt, oracle, monkeypatch):
    """The global overflow region of the traversal stack is capped (MIRT_OVF_CAP levels beyond the LDS part, default 24): a
    ray that outgrows it is finished by the reference walk (k_redo) instead of failing the launch.  With a cap of ONE level
    on the all-overlapping soup most rays take that route in traversals 0 and 1, and every result still equals the oracle's."""
    sc = scenes.random_soup(2000, 4, extent=0.3, tri_size=1.0).finalize()
    oracle.load(sc)
    monkeypatch.setenv("MIRT_OVF_CAP", "1")                    # read when the scene is created
    s = mirt.Scene(sc)
    monkeypatch.delenv("MIRT_OVF_CAP")
    rays = random_rays(sc, 20000, 12, True)
    ci, ct = oracle.trace(rays)
    ai, _ = oracle.trace(rays, anyhit=True)
    for trav in (0, 1):
        gi, gt = s.trace(rays, traversal=trav)
        assert (gi == ci).all() and (gt == ct).all(), f"closest, traversal {trav}: {(gi != ci).sum()} differ"
        hi, _ = s.trace(rays, anyhit=True, traversal=trav)
        assert (hi == ai).all(), f"any-hit, traversal {trav}"
    import torch
    # the cap bites: a rendered frame reports far more reference-walk rays than the same frame with the default region
    sc2 = scenes.conference_scene(20000)
    monkeypatch.setenv("MIRT_OVF_CAP", "1")
    capped = mirt.Scene(sc2)
    monkeypatch.delenv("MIRT_OVF_CAP")
    plain = mirt.Scene(sc2)
    w, h = 160, 90
    a0 = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    a1 = torch.zeros_like(a0)
    kw = dict(spp=4, spl=1, batch=64, tol=0.0, depth=4, rng_mode=1, traversal=0)
    st0 = plain.render_accum(w, h, a0.data_ptr(), **kw)
    st1 = capped.render_accum(w, h, a1.data_ptr(), **kw)
    torch.cuda.synchronize()
    assert torch.equal(a0, a1)
    assert st1["exact_fallback_rays"] > st0["exact_fallback_rays"], (st0["exact_fallback_rays"], st1["exact_fallback_rays"])
    for x in (s, capped, plain):
        x.close()


def test_six_shadow_rays_per_hit(mirt, oracle):
    """3 lights x 2 samples per light: the generic shadow-fold and emission loops of k_logic (the bench has K = 2)."""
    sc = scenes.cornell_scene(1, "diffuse")
    sc.lights = np.array([scenes.area_light((12, 12, 12), (0, 2.99, 0), (0, -1, 0), (1, 0, 0), (0, 0, 1)),
                          scenes.point_light((1.5, 1.5, 1.5), (1.5, 2.5, -1.5)),
                          scenes.directional_light((.3, .3, .3), (-1, -1, 1))])
    sc = sc.finalize()
    oracle.load(sc)
    oracle.set_camera(sc.camera)
    s = mirt.Scene(sc)
    w, h = 96, 64
    ref = oracle.render(w, h, spp=3, spl=2, batch=64, tol=0.0, depth=4)
    img = s.render(w, h, spp=3, spl=2, batch=64, tol=0.0, depth=4)
    assert (img == ref).mean() >= 0.999
    assert (s.last["closest_rays"], s.last["shadow_rays"]) == (oracle.last["closest"], oracle.last["shadow"])
    assert s.last["shadow_rays"] > 6 * w * h
    s.close()


def test_trace_edge_cases(mirt, oracle):
    sc = scenes.cornell_scene(1, "diffuse")
    oracle.load(sc)
    s = mirt.Scene(sc)
    assert s.trace(np.zeros((0, 8), np.float32))[0].shape == (0,)                       # empty batch
    rays = np.array([
        [0, 1, 0, 0, 0, 0, 1e-5, np.inf],            # zero direction: NaN everywhere
        [0, 1, 0, np.nan, 1, 0, 1e-5, np.inf],       # NaN direction
        [np.inf, 1, 0, 1, 0, 0, 1e-5, np.inf],       # non-finite origin
        [0, 1, 0, 0, -1, 0, 1e-5, np.inf],           # straight down: axis aligned, hits the floor
        [2, 1, 0, 0, 1, 0, 1e-5, np.inf],            # origin IN the right wall's plane, direction in the plane
        [0, 1, 0, 1, 0, 0, 1e-5, 0.5],               # max_t shorter than the distance to the wall
        [0, 1, 0, 1, 0, 0, 5.0, np.inf],             # min_t beyond every hit
        [0, 1, 0, 1e-30, 1, 1e-30, 1e-5, np.inf],    # denormal-ish components: 1/x overflows
    ], np.float32)
    for anyhit in (False, True):
        want = oracle.trace(rays, anyhit=anyhit)
        for trav in (0, 1, 2):
            got = s.trace(rays, anyhit=anyhit, traversal=trav)
            assert (got[0] == want[0]).all(), (anyhit, trav, got[0], want[0])
            same_t = (got[1] == want[1]) | (np.isnan(got[1].view(np.float32)) & np.isnan(want[1].view(np.float32)))
            assert same_t.all(), (anyhit, trav)
    s.close()


FILM_BY_MORE = 1e-4          # share of film bytes allowed to differ by more than one level (measured on all 16 reference films,
                             # both traversals: 0 differing bytes at all, profiles/r3_film_delta.txt)


@pytest.mark.parametrize("key", list(SCENES))
def test_film_reference_rng_vs_reference_golden(mirt, golden, key):
    """Whole frames with the reference's RNG stream vs the film the reference itself rendered (float-abs variant, the
    semantics of the reference's CUDA build)."""
    g = golden("renders")
    w, h, spp, spl, batch, depth = (int(v) for v in g[key + "_params"])
    tol = float(g[key + "_tol"])
    s = mirt.Scene(SCENES[key]())
    for trav in (0, 1):
        img = s.render(w, h, spp=spp, spl=spl, batch=batch, tol=tol, depth=depth, traversal=trav)
        d = np.abs(img - g[key + "_rgb_fabs"])
        assert (d == 0).mean() >= 0.999, f"{key}: only {(d == 0).mean():.5f} of the film bytes are identical"
        fd = film_delta(f"renders/{key} traversal {trav}", img, g[key + "_rgb_fabs"])
        # the bytes that differ: off by ONE tonemap level (an ulp of the device libm), except where a path changed (a bounce
        # direction an ulp away crossed a silhouette): those pixels are counted, and bounded
        assert fd["by_more"] <= FILM_BY_MORE * fd["bytes"], fd
        assert abs(img.mean() - g[key + "_rgb_fabs"].mean()) < 0.05
        assert s.last["tonemap_table_exact"] == 1
        assert s.last["traversal_used"] == trav           # the scene's own tree was built: no silent fallback to mode 1
    # the g++ build's integer abs() in GLASS, behind its option
    img = s.render(w, h, spp=spp, spl=spl, batch=batch, tol=tol, depth=depth, glass_int_abs=1)
    assert (img == g[key + "_rgb"]).mean() >= 0.999
    s.close()


def test_ray_counts_match_reference_control_flow(mirt, oracle):
    """Same RNG stream => same paths => the closest/shadow ray counts are equal, not just close."""
    sc = scenes.cornell_scene(2, "diffuse")
    oracle.load(sc)
    oracle.render(96, 72, spp=4, spl=2, depth=6)
    s = mirt.Scene(sc)
    s.render(96, 72, spp=4, spl=2, depth=6)
    assert s.last["closest_rays"] == oracle.last["closest"]
    assert s.last["shadow_rays"] == oracle.last["shadow"]
    assert s.last["paths"] == 96 * 72 * 4
    s.close()


def test_counter_rng_film_vs_oracle_counter_mode(mirt, oracle):
    sc = scenes.conference_scene(20000)
    oracle.set_glass_float_abs(1)
    oracle.load(sc)
    oracle.set_rng(1, 7)
    ref = oracle.render(96, 54, spp=4, depth=8, batch=99, nthreads=4)
    s = mirt.Scene(sc)
    img = s.render(96, 54, spp=4, depth=8, batch=99, rng_mode=1, seed=7)
    assert (img == ref).mean() >= 0.999
    assert s.last["closest_rays"] == oracle.last["closest"] and s.last["shadow_rays"] == oracle.last["shadow"]
    img2 = s.render(96, 54, spp=4, depth=8, batch=99, rng_mode=1, seed=8)
    assert (img2 != img).mean() > 0.3                              # the seed matters
    oracle.set_glass_float_abs(0)
    s.close()


def _rrmse(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.mean(b))


def test_radiance_rrmse_vs_reference(mirt, golden):
    """Converged radiance within tolerance: rRMSE(gpu, ref) <= 1.5 x rRMSE(ref, ref') at equal spp, and the
    mean luminance within 1 %.  GPU uses its own counter-based sample set."""
    import torch

    g = golden("radiance")
    w, h, spp, depth = (int(v) for v in g["params"])
    a, b = g["cornell_a"], g["cornell_b"]
    s = mirt.Scene(scenes.cornell_scene(2, "diffuse"))
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    s.render_accum(w, h, acc.data_ptr(), spp=spp, batch=spp + 1, depth=depth, rng_mode=1, seed=3)
    acc = acc.cpu().numpy()
    assert (acc[..., 3] == spp).all()
    rad = acc[..., :3] / acc[..., 3:4]
    base = _rrmse(a, b)
    assert _rrmse(rad, a) <= 1.5 * base and _rrmse(rad, b) <= 1.5 * base, (_rrmse(rad, a), base)
    lum = lambda x: float((x @ np.array([0.2126, 0.7152, 0.0722])).mean())
    assert abs(lum(rad) / lum(0.5 * (a + b)) - 1) <= 0.01
    s.close()


def test_tile_partition_sums_to_the_single_gpu_frame(mirt):
    import torch

    sc = scenes.cornell_scene(2, "mixed")
    s = mirt.Scene(sc)
    w, h = 101, 67                                                 # ragged: not multiples of the 8x8 tile
    kw = dict(spp=3, depth=5, batch=4)
    full = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    s.render_accum(w, h, full.data_ptr(), **kw)
    total = torch.zeros_like(full)
    part = torch.zeros_like(full)
    owned = torch.zeros((h, w), dtype=torch.int32, device="cuda:0")
    from raytracer_amd import dist as rdist

    own = rdist.tile_owner(w, h, 3)
    for r in range(3):
        s.render_accum(w, h, part.data_ptr(), tile_rank=r, tile_world=3, **kw)
        p = part.cpu().numpy()
        assert ((p[..., 3] != 0) == (own == r)).all()              # exactly this rank's tiles, zeros elsewhere
        total += part
    assert torch.equal(total, full)                                # bit-identical to one GPU
    rgb = mirt.tonemap(total.data_ptr(), w, h)
    assert (rgb == s.render(w, h, **kw)).all()
    s.close()


@pytest.mark.parametrize("rng_mode", [0, 1])
def test_adaptive_stop_survives_the_tile_partition(mirt, rng_mode):
    """The reference's per-pixel adaptive stop (raytracer.cuh:115-124) needs a pixel's samples in sequence on one device: the
    tile partition keeps them there, so it stays ON (spp slices / sample streams / progressive passes switch it off).  Two
    tile partitions of an adaptive render add up to the single-context adaptive frame bit for bit, film included, and the
    stop did fire (divisors below spp: on an early break the reference divides by i - 1, Q4)."""
    import torch

    s = mirt.Scene(scenes.cornell_scene(1, "diffuse"))
    w, h = 75, 50
    kw = dict(spp=48, depth=4, batch=8, tol=0.5, rng_mode=rng_mode, seed=3 if rng_mode else 0)
    full = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    st = dict(s.render_accum(w, h, full.data_ptr(), **kw))
    div = full[..., 3].cpu().numpy()
    assert (div < 48).mean() > 0.2 and (div == 48).any(), "the adaptive stop should fire for some pixels and not for others"
    assert st["paths"] < w * h * 48
    total = torch.zeros_like(full)
    part = torch.zeros_like(full)
    paths = 0
    for r in range(2):
        stp = s.render_accum(w, h, part.data_ptr(), tile_rank=r, tile_world=2, **kw)
        paths += stp["paths"]
        total += part
    assert torch.equal(total, full) and paths == st["paths"]
    assert (mirt.tonemap(total.data_ptr(), w, h) == s.render(w, h, **kw)).all()
    s.close()


def test_spp_slices_sum_to_the_full_frame(mirt):
    import torch

    s = mirt.Scene(scenes.cornell_scene(1, "diffuse"))
    w, h, spp = 64, 40, 10
    kw = dict(depth=4, batch=99, rng_mode=1, seed=5)
    full = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    s.render_accum(w, h, full.data_ptr(), spp=spp, **kw)
    total = torch.zeros_like(full)
    part = torch.zeros_like(full)
    from raytracer_amd import dist as rdist

    for begin, count in rdist.spp_slices(spp, 4):
        s.render_accum(w, h, part.data_ptr(), spp=spp, sample_begin=begin, sample_count=count, **kw)
        total += part
    assert torch.equal(total[..., 3], full[..., 3])
    torch.testing.assert_close(total, full, rtol=2e-6, atol=1e-7)   # same sample set, re-associated float sums
    with pytest.raises(mirt.MirtError):
        s.render_accum(w, h, part.data_ptr(), spp=spp, depth=4, rng_mode=0, sample_begin=2, sample_count=2)
    s.close()


def test_sample_streams_render_the_same_samples(mirt):
    """mirt_options.streams: S streams per pixel in flight, partial sums added in stream order.  Same samples as one
    stream; the float sum is re-associated, so: ray counts equal, accumulators equal to 1e-5 relative, film >= 99.9 %."""
    torch = pytest.importorskip("torch")
    sc = scenes.cornell_scene(2, "mixed")
    s = mirt.Scene(sc)
    w, h, spp = 200, 120, 13                       # 13: streams get unequal shares
    kw = dict(spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=5, rng_mode=1, seed=7)
    ref = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    st0 = dict(s.render_accum(w, h, ref.data_ptr(), **kw))
    img0 = mirt.tonemap(ref.data_ptr(), w, h)
    for streams in (2, 4, 16):
        acc = torch.zeros_like(ref)
        st = s.render_accum(w, h, acc.data_ptr(), streams=streams, **kw)
        assert (st["closest_rays"], st["shadow_rays"], st["paths"]) == (st0["closest_rays"], st0["shadow_rays"], st0["paths"])
        assert st["iterations"] < st0["iterations"]
        a, b = acc.cpu().numpy(), ref.cpu().numpy()
        assert (a[..., 3] == spp).all()
        assert np.allclose(a[..., :3], b[..., :3], rtol=1e-5, atol=1e-6)
        assert (mirt.tonemap(acc.data_ptr(), w, h) == img0).mean() >= 0.999
        # deterministic: a second run gives the same bits
        acc2 = torch.zeros_like(ref)
        s.render_accum(w, h, acc2.data_ptr(), streams=streams, **kw)
        assert torch.equal(acc, acc2)
    with pytest.raises(mirt.MirtError):
        s.render_accum(w, h, ref.data_ptr(), spp=4, rng_mode=0, streams=2)      # sequential RNG cannot be split
    s.close()


def test_full_size_properties_on_the_bench_workload(mirt):
    """BASELINE size (conference* 320 k triangles, 1920x1080), where the CPU oracle would take minutes: properties that do not
    need it.  (1) all three traversals agree bit for bit on 2 M primary rays (mode 2 IS the reference's walk); (2) the frame
    is deterministic; (3) the tile partition sums to it bit for bit; (4) spp slices, sample streams and progressive passes
    trace exactly the same rays and agree to float re-association."""
    torch = pytest.importorskip("torch")
    sc = scenes.conference_scene()
    w, h, spp = 1920, 1080, 8
    sc.camera = sc.camera.for_aspect(w, h)
    s = mirt.Scene(sc)
    i2, t2 = s.primary_hits(w, h, traversal=2)
    for trav in (0, 1):
        i, t = s.primary_hits(w, h, traversal=trav)
        assert (i == i2).all() and (t == t2).all(), f"traversal {trav}: {(i != i2).sum()} of {w * h} primary hits differ"
    assert (i2 >= 0).all()                                           # closed room
    kw = dict(spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=8, rng_mode=1)
    full = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    st = dict(s.render_accum(w, h, full.data_ptr(), **kw))
    rays = (st["closest_rays"], st["shadow_rays"], st["paths"])
    assert st["paths"] == w * h * spp and st["traversal_used"] == 0
    again = torch.zeros_like(full)
    s.render_accum(w, h, again.data_ptr(), **kw)
    assert torch.equal(full, again)
    tiles = torch.zeros_like(full)
    part = torch.zeros_like(full)
    n = [0, 0, 0]
    for r in range(2):
        stp = s.render_accum(w, h, part.data_ptr(), tile_rank=r, tile_world=2, **kw)
        tiles += part
        n = [a + b for a, b in zip(n, (stp["closest_rays"], stp["shadow_rays"], stp["paths"]))]
    assert torch.equal(full, tiles) and tuple(n) == rays
    for variant in (dict(streams=8), dict(streams=3)):
        acc = torch.zeros_like(full)
        stv = s.render_accum(w, h, acc.data_ptr(), **kw, **variant)
        assert (stv["closest_rays"], stv["shadow_rays"], stv["paths"]) == rays
        assert torch.allclose(acc, full, rtol=2e-5, atol=1e-6)
    prog = torch.zeros_like(full)
    n = [0, 0, 0]
    for p in range(2):
        stp = s.render_accum(w, h, prog.data_ptr(), sample_begin=4 * p, sample_count=4, accumulate=p > 0, streams=4, **kw)
        n = [a + b for a, b in zip(n, (stp["closest_rays"], stp["shadow_rays"], stp["paths"]))]
    assert tuple(n) == rays and torch.allclose(prog, full, rtol=2e-5, atol=1e-6)
    film = mirt.tonemap(full.data_ptr(), w, h)
    assert (mirt.tonemap(prog.data_ptr(), w, h) == film).mean() >= 0.999
    assert 20 < film.mean() < 235
    s.close()


def test_one_million_triangles_all_traversals_agree(mirt):
    """buddha* (1 046 530 triangles, BASELINE configs[2] geometry) at 1080p: the ordered traversal of the quantised tree, the
    folded reference tree and the reference walk give the same 2 M primary hits and t bits."""
    sc = scenes.blob_scene(1024, 512, "buddha*")
    sc.camera = sc.camera.for_aspect(1920, 1080)
    s = mirt.Scene(sc)
    i2, t2 = s.primary_hits(1920, 1080, traversal=2)
    for trav in (0, 1):
        i, t = s.primary_hits(1920, 1080, traversal=trav)
        assert (i == i2).all() and (t == t2).all(), f"traversal {trav}: {(i != i2).sum()} differ"
    assert 0.2 < (i2 >= 0).mean() < 1.0
    s.close()


def test_progressive_passes_add_up(mirt):
    """progressive output = passes of a few samples added into one float buffer (flags bit 2), a
    tonemapped snapshot after each.  Four passes of 4 spp equal, bit for bit, the sum of the same four slices rendered
    separately, and the film of the last snapshot is the 16-spp film (>= 99.9 %: the sum is associated per pass)."""
    torch = pytest.importorskip("torch")
    sc = scenes.cornell_scene(2, "mixed")
    s = mirt.Scene(sc)
    w, h = 160, 96
    kw = dict(spp=16, spl=1, batch=17, tol=0.0, depth=5, rng_mode=1, seed=3)
    prog = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    manual = torch.zeros_like(prog)
    part = torch.zeros_like(prog)
    snaps = []
    for p in range(4):
        s.render_accum(w, h, prog.data_ptr(), sample_begin=4 * p, sample_count=4, accumulate=p > 0, streams=2, **kw)
        snaps.append(mirt.tonemap(prog.data_ptr(), w, h))
        s.render_accum(w, h, part.data_ptr(), sample_begin=4 * p, sample_count=4, streams=2, **kw)
        manual += part
        assert (prog[..., 3] == 4 * (p + 1)).all()
    assert torch.equal(prog, manual)
    full = torch.zeros_like(prog)
    s.render_accum(w, h, full.data_ptr(), **kw)
    assert (snaps[-1] == mirt.tonemap(full.data_ptr(), w, h)).mean() >= 0.999
    assert np.abs(snaps[0].astype(int) - snaps[-1].astype(int)).mean() > 0.5          # the first snapshot is noisier
    s.close()


def test_device_tonemap_and_bmp(mirt, golden, tmp_path):
    import torch

    u = golden("units")
    x = u["film_in"]
    n = len(x)
    acc = torch.ones((n, 4), dtype=torch.float32)
    acc[:, :3] = torch.from_numpy(x)
    acc = acc.cuda()
    rgb = mirt.tonemap(acc.data_ptr(), n, 1).reshape(n, 3)
    assert (rgb == u["film_out"]).all()                             # device table search == reference powf film
    g = golden("known_answer")
    p = str(tmp_path / "k.bmp")
    mirt.save_bmp(p, g["rgb"])
    assert (np.fromfile(p, np.uint8) == g["bmp"]).all()


def test_structure_on_device_matches_golden(mirt, golden):
    import hashlib

    g = golden("bvh")
    sc = scenes.blob_scene(264, 132)
    s = mirt.Scene(sc)
    nodes, flat = s.bvh2()
    h = hashlib.sha256()
    h.update(nodes.tobytes())
    h.update(flat.tobytes())
    assert h.hexdigest() == str(g["bunny_sha"])
    info = s.info()
    assert [info["n_tris"], info["bvh2_nodes"], info["bvh2_depth"]] == list(g["bunny_info"])
    s.close()


def test_errors_and_guards(mirt):
    import ctypes as C

    sc = scenes.cornell_scene(1, "diffuse")
    s = mirt.Scene(sc)
    with pytest.raises(mirt.MirtError) as e:
        s.render(0, 10)
    assert "SAMPLE BUFFER EMPTY" in e.value.msg                     # renderer.cuh:130
    with pytest.raises(mirt.MirtError):
        s.render(8, 8, batch=0)                                     # the reference would divide by zero (i % 0)
    with pytest.raises(mirt.MirtError):
        s.render(8, 8, traversal=7)
    nolights = scenes.cornell_scene(1, "diffuse")
    nolights.lights = nolights.lights[:0]
    s2 = mirt.Scene(nolights)
    with pytest.raises(mirt.MirtError) as e:
        s2.render(8, 8)
    assert "LIGHTS NOT SET" in e.value.msg                          # renderer.cuh:126
    bad = scenes.cornell_scene(1, "diffuse")
    bad.mat = bad.mat.copy()
    bad.mat[3] = 99
    with pytest.raises(mirt.MirtError):
        mirt.Scene(bad)
    lib = mirt.load()
    assert lib.mirt_render(None, None, None, None, 8, 8, None, None) < 0
    # thin lens + near/far clip go through the same path
    sc.camera = scenes.Camera((0, 1.5, -1.95), (0, 1.2, 0), hfov=70, vfov=55, nclip=0.2, fclip=3.0, aperture=0.05)
    s.set_camera(sc.camera)
    img = s.render(32, 24, spp=2, depth=2)
    assert img.shape == (24, 32, 3) and img.max() > 0
    s.close()
    s2.close()


def test_one_pixel_and_tiny_scene(mirt, oracle):
    sc = scenes.random_soup(3, 4)
    oracle.load(sc)
    s = mirt.Scene(sc)
    for (w, h) in [(1, 1), (3, 2), (9, 1)]:
        assert (s.render(w, h, spp=2, depth=2) == oracle.render(w, h, spp=2, depth=2)).all()
        a = s.primary_hits(w, h)
        b = oracle.primary_hits(w, h)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    s.close()


def test_cpp_host_mirror_app_renders_the_same_film(mirt, tmp_path):
    """apps/render.cpp is written against the reference's host API (PathTracer, factories, Generator::room) via
    include/raytracer_amd/pathtracer.hpp; its BMP must equal the film the ctypes path renders for the same scene."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    app = os.path.join(root, "raytracer_amd", "render_app")
    assert os.path.exists(app), "build it with make -C raytracer_amd/csrc all"
    out = str(tmp_path / "app.bmp")
    w, h, spp, depth = 64, 36, 2, 5
    r = subprocess.run([app, out, "--w", str(w), "--h", str(h), "--spp", str(spp), "--depth", str(depth)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "Image saved" in r.stdout, r.stdout + r.stderr
    rp, rn, wall = scenes.room_triangles(4, 3, 4)
    wall_mat = {0: 0, 1: 0, 2: 1, 3: 2, 4: 0, 5: 0}
    pos = np.concatenate([rp, np.array([[-.8, 0, .5, .8, 0, .5, 0, 1.6, .2], [-.5, 2.999, -.5, .5, 2.999, -.5, .5, 2.999, .5]], np.float32)])
    nrm = np.concatenate([rn, np.array([[0, .2, -1] * 3, [0, -1, 0] * 3], np.float32)])
    mat = np.concatenate([[wall_mat[int(k)] for k in wall], [3, 4]]).astype(np.uint32)
    sd = scenes.SceneData("app", pos, nrm, mat, np.array([scenes.WHITE, scenes.RED, scenes.GREEN, scenes.MIRROR, scenes.emission(4, 4, 4)]),
                          np.array([scenes.area_light((12, 12, 12), (0, 2.99, 0), (0, -1, 0), (1, 0, 0), (0, 0, 1)),
                                    scenes.point_light((1.5, 1.5, 1.5), (1.5, 2.5, -1.5))]),
                          scenes.Camera((0, 1.5, -1.95), (0, 1.2, 0), hfov=70, vfov=float(np.float32(70) * np.float32(h) / np.float32(w)))).finalize()
    s = mirt.Scene(sd)
    img = s.render(w, h, spp=spp, batch=spp + 1, depth=depth)
    ref_bmp = str(tmp_path / "py.bmp")
    mirt.save_bmp(ref_bmp, img)
    assert open(out, "rb").read() == open(ref_bmp, "rb").read()
    s.close()


def test_culling_zero_contribution_shadow_rays_keeps_the_film(mirt):
    """mirt_options.flags bit 4: a shadow ray whose contribution is exactly zero (light behind the surface) is not traced.
    x + 0 = x, so the accumulators must be bit-identical; only the shadow-ray count drops."""
    import torch
    sc = scenes.conference_scene(20000)
    w, h = 320, 180
    out = []
    for cull in (False, True):
        s = mirt.Scene(sc)
        acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
        st = s.render_accum(w, h, acc.data_ptr(), spp=8, depth=6, rng_mode=1, streams=2, cull_zero=cull)
        out.append((acc.cpu(), st["closest_rays"], st["shadow_rays"]))
        s.close()
    assert torch.equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1]
    assert out[1][2] < out[0][2]
    print("shadow rays", out[0][2], "->", out[1][2])


def test_scene_used_from_another_host_thread(mirt):
    """The library context (streams, the traversal stack's overflow region) is per host thread, a scene is not: a scene
    created on one thread renders from another thread's context on the same device - that context grows its own overflow
    region to what the scene's trees need (the kernels index that region unchecked)."""
    import threading

    sc = scenes.conference_scene(20000)
    s = mirt.Scene(sc)
    want = s.primary_hits(96, 64, traversal=0)
    got, err = [], []

    def other():
        try:
            mirt.init(0)                                  # this thread's own context, fresh: no overflow region yet
            got.append(s.primary_hits(96, 64, traversal=0))
            got.append(s.render(48, 32, spp=2, depth=3))
        except Exception as e:                            # noqa: BLE001
            err.append(e)

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert not err, err
    assert (got[0][0] == want[0]).all() and (got[0][1] == want[1]).all()
    assert (got[1] == s.render(48, 32, spp=2, depth=3)).all()
    s.close()
