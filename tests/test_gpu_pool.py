"""The share pool (mirt_options.shares, no reference counterpart): W shares per pixel, each summed in sample order into its own
plane, the planes added in share order - by construction the image of `streams = W` - while only `streams` path slots per
pixel exist and take (share, pixel) items dynamically.  Bar: the float accumulation buffer is BIT-IDENTICAL to the render
with streams = W, the rays traced are the same, and the result does not depend on scheduling (two runs: same bits)."""
import numpy as np
import pytest

from raytracer_amd import scenes

pytestmark = pytest.mark.gpu


def _accum(torch, s, w, h, **kw):
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    st = dict(s.render_accum(w, h, acc.data_ptr(), **kw))
    return acc, st


RAYS = ("closest_rays", "shadow_rays", "paths")


@pytest.mark.parametrize("kind", ["mixed", "diffuse"])
def test_share_pool_equals_sample_streams_bit_for_bit(mirt, kind):
    torch = pytest.importorskip("torch")
    s = mirt.Scene(scenes.cornell_scene(2, kind))              # mixed: glass, mirror, microfacet (the deferred kernel) too
    w, h, spp = 203, 117, 12                                   # ragged 8x8 tiles: pixel slots that cover no pixel
    kw = dict(spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=5, rng_mode=1, seed=11)
    # home groups: the default, one tile, a size that leaves a ragged last group, and one group for the whole image
    for shares, streams, group in ((12, 1, 0), (12, 4, 64), (12, 5, 448), (6, 2, 1 << 20), (4, 1, 0), (3, 2, 64), (2, 1, 448),
                                   (12, 4, 1 << 20), (12, 1, 448)):
        ref, st0 = _accum(torch, s, w, h, streams=shares, **kw)
        acc, st = _accum(torch, s, w, h, streams=streams, shares=shares, pool_group=group, **kw)
        assert [st[k] for k in RAYS] == [st0[k] for k in RAYS], (shares, streams, group)
        assert (acc[..., 3] == spp).all()
        assert torch.equal(acc, ref), f"shares {shares} streams {streams} group {group}: {(acc != ref).any(dim=-1).sum().item()} pixels differ"
        again, _ = _accum(torch, s, w, h, streams=streams, shares=shares, pool_group=group, **kw)
        assert torch.equal(acc, again)
    # fewer iterations than the static assignment with the same number of slots
    _, st_static = _accum(torch, s, w, h, streams=4, **kw)
    _, st_pool = _accum(torch, s, w, h, streams=4, shares=12, **kw)
    assert st_pool["iterations"] <= st_static["iterations"]
    # shares <= streams: plain sample streams
    a, _ = _accum(torch, s, w, h, streams=4, shares=4, **kw)
    b, _ = _accum(torch, s, w, h, streams=4, **kw)
    assert torch.equal(a, b)
    for bad in (dict(shares=5), dict(shares=24), dict(shares=4, rng_mode=0), dict(shares=4, pool_group=100)):
        with pytest.raises(mirt.MirtError):
            _accum(torch, s, w, h, **{**kw, **bad})
    s.close()


def test_share_pool_composes_with_partitions_and_culling(mirt):
    torch = pytest.importorskip("torch")
    s = mirt.Scene(scenes.cornell_scene(2, "mixed"))
    w, h, spp = 150, 90, 16
    kw = dict(spp=spp, spl=2, batch=spp + 1, tol=0.0, depth=4, rng_mode=1, seed=3)
    full, st_full = _accum(torch, s, w, h, streams=2, shares=8, **kw)
    ref, _ = _accum(torch, s, w, h, streams=8, **kw)
    assert torch.equal(full, ref)
    # tile partition: disjoint pixels, bit-identical sum
    total = torch.zeros_like(full)
    rays = [0, 0, 0]
    for r in range(3):
        part, st = _accum(torch, s, w, h, streams=2, shares=8, tile_rank=r, tile_world=3, **kw)
        total += part
        rays = [a + st[k] for a, k in zip(rays, RAYS)]
    assert torch.equal(total, full)
    assert rays == [st_full[k] for k in RAYS]
    # spp slices with a pool each = the same slices with sample streams
    for begin, count in ((0, 8), (8, 8)):
        a, _ = _accum(torch, s, w, h, streams=2, shares=4, sample_begin=begin, sample_count=count, **kw)
        b, _ = _accum(torch, s, w, h, streams=4, sample_begin=begin, sample_count=count, **kw)
        assert torch.equal(a, b)
    # progressive passes through the pool
    prog = torch.zeros_like(full)
    for p in range(2):
        s.render_accum(w, h, prog.data_ptr(), sample_begin=8 * p, sample_count=8, accumulate=p > 0, streams=2, shares=8, **kw)
    want = torch.zeros_like(full)
    for p in range(2):
        s.render_accum(w, h, want.data_ptr(), sample_begin=8 * p, sample_count=8, accumulate=p > 0, streams=8, **kw)
    assert torch.equal(prog, want)
    # zero-contribution culling leaves the sums alone
    c, st_c = _accum(torch, s, w, h, streams=2, shares=8, cull_zero=True, **kw)
    assert torch.equal(c, full) and st_c["shadow_rays"] <= st_full["shadow_rays"]
    s.close()


def test_share_pool_on_a_tiny_image_with_mostly_empty_tiles(mirt):
    """5 x 3 pixels: one 8x8 tile of which 49 pixel slots cover nothing - most items the pool hands out are nobody's."""
    torch = pytest.importorskip("torch")
    s = mirt.Scene(scenes.cornell_scene(1, "diffuse"))
    w, h, spp = 5, 3, 16
    kw = dict(spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=4, rng_mode=1, seed=2)
    ref, st0 = _accum(torch, s, w, h, streams=16, **kw)
    for streams, group in ((1, 0), (2, 64), (5, 128)):
        acc, st = _accum(torch, s, w, h, streams=streams, shares=16, pool_group=group, **kw)
        assert torch.equal(acc, ref)
        assert [st[k] for k in RAYS] == [st0[k] for k in RAYS]
    s.close()


def test_share_pool_full_size_bench_share(mirt):
    """The 8-spp share of the bench frame (conference* 1080p): pool of 8 one-sample shares over 4 slots per pixel = the render
    with 8 sample streams, bit for bit, in fewer wavefront iterations than 4 static streams."""
    torch = pytest.importorskip("torch")
    sc = scenes.conference_scene()
    w, h, spp = 1920, 1080, 8
    sc.camera = sc.camera.for_aspect(w, h)
    s = mirt.Scene(sc)
    kw = dict(spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=8, rng_mode=1, seed=0)
    ref, st0 = _accum(torch, s, w, h, streams=8, **kw)
    acc, st = _accum(torch, s, w, h, streams=4, shares=8, **kw)
    assert torch.equal(acc, ref)
    assert [st[k] for k in RAYS] == [st0[k] for k in RAYS]
    _, st4 = _accum(torch, s, w, h, streams=4, **kw)
    assert st["iterations"] <= st4["iterations"]
    # the reference walk stays the exception (a ray record the ordered traversal misreads would send whole classes of rays
    # there: right results, a frame several times longer)
    for run in (st0, st, st4):
        assert run["exact_fallback_rays"] <= 1e-4 * (run["closest_rays"] + run["shadow_rays"]), run["exact_fallback_rays"]
    print(f"iterations: 4 static streams {st4['iterations']}, pool 4 x 8 {st['iterations']}, 8 static streams {st0['iterations']}; "
          f"render_ms {st4['render_ms']:.2f} / {st['render_ms']:.2f} / {st0['render_ms']:.2f}")
    s.close()
