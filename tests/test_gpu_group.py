"""mirt_group_* (include/mirt.h): one frame over several members from ONE host process -- the C-ABI route by which the C++
mirror of the reference's host API reaches N GPUs.  On the one-GPU box the members are logical partitions of GPU 0 (summed
by the in-order kernel); the RCCL binding is exercised with a single member, where ncclReduce is a no-op."""
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_logical_partitions_equal_the_single_context_frame(mirt):
    sc = scenes.cornell_scene(2, "mixed")
    w, h = 101, 67
    kw = dict(spp=6, depth=5, batch=7, tol=0.0)
    single = mirt.Scene(sc)
    want_ref_rng = single.render(w, h, **kw)                          # reference RNG: what the tile partition must reproduce
    want_counter = single.render(w, h, rng_mode=1, seed=9, **kw)
    counts = (single.last["closest_rays"], single.last["shadow_rays"], single.last["paths"])
    single.close()
    for n in (1, 2, 3):
        g = mirt.Group([0] * n)
        assert mirt.load().mirt_group_size(g.h) == n
        g.set_scene(sc)
        assert (g.render(w, h, partition="tile", **kw) == want_ref_rng).all()          # disjoint pixels: bit-identical
        img = g.render(w, h, partition="spp", rng_mode=1, seed=9, **kw)
        assert (img == want_counter).mean() >= 0.999                                   # same samples, re-associated sums
        assert (g.last["closest_rays"], g.last["shadow_rays"], g.last["paths"]) == counts
        g.close()
    # share pool through a group: every member reduces `shares` to a divisor of its own sample count (3 members, 6 spp: 2
    # samples each; tile partition: 6) - the films equal those of the same partitions rendered with sample streams
    for n, part in ((3, "spp"), (2, "tile")):
        g = mirt.Group([0] * n)
        g.set_scene(sc)
        pool = g.render(w, h, partition=part, rng_mode=1, seed=9, streams=1, shares=6, **kw)
        want = g.render(w, h, partition=part, rng_mode=1, seed=9, streams=2 if part == "spp" else 6, **kw)
        assert (pool == want).all()
        assert (g.last["closest_rays"], g.last["shadow_rays"], g.last["paths"]) == counts
        g.close()
    g = mirt.Group([0, 0])
    with pytest.raises(mirt.MirtError) as e:
        g.render(8, 8)
    assert "SCENE NOT SET" in e.value.msg
    g.set_scene(sc)
    with pytest.raises(mirt.MirtError):
        g.render(8, 8, partition="spp", spp=1)                        # fewer samples than members
    with pytest.raises(mirt.MirtError):
        g.render(8, 8, tile_rank=1, tile_world=2)                     # the group partitions the frame itself
    g.close()
    with pytest.raises(mirt.MirtError):
        mirt.Group([99])


def test_render_app_with_logical_partitions(tmp_path):
    """apps/render.cpp (PathTracer::devices) as N partitions of GPU 0 writes the BMP of the single-partition run."""
    app = os.path.join(ROOT, "raytracer_amd", "render_app")
    outs = []
    for extra in ([], ["--parts", "2", "--partition", "tile"], ["--parts", "4", "--partition", "tile"]):
        out = str(tmp_path / f"p{len(outs)}.bmp")
        r = subprocess.run([app, out, "--w", "96", "--h", "54", "--spp", "4", "--depth", "5"] + extra, capture_output=True,
                           text=True, timeout=180)
        assert r.returncode == 0 and "Image saved" in r.stdout, r.stdout + r.stderr
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] == outs[2]
    spp = []
    for extra in (["--rng", "1"], ["--rng", "1", "--parts", "2", "--partition", "spp"]):
        out = str(tmp_path / f"s{len(spp)}.bmp")
        r = subprocess.run([app, out, "--w", "96", "--h", "54", "--spp", "4", "--depth", "5"] + extra, capture_output=True,
                           text=True, timeout=180)
        assert r.returncode == 0, r.stdout + r.stderr
        spp.append(np.frombuffer(open(out, "rb").read(), np.uint8))
    assert (spp[0] == spp[1]).mean() >= 0.999


def test_rccl_binding_single_member(tmp_path):
    """The RCCL path (dlopen of librccl, ncclCommInitAll, ncclReduce on the member's stream) with one member on GPU 0: the
    reduce is a no-op, the frame must equal the plain one.  Runs in a child process (the switch is an environment variable)."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from raytracer_amd import capi, scenes\n"
        "capi.init(0)\n"
        "sc = scenes.cornell_scene(1, 'diffuse')\n"
        "s = capi.Scene(sc); a = s.render(64, 40, spp=3, depth=4); s.close()\n"
        "g = capi.Group([0]); g.set_scene(sc); b = g.render(64, 40, partition='tile', spp=3, depth=4); g.close()\n"
        "assert (a == b).all(); print('rccl ok')\n" % ROOT)
    env = dict(os.environ, MIRT_GROUP_FORCE_RCCL="1")
    r = subprocess.run(["python3", "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout + r.stderr


def test_particle_app_frame_loop(tmp_path):
    """apps/app.cpp = the reference's apps/app.cu (particle physics -> spheres + floor -> PathTracer -> BMP per frame) on a
    reduced configuration: the films do not depend on who builds the traversal's own tree (GPU PLOC by default, host SAH),
    frames differ (the particles move) and are lit."""
    app = os.path.join(ROOT, "raytracer_amd", "particle_app")
    assert os.path.exists(app), "build it with make -C raytracer_amd/csrc all"
    sets = []
    for extra in ([], ["--host-build"]):
        prefix = str(tmp_path / f"b{len(sets)}_")
        r = subprocess.run([app, "--frames", "3", "--particles", "30000", "--w", "200", "--h", "150", "--spp", "8", "--prefix", prefix]
                           + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "frames/s" in r.stdout, r.stdout + r.stderr
        sets.append([open(prefix + f"{i}.bmp", "rb").read() for i in range(3)])
    assert sets[0] == sets[1]
    a, b = (np.frombuffer(x, np.uint8)[54:].astype(int) for x in sets[0][:2])
    assert (a != b).mean() > 0.01 and a.mean() > 2
