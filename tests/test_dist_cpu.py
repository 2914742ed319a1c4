"""The N > 1 composition rule (tile / spp partition + one sum-reduce of the accumulation buffer), rehearsed on CPU with
gloo, world_size 2.  The per-rank renderer here is the C oracle in counter-RNG mode (test infrastructure); on GPUs the same
partition options go to libmirt and the reduce runs over RCCL (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpulibs import CpuTracer
    from raytracer_amd import dist as rdist, scenes

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, spp, depth = 40, 24, 6, 3
    o = CpuTracer("oracle").load(scenes.cornell_scene(1, "diffuse"))
    o.set_rng(1, 11)

    def render_accum(buf, **opts):
        """Stub renderer with libmirt's partition semantics (the C oracle in counter-RNG mode; on GPUs this is
        capi.Scene.render_accum): writes this rank's share into `buf`, zeros elsewhere."""
        if "tile_rank" in opts:
            _, acc = o.render(w, h, spp=spp, batch=spp + 1, depth=depth, accum=True)
            acc = acc * (rdist.tile_owner(w, h, opts["tile_world"]) == opts["tile_rank"])[..., None]
        else:
            _, acc = o.render(w, h, spp=opts["sample_count"], batch=spp + 1, depth=depth, accum=True,
                              sample_begin=opts["sample_begin"])
        buf.copy_(torch.from_numpy(np.ascontiguousarray(acc)))
        return dict(o.last)

    t = torch.zeros((h, w, 4), dtype=torch.float32)
    rdist.render_partitioned(render_accum, t, dist, rank, world, kind, spp)      # the function bench.py calls
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["tile", "spp"])
def test_two_rank_partition_equals_one_rank(kind, tmp_path):
    from cpulibs import CpuTracer
    from raytracer_amd import scenes

    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(2, _free_port(), kind, out), nprocs=2, join=True)
    got = np.load(out)
    o = CpuTracer("oracle").load(scenes.cornell_scene(1, "diffuse"))
    o.set_rng(1, 11)
    _, want = o.render(40, 24, spp=6, batch=7, depth=3, accum=True)
    assert (got[..., 3] == 6).all()                                   # divisors add up to the full sample count
    if kind == "tile":
        assert (got.view(np.uint32) == want.view(np.uint32)).all()    # disjoint supports: bit-identical
    else:
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7)   # re-associated float sums


def test_partition_helpers():
    from raytracer_amd import dist as rdist

    own = rdist.tile_owner(20, 17, 3)
    assert own.shape == (17, 20) and set(np.unique(own)) == {0, 1, 2}
    assert (own[:8, :8] == 0).all() and (own[:8, 8:16] == 1).all() and (own[8:16, :8] == (3 % 3)).all()
    with pytest.raises(ValueError):
        rdist.spp_slices(2, 4)                                       # an empty slice would mean "all samples" to libmirt
    with pytest.raises(ValueError):
        rdist.partition_options("spp", 0, 4, 2)
    assert rdist.partition_options("spp", 1, 1, 64) == {}
    for spp, world in [(64, 8), (7, 3), (4, 4)]:
        sl = rdist.spp_slices(spp, world)
        assert sum(c for _, c in sl) == spp and sl[0][0] == 0
        assert all(sl[i][0] + sl[i][1] == sl[i + 1][0] for i in range(world - 1))
