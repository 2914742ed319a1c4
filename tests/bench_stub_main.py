"""bench.py's launch / partition / reduce plumbing with a stand-in renderer, for boxes without a GPU
(tests/test_bench_launch.py).  Started as `python tests/bench_stub_main.py --gpus 2 --backend gloo ...` it goes through
bench.main() unchanged -- the parent launches the ranks with torch.distributed.run, every rank checks WORLD_SIZE against
--gpus, renders "its partition", joins the one sum-reduce, rank 0 prints the JSON line -- with bench.GpuBackend replaced by
the class below.  Test infrastructure: nothing in bench.py or the package refers to this file."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


class StubBackend:
    """Same interface as bench.GpuBackend; the "render" writes what the partition options say this rank owns."""

    def __init__(self, args, rank, local_rank, world):
        import torch

        self.torch, self.args, self.rank, self.world = torch, args, rank, world
        self.device = "cpu"
        self.seen = []

    def init_process_group(self, dist):
        assert self.args.backend == "gloo", "the stub has no GPU: --backend gloo"
        dist.init_process_group("gloo")

    def setup(self):
        return self.torch.zeros((self.args.height, self.args.width, 4), dtype=self.torch.float32)

    def sync(self):
        pass

    def render_accum(self, accum, **opt):
        from raytracer_amd import dist as rdist

        a = self.args
        self.seen.append({k: opt[k] for k in ("tile_rank", "tile_world", "sample_begin", "sample_count") if k in opt})
        if "tile_world" in opt:
            own = self.torch.from_numpy(rdist.tile_owner(a.width, a.height, opt["tile_world"]) == opt["tile_rank"])
            n = a.spp
        else:
            own = self.torch.ones((a.height, a.width), dtype=self.torch.bool)
            n = opt.get("sample_count") or a.spp
        accum.zero_()
        accum[..., 3] = own * float(n)
        accum[..., 0] = own * float(n) * 0.5
        pix = int(own.sum())
        return {"closest_rays": pix * n, "shadow_rays": 2 * pix * n, "paths": pix * n, "exact_fallback_rays": 0}

    def tonemap(self, accum):
        self.divisor = (float(accum[..., 3].min()), float(accum[..., 3].max()))

    def report(self, opt, stats, rays, ms_per_step, world, accum):
        return {"data": "stub (launch rehearsal, no renderer)",
                "config": {"workload": "stub", "partition": f"{self.args.partition} x{world}", "divisor_min_max": self.divisor,
                           "streams": opt["streams"], "partition_options_rank0": self.seen[-1]},
                "roofline": None}

    def cpu_baseline(self):
        return None


if __name__ == "__main__":
    bench.main(backend_cls=StubBackend, script=os.path.abspath(__file__))
