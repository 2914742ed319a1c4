#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: Mrays/s (all bounces) + ms/frame on conference* 1080p 64 spp depth 8.

A "step" is one whole frame of the hot path: wavefront path tracing of every (pixel, sample) of the frame with the scene
(BVH, triangles, materials, lights) already resident in HBM, ending with the float accumulation buffer summed onto rank 0
and tonemapped into host memory.  With N > 1 the frame is split over the ranks by spp slices (default; counter RNG, so every
(pixel, sample) is independent) or by interleaved 8x8 tiles (`--partition tile`), one process per GPU, `torch.distributed`
over RCCL; the only collective is the sum-reduce of the accumulation buffer.

    python bench.py [--gpus N --steps K --warmup W]

N > 1: the driver launches this file under `torch.distributed.run` (WORLD_SIZE set: every rank checks WORLD_SIZE == --gpus).
Started plainly with `--gpus N` and no WORLD_SIZE, the process launches those N ranks itself -- as children, BEFORE it makes
any GPU call (it never imports torch) -- waits for them and relays rank 0's JSON line.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (the closest-hit wide-BVH traversal) against HBM
peak: algorithmic bytes = rays x (160 B of wavefront state + node visits x 96 B + triangle tests x 48 B),
time = that kernel's launches measured with HIP events on its own stream inside the timed steps; `roofline.kernels` prices
all three hot kernels, `roofline.frame` adds up the measured HBM traffic of the whole frame, `roofline.sq` quotes the issue /
lane / wait fractions of the traversal kernels from the committed counter passes (profiles/r3d_pmc_sq.json).
`cpu_baseline` times the reference's own CPU path (oracle/_ref, built from /root/reference in the build container) -- or the
C oracle when that library is absent -- on a bounded crop of the same workload, on the box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
B_STATE_CLOSEST = 160          # ray r/w 64 + hit r/w 32 + path state r/w 64
S_NODE = {0: 96, 1: 128, 2: 32}  # bytes of one node: quantised 8-wide (traversal 0), 4-wide (1), reference BVH2 (2)
S_TRI = 48                     # bytes of one packed triangle (p1, e1, e2)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="conference*")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--rng", type=int, default=1, help="0 reference xorshift stream, 1 counter-based (default)")
    ap.add_argument("--traversal", type=int, default=0)
    ap.add_argument("--partition", default="spp", choices=["tile", "spp"],
                    help="N > 1: spp slices (default: the wavefront loop runs spp/N times fewer, full-size iterations; 0.83-0.97 "
                         "per-rank efficiency measured with tools/part_probe.py) or interleaved 8x8 tiles (bit-identical to one "
                         "GPU, but every iteration shrinks to 1/N of the rays: 0.23 at N = 8)")
    ap.add_argument("--streams", type=int, default=0,
                    help="sample streams per pixel in flight (mirt_options.streams): same samples, fewer and larger wavefront "
                         "iterations; 1 = one path per pixel at a time, the reference's order of summation; 0 (default) = with "
                         "the share pool 8 when a rank renders more than 32 samples per pixel, else 4 (tools/pool_streams.sh); "
                         "without it 8 from 16 samples per pixel on (tools/streams64.sh, tools/share_sweep.sh)")
    ap.add_argument("--shares", type=int, default=0,
                    help="share pool (mirt_options.shares): the rank's samples of a pixel are cut into this many shares (planes), "
                         "and the --streams path slots per pixel take (share, pixel) items of their 8x8 tile dynamically, so a "
                         "tile's slots stay busy until the tile runs out of work; the image is exactly that of --streams W.  "
                         "0 (default) = one share per sample when the rank renders at most 64 samples per pixel, off above that; "
                         "1 = off (tools/pool_ab.sh: 122.9 -> 119.6 ms at 64 spp, 19.6 -> 18.6 ms for the 8-spp share)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL, one GPU per rank (what the driver runs); gloo = rehearsal of the N > 1 path on a box "
                         "with fewer GPUs than ranks: ranks share the GPUs and the buffers are summed through host memory")
    ap.add_argument("--builder", default=None, choices=["host", "device", "ploc"],
                    help="who builds the ordered traversal's own tree (default: the library's, host binned SAH)")
    ap.add_argument("--cull-zero", action="store_true",
                    help="do not trace shadow rays whose contribution is exactly zero (same film, fewer rays; off by default: "
                         "the headline counts the rays the reference's control flow traces)")
    ap.add_argument("--no-film", action="store_true",
                    help="leave the film hand-over (tonemap, PCIe, widening) out of the step: what a rank other than 0 of an N-rank frame "
                         "runs; for pricing per-rank shares on one GPU (DESIGN.md 6), not a bench line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ N > 1 launch
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, script, argv):
    """`--gpus N` without WORLD_SIZE: start the N ranks as CHILD processes (torch.distributed.run, one rank per GPU,
    rendezvous on 127.0.0.1) from this process, which has made no GPU call and makes none; relay their output (rank 0 prints
    the JSON line) and return their exit code.  Never a re-exec of a process that has touched the GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for ln in proc.stdout:                                  # rank 0's JSON line to stdout, anything else a rank printed (gloo's
        dst = sys.stdout if ln.startswith("{") else sys.stderr      # connection notices) to stderr: stdout stays ONE JSON line
        dst.write(ln)
        dst.flush()
    return proc.wait()


# ------------------------------------------------------------------------------------------------ the renderer behind a step
class GpuBackend:
    """libmirt on this rank's GPU: the only renderer bench.py has (no CPU path).  tests/bench_stub_main.py substitutes a
    stand-in with this interface to exercise the launch / partition / reduce plumbing on a box without GPUs."""

    def __init__(self, args, rank, local_rank, world):
        import torch

        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
        n_dev = torch.cuda.device_count()
        if args.backend == "gloo":
            local_rank %= n_dev
        elif local_rank >= n_dev:
            raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU of its own ({n_dev} visible); "
                             "use --backend gloo to rehearse more ranks than GPUs")
        torch.cuda.set_device(local_rank)
        self.torch = torch
        self.device = f"cuda:{local_rank}"
        self.local_rank = local_rank
        self.args = args

    def init_process_group(self, dist):
        if self.args.backend == "nccl":
            dist.init_process_group("nccl", device_id=self.torch.device("cuda", self.local_rank))
        else:
            dist.init_process_group("gloo")

    def setup(self):
        from raytracer_amd import capi, scenes

        args = self.args
        self.capi = capi
        capi.init(self.local_rank)
        sd, self.data_kind = scenes.workload(args.workload)         # objects/<name>.obj when present, else the stand-in of SURVEY 8(d)
        sd.camera = sd.camera.for_aspect(args.width, args.height)
        self.sd = sd
        t0 = time.time()
        self.scene = capi.Scene(sd, builder=args.builder)
        self.build_s = time.time() - t0
        self.build_ms = {k: round(v, 2) for k, v in self.scene.build_times().items()}
        self.film = np.zeros((args.height, args.width, 3), np.int32)   # the caller's film buffer lives across frames (renderer.cuh:61)
        return self.torch.zeros((args.height, args.width, 4), dtype=self.torch.float32, device=self.device)

    def sync(self):
        self.torch.cuda.synchronize()

    def render_accum(self, accum, **opt):
        a = self.args
        kw = dict(spp=a.spp, spl=1, batch=a.spp + 1, tol=0.0, depth=a.depth)    # fixed spp: no adaptive stop
        kw.update(opt)
        return dict(self.scene.render_accum(a.width, a.height, accum.data_ptr(), **kw))

    def tonemap(self, accum):
        self.capi.tonemap(accum.data_ptr(), self.args.width, self.args.height, out=self.film)    # film in host memory

    def report(self, opt, stats, rays, ms_per_step, world, accum):
        return report(self, self.args, opt, stats, rays, ms_per_step, world, accum)

    def cpu_baseline(self):
        return cpu_baseline(self.sd, self.args)


def traffic_table(args, world):
    """HBM bytes per launch of the three hot kernels, from the committed PMC passes of this same command (rocprofv3 cannot
    wrap itself: tools/make_profiles.sh collects FETCH_SIZE / WRITE_SIZE in separate runs and writes profiles/traffic.json
    with the corrected figures).  ({}, None) when the file is absent or was collected on another configuration."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            d = json.load(fh)
        mine = {"workload": args.workload, "width": args.width, "height": args.height, "spp": args.spp, "depth": args.depth,
                "streams": args.streams, "shares": pick_shares(args.spp, args.shares, args.streams) if args.rng == 1 else 0,
                "traversal": args.traversal, "n_gpus": world}
        if d.get("configuration") != mine:
            return {}, None
        return {k: v["hbm_bytes_per_launch"] for k, v in d["kernels"].items()}, d.get("source")
    except (OSError, KeyError, ValueError, TypeError):
        return {}, None


def sq_table(args, world):
    """Issue / lane / wait fractions of the hot kernels from the committed SQ counter passes (tools/pmc_run.sh on the bench
    configuration -> profiles/r3d_pmc_sq.csv -> tools/pmc_sq_json.py -> profiles/r3d_pmc_sq.json).  None when absent or when
    it was collected for another workload / traversal."""
    try:
        with open(os.path.join(ROOT, "profiles", "r3d_pmc_sq.json")) as fh:
            d = json.load(fh)
        c = d.get("configuration", {})
        if (c.get("workload"), c.get("traversal"), c.get("n_gpus")) != (args.workload, args.traversal, world):
            return None
        return d
    except (OSError, ValueError, TypeError):
        return None


def cpu_baseline(scene_data, args):
    """The reference's CPU path on bounded samples of the same workload: all threads of the box's CPU share
    on a half-resolution frame (a quarter of the pixels, same spp and depth: ~10-20 s), and ONE thread on a 1/64 sample
    (quarter resolution, a quarter of the spp: a few seconds).  `kind` "reference" = oracle/_ref/libref.so, the reference's own
    headers compiled in the build container (the source never travels; the prebuilt library does); "port" = the C oracle,
    when that library is absent.  Rays are counted by the C oracle, whose control flow is bit-identical."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpulibs import CpuTracer, ref_available

    nproc = os.cpu_count() or 1
    threads = args.cpu_threads or min(16, nproc)          # the box's CPU share for one GPU is 16 cores; nproc shows the whole machine
    cam = scene_data.camera
    kind = "reference" if ref_available() else "port"

    def run(w, h, spp, nthreads):
        kw = dict(spp=spp, spl=1, batch=spp + 1, tol=0.0, depth=args.depth, nthreads=nthreads)
        orc = CpuTracer("oracle").load(scene_data)
        orc.set_camera(cam)
        orc.render(w, h, **kw)
        rays, secs = orc.last["closest"] + orc.last["shadow"], orc.last["seconds"]
        orc.close()
        if kind == "reference":
            ref = CpuTracer("ref").load(scene_data)
            ref.set_camera(cam)
            ref.render(w, h, **kw)
            secs = ref.last["seconds"]
            ref.close()
        return rays / secs / 1e6, f"{scene_data.name} {w}x{h}, {spp} spp, depth {args.depth}, {rays} rays in {secs:.2f} s"

    many, many_sample = run(args.width // 2, args.height // 2, args.spp, threads)
    one, one_sample = run(max(8, args.width // 4), max(8, args.height // 4), max(1, args.spp // 4), 1)
    return {"value": many, "unit": "Mrays/s", "cores": threads, "kind": kind, "sample": many_sample, "nproc": nproc,
            "single_thread": {"value": one, "unit": "Mrays/s", "cores": 1, "sample": one_sample}}


def logic_bytes(st, K):
    """Algorithmic bytes k_logic moves in the frame `st` describes, from what the frame actually did (SURVEY 8d's accounting,
    16-byte records): a slot that runs the slot function reads and writes its state and running sum (64 B) and appends itself
    to the live list (4 B); a slot a launch only looks at costs its state record (16 B); every extension ray is written
    (origin, direction, path mask: 48 B + 4 B of queue) and, one iteration later, its hit read (16 B); every shaded hit reads
    the ray, the mask, the primitive's normals and its material (32 + 16 + 48 + 48 B) -- shaded hits are counted as shadow
    rays / K, exact for scenes without delta materials (conference*) and a lower bound otherwise -- and parks the mask for the
    shadow fold (16 B written, 16 B read); every shadow ray is written (origin, direction, contribution: 48 B + 4 B of queue)
    and its result read (4 + 16 B); every finished sample reads and writes the pixel's accumulator (32 B)."""
    live, scanned = st["logic_live_slots"], st["logic_scanned_slots"]
    ext, sh, paths = st["closest_rays"], st["shadow_rays"], st["paths"]
    hits = sh / max(1, K)
    return (live * (64 + 4) + (scanned - live) * 16 + ext * (48 + 4 + 16) + hits * (32 + 16 + 48 + 48 + 32)
            + sh * (48 + 4 + 20) + paths * 32)


def pick_shares(spp, want, streams):
    """mirt_options.shares for a rank that renders `spp` samples per pixel: `want` rounded down to a divisor of spp; 0 (pool
    off) when that is not more than the sample streams.  want = 0, the default: one share per sample up to 64 samples per
    pixel, and no pool above that - a slot that renders 32 or more samples in sequence is balanced by their number, and the
    pool's item fetches in k_logic only cost (buddha* 256 spp: 304 ms without, 312 ms with; sportsCar* 4K 1024 spp: 6.52 s /
    6.63 s; conference* 64 spp: 118.8 / 116.0 ms; tools/pool_ab.sh, profiles/r3b_buddha_pool_probe.txt)."""
    if want == 1 or spp < 2 or (want == 0 and spp > 64):
        return 0
    cap = min(spp, want if want > 0 else 64)
    w = max(d for d in range(1, cap + 1) if spp % d == 0)
    return w if w > max(1, streams) else 0


def main(argv=None, backend_cls=GpuBackend, script=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks and does nothing else
        raise SystemExit(launch_ranks(args.gpus, script or os.path.abspath(__file__), argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}: the two must agree")

    be = backend_cls(args, rank, local_rank, world)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        be.init_process_group(dist)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: the process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
    import torch

    coll_dev = be.device if args.backend == "nccl" else "cpu"

    if args.streams == 0:
        spp_rank = args.spp // world if (world > 1 and args.partition == "spp") else args.spp
        if args.shares != 1 and args.rng == 1 and (args.shares > 1 or spp_rank <= 64):
            args.streams = 8 if spp_rank > 32 else 4
        else:
            args.streams = 8 if spp_rank >= 16 else 4

    # share pool: shares must divide the samples THIS rank renders (the last rank of an spp partition takes the remainder)
    from raytracer_amd import dist as rdist

    spp_mine = rdist.spp_slices(args.spp, world)[rank][1] if (world > 1 and args.partition == "spp" and args.spp >= world) else args.spp
    shares = pick_shares(spp_mine, args.shares, args.streams) if args.rng == 1 else 0

    accum = be.setup()
    w, h = args.width, args.height

    # the timed steps carry HIP events around the dominant kernel's launches only (every event pair costs a little)
    opt = dict(rng_mode=args.rng, traversal=args.traversal, profile_closest=True, streams=args.streams if args.rng == 1 else 0,
               cull_zero=args.cull_zero, shares=shares)

    if world > 1 and args.partition == "spp" and args.spp < world:
        raise SystemExit("spp partition needs at least one sample per rank: use --partition tile")

    def barrier():
        be.sync()
        if dist is not None:
            dist.barrier()
        be.sync()

    def step():
        # the one N > 1 code path (raytracer_amd/dist.py, rehearsed with gloo in tests/test_dist_cpu.py): partition options
        # for this rank -> libmirt -> ONE sum-reduce of the accumulation buffer onto rank 0
        st = rdist.render_partitioned(lambda acc, **part: be.render_accum(acc, **{**opt, **part}),
                                      accum, dist, rank, world, args.partition, args.spp, via_host=args.backend == "gloo",
                                      sync=be.sync)
        if rank == 0 and not args.no_film:
            be.tonemap(accum)
        return st

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0

    rays = sum(s["closest_rays"] + s["shadow_rays"] for s in stats)
    reduce_ms = sum(s.get("reduce_ms", 0.0) for s in stats) / max(1, args.steps)
    n_ranks = world
    tl = torch.tensor([elapsed, float(rays), reduce_ms], dtype=torch.float64, device=coll_dev)
    if dist is not None:
        tmax = tl.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tl, op=dist.ReduceOp.SUM)
        elapsed, rays, reduce_ms = float(tmax[0]), float(tl[1]), float(tmax[2])
        n_ranks = dist.get_world_size()                    # the ranks the collective library actually connected
    ms_per_step = 1e3 * elapsed / max(1, args.steps)

    if rank == 0:
        out = {
            "metric": f"Mrays/s (all bounces), {args.workload.rstrip('*')}.obj stand-in {args.width}x{args.height} {args.spp}spp"
                      if (args.workload, args.width, args.height, args.spp) != ("conference*", 1920, 1080, 64)
                      else "Mrays/s (all bounces), conference.obj stand-in 1080p 64spp",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": n_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_frame": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
        }
        out.update(be.report(opt, stats, rays, ms_per_step, world, accum))      # data, config, roofline
        if world > 1:
            out["reduce_ms"] = reduce_ms                   # the frame's one collective, slowest rank, per step
            out["config"]["backend"] = args.backend
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = be.cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def report(be, args, opt, stats, rays, ms_per_step, world, accum):
    """Rank 0, after the timed steps: `data`, `config` and `roofline` of the JSON line (two more untimed frames: a counting
    build of the kernels, and one with events around every launch)."""
    scene, sd = be.scene, be.sd
    w, h = args.width, args.height
    # traversal work per ray (counting build of the same kernels, one untimed reduced-spp frame)
    cnt = be.render_accum(accum, **{**opt, "spp": min(4, args.spp), "batch": args.spp + 1, "count": True, "sample_count": 0,
                                    "sample_begin": 0, "shares": 0})
    # one more untimed frame with events around EVERY launch: the per-kernel lines of the roofline (the timed steps carry
    # events around the closest-hit launches only)
    split = be.render_accum(accum, **{**opt, "profile": True, "profile_closest": False, "sample_count": 0, "sample_begin": 0})
    n_node = cnt["closest_node_visits"] / max(1, cnt["closest_rays"])
    n_tri = cnt["closest_tri_tests"] / max(1, cnt["closest_rays"])
    b_ray = B_STATE_CLOSEST + n_node * S_NODE[args.traversal] + n_tri * S_TRI
    launches = sum(s["closest_launches"] for s in stats)
    k_ms = sum(s["trace_closest_ms"] for s in stats)
    k_rays = sum(s["closest_rays"] for s in stats)
    avg_launch_ms = k_ms / max(1, launches)
    achieved = (k_rays * b_ray) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic, traffic_src = traffic_table(args, world)
    tname = "k_trace_fast" if args.traversal == 0 else "k_trace4" if args.traversal == 1 else "k_trace2"
    # shadow rays: 96 B of state (ray write + read, contribution write + read) + their own traversal counts
    K = len(sd.lights) * 1
    sn = cnt["shadow_node_visits"] / max(1, cnt["shadow_rays"])
    st_ = cnt["shadow_tri_tests"] / max(1, cnt["shadow_rays"])
    b_shadow = 96 + sn * S_NODE[args.traversal] + st_ * S_TRI
    # k_logic: bytes of the slots that actually ran (logic_bytes), not a full slot record for every slot of every launch
    lb = logic_bytes(split, K)
    live = split["logic_live_slots"]

    def line(name, units, unit_name, total_bytes, ms, n_launch):
        ach = total_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        t = traffic.get(name)
        return {"kernel": name, "launches": n_launch, "avg_launch_ms": ms / max(1, n_launch), f"{unit_name}_per_launch": units / max(1, n_launch),
                f"bytes_per_{unit_name}": total_bytes / max(1, units), "algorithmic_bytes_per_frame": total_bytes,
                "achieved": ach, "frac": ach / HBM_PEAK_GBS, "traffic": t,
                "hbm_frac": None if t is None or ms <= 0 else t / (ms * 1e-3 / max(1, n_launch)) / 1e9 / HBM_PEAK_GBS}

    per_kernel = [
        line(f"{tname}<closest>", split["closest_rays"], "ray", split["closest_rays"] * b_ray, split["trace_closest_ms"], split["closest_launches"]),
        line(f"{tname}<any>", split["shadow_rays"], "ray", split["shadow_rays"] * b_shadow, split["trace_shadow_ms"], split["shadow_launches"]),
        line("k_logic", live, "live_slot", lb, split["logic_ms"], split["logic_launches"]),
    ]
    per_kernel[2]["scanned_slots_per_launch"] = split["logic_scanned_slots"] / max(1, split["logic_launches"])
    # whole frame: the kernels overlap, so the per-kernel `frac`s are rates while each kernel runs, not shares of one HBM;
    # what one HBM has to carry per frame is the sum below, over the frame time
    frame_alg = sum(k["algorithmic_bytes_per_frame"] for k in per_kernel)
    frame = {"algorithmic_bytes": frame_alg, "algorithmic_frac": frame_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "hbm_bytes": None, "hbm_frac": None, "source": traffic_src}
    if traffic and all(k["traffic"] is not None for k in per_kernel):
        hb = sum(k["traffic"] * k["launches"] for k in per_kernel)
        frame["hbm_bytes"] = hb
        frame["hbm_frac"] = hb / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS
    return {
        "data": "real (objects/%s.obj)" % args.workload.rstrip("*") if be.data_kind == "real"
                else "synthetic (procedural stand-in scene: the reference's .obj meshes are absent)",
        "config": {"workload": f"{args.workload} {sd.n} tris, {w}x{h}, {args.spp} spp, depth {args.depth}, "
                               f"{len(sd.lights)} light{'s' if len(sd.lights) != 1 else ''}, rng {'counter' if opt['rng_mode'] else 'xorshift'}, "
                               f"traversal {args.traversal}, {opt['streams']} sample streams"
                               + (f", pool of {opt['shares']} shares" if opt.get("shares") else "")
                               + (", WITHOUT the film hand-over (--no-film: a share, not a frame)" if args.no_film else ""),
                   "partition": f"{args.partition} x{world}", "rays_per_frame": rays / max(1, args.steps),
                   "reference_walk_rays_per_frame": sum(s["exact_fallback_rays"] for s in stats) / max(1, args.steps),
                   "scene_build_s": be.build_s, "builder": args.builder or "host", "build_ms": be.build_ms},
        # the dominant kernel (closest-hit traversal) measured inside the timed steps; `kernels`: all three hot kernels in
        # one extra frame with events around every launch.  `frac` = algorithmic bytes / time / HBM peak (the contract's
        # figure); `hbm_frac` = measured HBM traffic (PMC) / time / HBM peak: what the kernel really asks of HBM - the
        # traversal kernels are bound by dependent-fetch latency under divergence, not by HBM (`sq`)
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic.get(f"{tname}<closest>"),
                     "kernel": f"{tname}<closest>", "avg_launch_ms": avg_launch_ms, "launches": launches,
                     "bytes_per_ray": b_ray, "node_visits_per_ray": n_node, "tri_tests_per_ray": n_tri,
                     "kernel_rays_per_s": k_rays / (k_ms * 1e-3) if k_ms > 0 else 0.0,
                     "closest_kernel_ms_timed_steps": k_ms,
                     "hbm_frac": None if traffic.get(f"{tname}<closest>") is None or avg_launch_ms <= 0
                     else traffic[f"{tname}<closest>"] / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "kernels": per_kernel, "frame": frame, "sq": sq_table(args, world)},
    }


if __name__ == "__main__":
    main()
