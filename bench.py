#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: Mrays/s (all bounces) + ms/frame on conference* 1080p 64 spp depth 8.

A "step" is one whole frame of the hot path: wavefront path tracing of every (pixel, sample) of the frame with the scene
(BVH, triangles, materials, lights) already resident in HBM, ending with the float accumulation buffer summed onto rank 0
and tonemapped into host memory.  With N > 1 the frame is split over the ranks by spp slices (default; counter RNG, so every
(pixel, sample) is independent) or by interleaved 8x8 tiles (`--partition tile`), one process per GPU, `torch.distributed`
over RCCL; the only collective is the sum-reduce of the accumulation buffer.

    python bench.py [--gpus N --steps K --warmup W]       # N > 1: launched under torch.distributed.run by the driver

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (the closest-hit wide-BVH traversal) against HBM
peak: algorithmic bytes = rays x (160 B of wavefront state + node visits x 96 B + triangle tests x 48 B),
time = that kernel's launches measured with HIP events on its own stream inside the timed steps.  `cpu_baseline` times the
reference's own CPU path (oracle/_ref, built from /root/reference in the build container) -- or the C oracle when that
library is absent -- on a bounded crop of the same workload, on the box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
B_STATE_CLOSEST = 160          # ray r/w 64 + hit r/w 32 + path state r/w 64
S_NODE = {0: 96, 1: 128, 2: 32}  # bytes of one node: quantised 8-wide (traversal 0), 4-wide (1), reference BVH2 (2)
S_TRI = 48                     # bytes of one packed triangle (p1, e1, e2)


def parse():
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
                         "iterations; 1 = one path per pixel at a time, the reference's order of summation; 0 (default) = 8 when "
                         "a rank renders at least 16 samples per pixel, else 4 (measured: tools/streams64.sh, tools/share_sweep.sh)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL, one GPU per rank (what the driver runs); gloo = rehearsal of the N > 1 path on a box "
                         "with fewer GPUs than ranks: ranks share the GPUs and the buffers are summed through host memory")
    ap.add_argument("--builder", default=None, choices=["host", "device", "ploc"],
                    help="who builds the ordered traversal's own tree (default: the library's, host binned SAH)")
    ap.add_argument("--cull-zero", action="store_true",
                    help="do not trace shadow rays whose contribution is exactly zero (same film, fewer rays; off by default: "
                         "the headline counts the rays the reference's control flow traces)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args()


def traffic_table(args, world):
    """HBM bytes per launch of the three hot kernels, from the committed PMC passes of this same command (rocprofv3 cannot
    wrap itself: tools/make_profiles.sh collects FETCH_SIZE / WRITE_SIZE in separate runs and writes profiles/traffic.json
    with the corrected figures).  {} when the file is absent or was collected on another configuration than the one running."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            d = json.load(fh)
        mine = {"workload": args.workload, "width": args.width, "height": args.height, "spp": args.spp, "depth": args.depth,
                "streams": args.streams, "traversal": args.traversal, "n_gpus": world}
        if d.get("configuration") != mine:
            return {}
        return {k: v["hbm_bytes_per_launch"] for k, v in d["kernels"].items()}
    except (OSError, KeyError, ValueError, TypeError):
        return {}


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


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    coll_dev = f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"

    from raytracer_amd import capi, scenes

    if args.streams == 0:
        spp_rank = args.spp // world if (world > 1 and args.partition == "spp") else args.spp
        args.streams = 8 if spp_rank >= 16 else 4

    capi.init(local_rank)
    sd, data_kind = scenes.workload(args.workload)             # objects/<name>.obj when present, else the stand-in of SURVEY 8(d)
    sd.camera = sd.camera.for_aspect(args.width, args.height)
    t0 = time.time()
    scene = capi.Scene(sd, builder=args.builder)
    build_s = time.time() - t0
    build_ms = {k: round(v, 2) for k, v in scene.build_times().items()}
    w, h = args.width, args.height
    accum = torch.zeros((h, w, 4), dtype=torch.float32, device=f"cuda:{local_rank}")

    # the timed steps carry HIP events around the dominant kernel's launches only (every event pair costs a little)
    opt = dict(rng_mode=args.rng, traversal=args.traversal, profile_closest=True, streams=args.streams if args.rng == 1 else 0,
               cull_zero=args.cull_zero)
    from raytracer_amd import dist as rdist

    if world > 1 and args.partition == "spp" and args.spp < world:
        raise SystemExit("spp partition needs at least one sample per rank: use --partition tile")
    kw = dict(spp=args.spp, spl=1, batch=args.spp + 1, tol=0.0, depth=args.depth)   # fixed spp: no adaptive stop

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    film = np.zeros((h, w, 3), np.int32)       # the caller's film buffer lives across frames, as the reference's sample_buffer does

    def step():
        # the one N > 1 code path (raytracer_amd/dist.py, rehearsed with gloo in tests/test_dist_cpu.py): partition options
        # for this rank -> libmirt -> ONE sum-reduce of the accumulation buffer onto rank 0
        st = rdist.render_partitioned(lambda acc, **part: scene.render_accum(w, h, acc.data_ptr(), **kw, **{**opt, **part}),
                                      accum, dist, rank, world, args.partition, args.spp, via_host=args.backend == "gloo")
        if dist is not None:
            torch.cuda.synchronize()
        if rank == 0:
            capi.tonemap(accum.data_ptr(), w, h, out=film)          # film in host memory
        return st

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0

    rays = sum(s["closest_rays"] + s["shadow_rays"] for s in stats)
    tl = torch.tensor([elapsed, float(rays)], dtype=torch.float64, device=coll_dev)
    if dist is not None:
        tmax = tl.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tl, op=dist.ReduceOp.SUM)
        elapsed, rays = float(tmax[0]), float(tl[1])
    ms_per_step = 1e3 * elapsed / max(1, args.steps)

    if rank == 0:
        # traversal work per ray (counting build of the same kernels, one untimed reduced-spp frame)
        cnt = scene.render_accum(w, h, accum.data_ptr(), spp=min(4, args.spp), spl=1, batch=args.spp + 1, tol=0.0,
                                 depth=args.depth, **{**opt, "count": True, "sample_count": 0, "sample_begin": 0})
        # one more untimed frame with events around EVERY launch: the per-kernel lines of the roofline (the timed steps carry
        # events around the closest-hit launches only)
        split = scene.render_accum(w, h, accum.data_ptr(), **kw, **{**opt, "profile": True, "profile_closest": False,
                                                                  "sample_count": 0, "sample_begin": 0})
        n_node = cnt["closest_node_visits"] / max(1, cnt["closest_rays"])
        n_tri = cnt["closest_tri_tests"] / max(1, cnt["closest_rays"])
        b_ray = B_STATE_CLOSEST + n_node * S_NODE[args.traversal] + n_tri * S_TRI
        launches = sum(s["closest_launches"] for s in stats)
        k_ms = sum(s["trace_closest_ms"] for s in stats)
        k_rays = sum(s["closest_rays"] for s in stats)
        avg_launch_ms = k_ms / max(1, launches)
        achieved = (k_rays * b_ray) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = traffic_table(args, world)
        tname = "k_trace_fast" if args.traversal == 0 else "k_trace4" if args.traversal == 1 else "k_trace2"
        # shadow rays: 96 B of state (ray write + read, contribution write + read) + their own traversal counts;
        # k_logic per slot and iteration: state it reads (7 x 16 B + K x 20 B of shadow results + 48 B of normals + 48 B of
        # material) and writes (6 x 16 B + K x 48 B of shadow rays + queue and live-list entries)
        K = len(sd.lights) * 1
        sn = cnt["shadow_node_visits"] / max(1, cnt["shadow_rays"])
        st_ = cnt["shadow_tri_tests"] / max(1, cnt["shadow_rays"])
        b_shadow = 96 + sn * S_NODE[args.traversal] + st_ * S_TRI
        b_slot = (208 + 20 * K) + (96 + 48 * K) + (8 + 4 * K)
        slots_per_launch = (w + 7) // 8 * ((h + 7) // 8) * 64 * max(1, opt["streams"]) / max(1, round(split["logic_launches"] / max(1, split["iterations"])))

        def line(name, units, unit_name, b_unit, ms, n_launch):
            ach = units * b_unit / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            t = traffic.get(name)
            return {"kernel": name, "launches": n_launch, "avg_launch_ms": ms / max(1, n_launch), f"{unit_name}_per_launch": units / max(1, n_launch),
                    f"bytes_per_{unit_name}": b_unit, "achieved": ach, "frac": ach / HBM_PEAK_GBS, "traffic": t,
                    "hbm_frac": None if t is None or ms <= 0 else t / (ms * 1e-3 / max(1, n_launch)) / 1e9 / HBM_PEAK_GBS}

        per_kernel = [
            line(f"{tname}<closest>", split["closest_rays"], "ray", b_ray, split["trace_closest_ms"], split["closest_launches"]),
            line(f"{tname}<any>", split["shadow_rays"], "ray", b_shadow, split["trace_shadow_ms"], split["shadow_launches"]),
            line("k_logic", slots_per_launch * split["logic_launches"], "slot", b_slot, split["logic_ms"], split["logic_launches"]),
        ]
        out = {
            "metric": f"Mrays/s (all bounces), {args.workload.rstrip('*')}.obj stand-in {args.width}x{args.height} {args.spp}spp"
                      if (args.workload, args.width, args.height, args.spp) != ("conference*", 1920, 1080, 64)
                      else "Mrays/s (all bounces), conference.obj stand-in 1080p 64spp",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_frame": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "real (objects/%s.obj)" % args.workload.rstrip("*") if data_kind == "real"
                    else "synthetic (procedural stand-in scene: the reference's .obj meshes are absent)",
            "config": {"workload": f"{args.workload} {sd.n} tris, {w}x{h}, {args.spp} spp, depth {args.depth}, "
                                   f"{len(sd.lights)} light{'s' if len(sd.lights) != 1 else ''}, rng {'counter' if opt['rng_mode'] else 'xorshift'}, "
                                   f"traversal {args.traversal}, {opt['streams']} sample streams",
                       "partition": f"{args.partition} x{world}", "rays_per_frame": rays / max(1, args.steps),
                       "reference_walk_rays_per_frame": sum(s["exact_fallback_rays"] for s in stats) / max(1, args.steps),
                       "scene_build_s": build_s, "builder": args.builder or "host", "build_ms": build_ms},
            # the dominant kernel (closest-hit traversal) measured inside the timed steps; `kernels`: all three hot kernels in
            # one extra frame with events around every launch.  `frac` = algorithmic bytes / time / HBM peak (the contract's
            # figure); `hbm_frac` = measured HBM traffic (PMC) / time / HBM peak: what the kernel really asks of HBM - the
            # traversal kernels are bound by the vector-L1 / L2 request rate of their divergent gathers, not by HBM
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic.get(f"{tname}<closest>"),
                         "kernel": f"{tname}<closest>", "avg_launch_ms": avg_launch_ms, "launches": launches,
                         "bytes_per_ray": b_ray, "node_visits_per_ray": n_node, "tri_tests_per_ray": n_tri,
                         "kernel_rays_per_s": k_rays / (k_ms * 1e-3) if k_ms > 0 else 0.0,
                         "closest_kernel_ms_timed_steps": k_ms,
                         "hbm_frac": None if traffic.get(f"{tname}<closest>") is None or avg_launch_ms <= 0
                         else traffic[f"{tname}<closest>"] / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "kernels": per_kernel},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, args)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
