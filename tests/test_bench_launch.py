"""`bench.py --gpus N` without a launcher around it: the process starts N ranks itself (children, before any GPU call),
every rank checks WORLD_SIZE against --gpus, and rank 0 prints n_gpus from the process group.  Runs here with a stand-in
renderer (tests/bench_stub_main.py) and gloo: argument plumbing, partition options, the one reduce -- no GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "bench_stub_main.py")


def _run(argv, env_extra=None, expect_ok=True):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, STUB] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    if expect_ok:
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout                    # ONE JSON line, from rank 0
        return json.loads(lines[0])
    return p


@pytest.mark.parametrize("partition", ["spp", "tile"])
def test_gpus_2_launches_two_ranks(partition):
    d = _run(["--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--width", "64", "--height", "40", "--spp", "9",
              "--partition", partition])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["partition"] == f"{partition} x2" and d["config"]["backend"] == "gloo"
    assert d["config"]["divisor_min_max"] == [9.0, 9.0]      # both ranks' shares arrived on rank 0: every pixel has all 9 samples
    assert d["config"]["partition_options_rank0"] == ({"sample_begin": 0, "sample_count": 4} if partition == "spp"
                                                       else {"tile_rank": 0, "tile_world": 2})
    assert d["reduce_ms"] >= 0 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["streams"] == 4                       # 9 spp over 2 ranks: fewer than 16 per rank
    rays_per_frame = 64 * 40 * 9 * 3
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 / rays_per_frame - 1) < 1e-6     # whole-job rays over the slowest rank's time


def test_gpus_1_runs_in_process():
    d = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--width", "32", "--height", "16", "--spp", "4", "--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and "reduce_ms" not in d and d["config"]["divisor_min_max"] == [4.0, 4.0]


def test_world_size_must_equal_gpus():
    p = _run(["--gpus", "2", "--backend", "gloo"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, expect_ok=False)
    assert p.returncode != 0 and "must agree" in p.stderr


def test_share_pool_policy_of_the_bench():
    """bench.pick_shares: one share per sample up to 64 samples per pixel and rank (a divisor of the rank's sample count),
    off above that, off when it would not exceed the sample streams, explicit requests rounded down to a divisor."""
    import bench

    assert bench.pick_shares(64, 0, 8) == 64 and bench.pick_shares(8, 0, 4) == 8 and bench.pick_shares(32, 0, 4) == 32
    assert bench.pick_shares(256, 0, 8) == 0 and bench.pick_shares(1024, 0, 8) == 0          # balanced by their sample count
    assert bench.pick_shares(8, 0, 8) == 0 and bench.pick_shares(64, 1, 8) == 0 and bench.pick_shares(1, 0, 1) == 0
    assert bench.pick_shares(100, 64, 8) == 50 and bench.pick_shares(256, 64, 8) == 64 and bench.pick_shares(13, 0, 4) == 13
    for spp in range(2, 70):
        w = bench.pick_shares(spp, 0, 4)
        assert w == 0 or (spp % w == 0 and 4 < w <= 64)
