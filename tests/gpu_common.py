"""Scenes and ray sets shared by the GPU parity tests (tests/test_gpu_*.py)."""
import numpy as np

from raytracer_amd import scenes

SCENES = {
    "cornell_diffuse": lambda: scenes.cornell_scene(2, "diffuse"),
    "cornell_mixed": lambda: scenes.cornell_scene(2, "mixed"),
    "blob64": lambda: scenes.blob_scene(64, 32),
    "soup500": lambda: scenes.random_soup(500, 1),
    "conf20k": lambda: scenes.conference_scene(20000),
    "adaptive": lambda: scenes.cornell_scene(1, "diffuse"),
}
SPHERE_SCENES = {"sphere_soup": lambda: scenes.sphere_soup(300, 60, 3), "known_answer": scenes.known_answer_scene}


def random_rays(sc, n, seed, unit):
    rng = np.random.default_rng(seed)
    lo = sc.pos.reshape(-1, 3).min(0) - 0.5
    hi = sc.pos.reshape(-1, 3).max(0) + 0.5
    r = np.zeros((n, 8), np.float32)
    r[:, :3] = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    if unit:
        d /= np.linalg.norm(d, axis=1, keepdims=True)
    r[:, 3:6] = d
    r[:, 6] = 1e-5
    r[:, 7] = np.inf
    r[: n // 8, 7] = rng.uniform(0.1, 3.0, n // 8)                  # finite max_t
    r[n // 8: n // 4, 3] = 0.0                                       # zero direction component: 0 * inf slabs
    r[n // 4: n // 4 + 60, 3:6] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 60)] * rng.choice([-1, 1], (60, 1))
    return r
