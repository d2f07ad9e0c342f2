"""Developer run (not a test): the filter audit at scale.  For each scene, 400 000 rays -- camera rays, their
first and second bounces as the render kernel would shoot them (surface starts, unnormalised
directions), silhouette-grazing rays -- against every sphere, both sphere loops, both arithmetic modes:
prints accepted pairs, missed pairs (must be 0) and the smallest slack against the margin."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import craytracer_amd as ca  # noqa: E402
from campaign_two_loops import random_scene  # noqa: E402
from test_gpu_functions import _grazing_rays  # noqa: E402

G = {(True, True): 1.0e-5, (False, True): 2.5e-6, (True, False): 4.0e-5, (False, False): 3.25e-5}   # (strict, vector_loop) -> margin


def bounce_rays(gpu, rays, rng):
    """Rays leaving the surfaces the given rays hit: origin at the hit point, direction normal + random unit vector."""
    h = gpu.test_trace(rays, strict=True)
    ok = h["hit"] == 1
    p, n = h["p"][ok], h["n"][ok]
    u = rng.normal(size=p.shape)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    return np.concatenate([p, (n + u).astype(np.float32)], axis=1).astype(np.float32)


def audit(gpu, name, fs, cam, rng, n=100000):
    gpu.upload_scene(fs)
    uv = rng.uniform(0, 1, size=(n, 2)).astype(np.float32)
    cam_rays = gpu.test_camera_ray(cam, uv, seed=1, strict=True)["ray"]
    b1 = bounce_rays(gpu, cam_rays, rng)
    b2 = bounce_rays(gpu, b1, rng) if len(b1) else b1
    rays = [cam_rays, b1, b2]
    if fs.n_spheres > 2 and np.abs(fs.objects["g"][:, :3]).max() < 1e3:
        rays.append(_grazing_rays(rng, fs, n))
    rays = np.concatenate([r for r in rays if len(r)])
    total_missed = 0
    for strict in (True, False):
        for vector_loop in (False, True):
            a = gpu.filter_audit(rays, strict=strict, vector_loop=vector_loop)
            fin = a["slack"][np.isfinite(a["slack"])]
            g = G[(strict, vector_loop)]
            total_missed += int(a["missed"].sum())
            print(f"{name:28s} {len(rays):7d} rays {'strict' if strict else 'fast  '} {'vector' if vector_loop else 'matrix'}: "
                  f"{int(a['accepted'].sum()):9d} accepted, {int(a['missed'].sum())} missed, min slack "
                  f"{(fin.min() if len(fin) else float('nan')):.3e} of {g:.2e}", flush=True)
    return total_missed


def main():
    gpu = ca.CasterHip(0)
    rng = np.random.default_rng(77)
    cam = ca.default_camera(16 / 9)
    missed = 0
    missed += audit(gpu, "rtiow", ca.scene_rtiow(1), cam, rng)
    missed += audit(gpu, "perlin layout", ca.scene_perlin(1), cam, rng)
    missed += audit(gpu, "stress 4500", ca.scene_stress(4500, 2), cam, rng)
    missed += audit(gpu, "stress 10000", ca.scene_stress(10000, 2), cam, rng, n=40000)
    crng = np.random.default_rng(2024)
    for case in range(40):
        fs, centre, extent, nsph = random_scene(crng, case)
        eye = centre + crng.normal(size=3) * extent * float(crng.choice([0.3, 1.5, 4.0]))
        ccam = ca.calculate_cam_info(tuple(eye), tuple(centre), (0, 1, 0), 40.0, np.float32(16 / 9), 0.0, 1.0)
        if case % 4 == 0:
            missed += audit(gpu, f"campaign {case} n={nsph} ext={extent:.3g}", fs, ccam, rng, n=30000)
    print("missed pairs in total:", missed)
    return 1 if missed else 0


if __name__ == "__main__":
    sys.exit(main())
