"""Not collected by pytest (run by name on a GPU box): a randomised campaign for the conservative
sphere filter on the matrix cores.  Random scenes -- sphere counts 34 to 4000, extents from 1e-2 to
1e4 scene units, centres of mass up to 1e5 from the origin, radii over a factor of 25, random
cameras inside and outside the cloud -- rendered in both arithmetic modes with the filter on the
matrix cores and on the vector unit, and once more with every camera ray meeting every sphere
instead of its tile's list (lenses from a pinhole to half the cloud across, focus planes before, in
and behind the cloud); all frames of a case must agree bit for bit.
usage: python tests/campaign_two_loops.py [--noise] [--cases N]
--noise: every other diffuse or metal material of a scene carries the marble texture (the texture
instances of the kernels, whose noise lookups are evaluated for both of a lane's paths at once).
Every render ends with the waves' last paths gathered into one column block of the matrix loop,
with or without textures."""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import craytracer_amd as ca  # noqa: E402


def random_scene(rng, base):
    n = int(np.exp(rng.uniform(np.log(34), np.log(4000))))
    fs = ca.scene_stress(n, int(rng.integers(1, 1000)))
    objs = fs.objects.copy()
    small = objs["g"][:, 3] < 50
    k = int(small.sum())
    extent = float(np.exp(rng.uniform(np.log(1e-2), np.log(1e4))))
    centre = rng.normal(size=3) * float(np.exp(rng.uniform(np.log(1e-1), np.log(1e5))))
    r0 = extent * float(np.exp(rng.uniform(np.log(2e-3), np.log(5e-2))))
    objs["g"][small, 3] = (r0 * np.exp(rng.uniform(0, np.log(25.0), size=k))).astype(np.float32)
    objs["g"][small, :3] = (centre + rng.uniform(-extent, extent, size=(k, 3))).astype(np.float32)
    # ground and sky move with the cloud
    big = ~small
    objs["g"][big, :3] += centre.astype(np.float32)
    objs["g"][big, 3] *= np.float32(max(extent / 50.0, 1e-3))
    objs["g"][big, 1] -= np.float32(extent)                      # ground below the cloud
    mats, noise = fs.materials, None
    if NOISE:
        mats = mats.copy()
        marble = np.flatnonzero((mats["type"] == 0) | (mats["type"] == 1))[::2]
        mats["tex_type"][marble] = 4                                            # TTYPE_NOISE
        mats["color1"][marble, 0] = np.float32(rng.uniform(0.5, 8.0) / max(extent, 1e-6))   # stripes per scene unit
        noise = ca.make_noise(int(rng.integers(1, 1000)))
    return ca.FlatScene(objs, mats, [], noise), centre, extent, n


NOISE = "--noise" in sys.argv
CASES = int(sys.argv[sys.argv.index("--cases") + 1]) if "--cases" in sys.argv else 40


def main():
    rng = np.random.default_rng(2025 if NOISE else 2024)
    gpu = ca.CasterHip(0)
    w, h, spp = 320, 180, 8
    bad = 0
    for case in range(CASES):
        fs, centre, extent, n = random_scene(rng, case)
        gpu.upload_scene(fs)
        eye = centre + rng.normal(size=3) * extent * float(rng.choice([0.3, 1.5, 4.0]))
        # lens from a pinhole to half the cloud's extent across, focus plane before, in and behind the cloud
        dist = float(np.linalg.norm(eye - centre))
        aperture = extent * float(rng.choice([0.0, 0.02, 0.5]))
        focus = max(dist, 1e-6 * extent) * float(rng.choice([0.3, 1.0, 3.0]))
        cam = ca.calculate_cam_info(tuple(eye), tuple(centre), (0, 1, 0), 40.0, np.float32(w / h), aperture, focus)
        line = f"case {case:2d} n={n:5d} extent={extent:9.3g} |centre|={np.linalg.norm(centre):9.3g}"
        for strict in (True, False):
            m, mr, _ = gpu.render_image(cam, w, h, spp, 20, case, strict=strict)
            v, vr, _ = gpu.render_image(cam, w, h, spp, 20, case, strict=strict, vector_loop=True)
            same = mr == vr and np.array_equal(m, v, equal_nan=True)
            # the same frame with every camera ray meeting every sphere (no per-tile lists)
            a, ar, _ = gpu.render_image(cam, w, h, spp, 20, case, strict=strict, tile_lists=False)
            same = same and mr == ar and np.array_equal(m, a, equal_nan=True)
            bad += 0 if same else 1
            line += f"  {'strict' if strict else 'fast'}: rays {mr} {'==' if same else '!='}"
        print(line, flush=True)
    print("mismatching frames:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
