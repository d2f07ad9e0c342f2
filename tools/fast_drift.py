"""Developer check: how far the fast arithmetic mode drifts from the strict one, scene by scene:
ray counts, rays per bounce depth bucket, and the image distance in display space."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import craytracer_amd as ca  # noqa: E402
from campaign_two_loops import random_scene  # noqa: E402


def compare(gpu, name, fs, cam, w, h, spp, depth, seed):
    gpu.upload_scene(fs)
    out = []
    for d in (depth, 10, 5):
        s, rs, _ = gpu.render_image(cam, w, h, spp, d, seed, strict=True)
        f, rf, _ = gpu.render_image(cam, w, h, spp, d, seed, strict=False)
        a = np.sqrt(np.clip(s[..., :3] / spp, 0, None)).clip(0, 1)
        b = np.sqrt(np.clip(f[..., :3] / spp, 0, None)).clip(0, 1)
        mae = np.abs(a - b).mean()
        psnr = 10 * np.log10(1.0 / max(np.mean((a - b) ** 2), 1e-12))
        out.append(f"depth {d:2d}: rays {rs} / {rf} ({(rf - rs) / rs * 100:+.2f} %) mae*255 {mae * 255:.3f} psnr {psnr:.1f}")
    mats = fs.materials["type"][fs.objects["material"]]
    print(f"{name:14s} n={len(fs.objects):5d} glass {np.mean(mats == 2) * 100:4.1f} %  " + " | ".join(out), flush=True)


def main():
    gpu = ca.CasterHip(0)
    w, h, spp = 320, 180, 32
    cam = ca.default_camera(w / h)
    compare(gpu, "rtiow", ca.scene_rtiow(1), cam, w, h, spp, 50, 1234)
    compare(gpu, "stress_500", ca.scene_stress(500, 2), cam, w, h, spp, 50, 1234)
    compare(gpu, "stress_4500", ca.scene_stress(4500, 2), cam, w, h, spp, 50, 1234)
    compare(gpu, "scene_0", ca.scene_0(), cam, w, h, spp, 50, 1234)
    fs = ca.scene_rtiow(5)
    objs = fs.objects.copy()
    objs["g"][:, 0] += 3000.0
    objs["g"][:, 2] -= 1500.0
    ocam = ca.calculate_cam_info((3013, 2, -1497), (3000, 0, -1500), (0, 1, 0), 20.0, np.float32(w / h), 0.1, 10.0)
    compare(gpu, "offset", ca.FlatScene(objs, fs.materials, []), ocam, w, h, spp, 50, 1234)
    rng = np.random.default_rng(2024)
    for case in range(40):
        fs, centre, extent, n = random_scene(rng, case)
        eye = centre + rng.normal(size=3) * extent * float(rng.choice([0.3, 1.5, 4.0]))
        if case in (3, 12, 25, 37):
            ccam = ca.calculate_cam_info(tuple(eye), tuple(centre), (0, 1, 0), 40.0, np.float32(w / h), 0.0, 1.0)
            compare(gpu, f"campaign_{case} ext {extent:.3g}", fs, ccam, w, h, 8, 20, case)


if __name__ == "__main__":
    main()
