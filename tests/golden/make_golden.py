#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/ from the CPU restatement (oracle/).

The reference ships no tests, fixtures or golden images and cannot be compiled in this image
(every translation unit includes <windows.h>), so these vectors come from the restatement itself:
they pin the restatement against regressions and give the GPU tests fixed expected values, but
they do not pin it to the reference ("parity unpinned").

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import craytracer_amd as ca  # noqa: E402  (host side only: scenes and camera)
from oracle import oracle as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

IMAGE_CASES = {
    # name: (scene factory, width, height, spp, depth, seed)
    "scene_0": (lambda: ca.scene_0(), 64, 36, 16, 50, 0xC0FFEE),
    "scene_0_textured": (lambda: ca.scene_0(textured=True), 48, 27, 8, 50, 17),
    "rtiow": (lambda: ca.scene_rtiow(1), 48, 27, 4, 50, 3),
    "perlin": (lambda: ca.scene_perlin(1), 48, 27, 4, 50, 7),
}


def main():
    images = {}
    for name, (make, w, h, spp, depth, seed) in IMAGE_CASES.items():
        fs = make()
        cam = ca.default_camera(w / h)
        sums, rays = orc.render(orc.Scene.from_flat(fs), cam, w, h, spp, depth, seed)
        images[name + "_sums"] = sums
        images[name + "_rays"] = np.int64(rays)
        images[name + "_rgba"] = orc.resolve(sums, spp)
        images[name + "_cfg"] = np.array([w, h, spp, depth, seed], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "images.npz"), **images)

    rng = np.random.default_rng(2024)
    vec = {}
    vec["rng_stream"] = orc.rng(0xC0FFEE, 12345, 7, 64)
    n = 512
    sph = np.zeros((n, 4), np.float32)
    sph[:, :3] = rng.uniform(-8, 8, (n, 3))
    sph[:, 3] = rng.uniform(0.2, 1.5, n)
    sph[:16] = [0, -1000, 0, 1000]
    sph[16:32] = [0, 0, 0, 100000]
    o = rng.uniform(-12, 12, (n, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) * 0.3 + 0.05
    tgt = sph[:, :3] + rng.normal(size=(n, 3)).astype(np.float32) * sph[:, 3:4] * 0.9
    d = ((tgt - o) * rng.uniform(0.2, 2.0, (n, 1))).astype(np.float32)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    tr = np.tile(np.array([0.001, 16777215.0], np.float32), (n, 1))
    res = orc.hit_sphere(sph, rays, tr)
    vec.update(hs_sph=sph, hs_rays=rays, hs_tr=tr, hs_hit=res["hit"], hs_t=res["t"], hs_p=res["p"], hs_n=res["n"],
               hs_face=res["face"], hs_uv=res["uv"])
    cam = ca.default_camera(16 / 9)
    uv = rng.uniform(0, 1, (256, 2)).astype(np.float32)
    cr = orc.camera_ray(cam, uv, 99)
    vec.update(cam=cam, cam_uv=uv, cam_rays=cr["ray"], cam_draws=cr["draws"])
    cosv = np.linspace(0, 1, 33, dtype=np.float32)
    vec["schlick_cos"] = cosv
    vec["schlick_15"] = np.array([orc.reflectance(c, 1.5) for c in cosv], np.float32)
    vec["schlick_inv15"] = np.array([orc.reflectance(c, np.float32(1.0) / np.float32(1.5)) for c in cosv], np.float32)
    noise = ca.make_noise(1)
    pts = rng.uniform(-30, 30, (256, 3)).astype(np.float32)
    vec.update(noise_tables=np.frombuffer(noise.tobytes(), np.uint8), noise_points=pts,
               noise_value=orc.noise_value(noise, pts), noise_turb=orc.noise_turb(noise, pts))
    np.savez_compressed(os.path.join(OUT, "vectors.npz"), **vec)
    print("wrote", os.path.join(OUT, "images.npz"), os.path.join(OUT, "vectors.npz"))


if __name__ == "__main__":
    main()
