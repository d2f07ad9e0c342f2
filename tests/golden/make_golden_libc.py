#!/usr/bin/env python3
"""Renders the high-spp reference images on libc rand() into tests/golden/libc_images.npz.

The reference draws every random number from rand() (src/common.c:10-13), seeded from the wall clock
(src/common.c:4-8): its images are not reproducible, only their expectation is.  The third parity
check asks that a high-spp image of the HIP path agree with such an image to a mean absolute error
below 1/255 and a PSNR above 45 dB.  The images here come from the CPU restatement (oracle/) with its
random source switched to libc rand() exactly as common.c uses it -- still the restatement, not a
build of the reference (its sources need <windows.h>): parity unpinned, as everywhere.

rand() keeps one hidden state per process, so the rows of a frame are rendered by worker processes,
each seeded with its own srand() value.  Sample counts are sized from the per-sample variance measured
on the same scenes (gamma space: 0.020 for the random-spheres scene, 0.29 for scene_0), so that the
noise of this image plus that of a 4 x longer GPU render stays near 49 dB.

Run from the repository root:  python tests/golden/make_golden_libc.py   (about two minutes on 8 cores)
"""
import ctypes
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (width, height, spp, depth)
    "rtiow": (128, 72, 2048, 50),
    "scene_0": (64, 36, 32768, 50),
}


def _scene(name):
    import craytracer_amd as ca

    return ca.scene_rtiow(1) if name == "rtiow" else ca.scene_0()


def _rows(job):
    name, j0, j1, srand_seed = job
    import craytracer_amd as ca
    from oracle import oracle as orc

    w, h, spp, depth = CASES[name]
    ctypes.CDLL(None).srand(ctypes.c_uint(srand_seed))
    sums, rays = orc.render(orc.Scene.from_flat(_scene(name)), ca.default_camera(w / h), w, h, spp, depth, 0,
                            rng_mode=orc.RNG_LIBC, rows=(j0, j1))
    return j0, j1, sums[j0:j1], rays


def main():
    workers = min(8, os.cpu_count() or 1)
    out = {}
    with mp.get_context("spawn").Pool(workers) as pool:
        for name, (w, h, spp, depth) in CASES.items():
            step = max(1, h // (4 * workers))
            jobs = [(name, j0, min(h, j0 + step), 1000 + k) for k, j0 in enumerate(range(0, h, step))]
            mean = np.zeros((h, w, 3), np.float32)
            rays = 0
            for j0, j1, part, r in pool.imap_unordered(_rows, jobs):
                assert np.all(part[..., 3] == spp)
                mean[j0:j1] = part[..., :3] / np.float32(spp)
                rays += r
            out[name + "_mean"] = mean                     # linear radiance per sample, row 0 = image bottom
            out[name + "_cfg"] = np.array([w, h, spp, depth], np.int64)
            out[name + "_rays"] = np.int64(rays)
            print(name, w, h, spp, "rays", rays, flush=True)
    np.savez_compressed(os.path.join(OUT, "libc_images.npz"), **out)
    print("wrote", os.path.join(OUT, "libc_images.npz"))


if __name__ == "__main__":
    main()
