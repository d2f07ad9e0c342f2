"""Developer timing loop: renders one scene on cuda:0 in either arithmetic mode and prints
Mrays/s and the fraction of the FP32 VALU peak (157.3 TFLOP/s)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import craytracer_amd.binding as _b  # noqa: E402

if "--lib" in sys.argv:     # a build variant of libcaster_hip (developer A/B runs)
    _i = sys.argv.index("--lib")
    _variant = sys.argv[_i + 1]
    del sys.argv[_i:_i + 2]
    _paths = _b.lib_paths()
    _paths["caster"] = os.path.join(os.path.dirname(_paths["caster"]), _variant)
    _b.lib_paths = lambda: _paths
import craytracer_amd as ca  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="rtiow")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=32)
ap.add_argument("--depth", type=int, default=50)
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--modes", default="fast,strict")
ap.add_argument("--vector-loop", action="store_true", help="sphere filter on the vector unit (CASTER_HIP_VECTOR_LOOP)")
ap.add_argument("--order", default="list", choices=["list", "morton", "random"],
                help="experiment: reorder the scene's object list (changes which of two equal hits wins, nothing else)")
args = ap.parse_args()

if args.scene == "rtiow":
    fs = ca.scene_rtiow(1)
elif args.scene == "perlin":
    fs = ca.scene_perlin(1)
elif args.scene == "scene_0":
    fs = ca.scene_0()
elif args.scene.startswith("stress:"):
    fs = ca.scene_stress(int(args.scene.split(":")[1]), 2)
else:
    raise SystemExit("unknown scene")
if args.order != "list":
    g = fs.objects["g"]
    if args.order == "random":
        perm = np.random.default_rng(1).permutation(len(fs.objects))
    else:
        small = g[:, 3] < 50
        lo, hi = g[small, :3].min(axis=0), g[small, :3].max(axis=0)
        q = np.clip((g[:, :3] - lo) / np.maximum(hi - lo, 1e-9) * 1023, 0, 1023).astype(np.uint64)

        def spread(v):      # 10 bits -> every third bit
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249

        code = spread(q[:, 0]) | (spread(q[:, 2]) << 1) | (spread(q[:, 1]) << 2)
        code[~small] = np.uint64(1) << 40          # the huge spheres last
        perm = np.argsort(code, kind="stable")
    fs = ca.FlatScene(fs.objects[perm].copy(), fs.materials, fs.images, fs.noise)
ns, nr = fs.n_spheres, fs.n_rects
gpu = ca.CasterHip(0)
gpu.upload_scene(fs)
if args.chunk:
    gpu.set_chunk(args.chunk)
cam = ca.default_camera(args.width / args.height)
nbytes = args.width * args.height * 16
acc = gpu.malloc(nbytes)
for mode in args.modes.split(","):
    strict = mode == "strict"
    for rep in range(args.reps):
        gpu.memset_zero(acc, nbytes)
        gpu.synchronize()
        t0 = time.perf_counter()
        gpu.render(cam, args.width, args.height, 0, args.spp, args.depth, 0xC0FFEE, acc, strict=strict, vector_loop=args.vector_loop)
        info = gpu.render_info()
        rays, ms = info["rays"], info["kernel_ms"]
        wall = time.perf_counter() - t0
        flops = rays * (18 * ns + 12 * nr + 100)
        print(f"{mode:6s} rep{rep} rays {rays} ({rays / (args.width * args.height * args.spp):.2f}/path) kernel {ms:.2f} ms wall {wall * 1e3:.2f} ms "
              f"{rays / ms * 1e-3:.1f} Mrays/s  {flops / (ms * 1e-3) / 157.3e12 * 100:.1f}% of FP32 peak (algorithmic)  "
              f"loop {info['loop']} blocks {info['mfma_blocks']} trips {info['wave_trips']} batches {info.get('primary_batches', 0)}", flush=True)
gpu.free(acc)
