"""Developer check: prints a digest of the per-pixel sums of one small frame per arithmetic mode and sphere
loop, for a build variant of libcaster_hip (--lib, as tools/quick_bench.py).  Two builds whose arithmetic is
meant to be the same print the same digests."""
import argparse
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import craytracer_amd.binding as _b  # noqa: E402

if "--lib" in sys.argv:
    _i = sys.argv.index("--lib")
    _variant = sys.argv[_i + 1]
    del sys.argv[_i:_i + 2]
    _paths = _b.lib_paths()
    _paths["caster"] = os.path.join(os.path.dirname(_paths["caster"]), _variant)
    _b.lib_paths = lambda: _paths
import craytracer_amd as ca  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="perlin")
ap.add_argument("--width", type=int, default=320)
ap.add_argument("--height", type=int, default=180)
ap.add_argument("--spp", type=int, default=24)
args = ap.parse_args()
fs = {"perlin": lambda: ca.scene_perlin(1), "rtiow": lambda: ca.scene_rtiow(1), "scene_0": ca.scene_0}[args.scene]()
gpu = ca.CasterHip(0)
gpu.upload_scene(fs)
cam = ca.default_camera(args.width / args.height)
for strict in (True, False):
    for vector_loop in (False, True):
        s, rays, _ = gpu.render_image(cam, args.width, args.height, args.spp, 50, 77, strict=strict, vector_loop=vector_loop)
        print("strict" if strict else "fast", "vector" if vector_loop else "matrix", rays, hashlib.sha256(s.tobytes()).hexdigest()[:16])
