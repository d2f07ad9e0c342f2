"""Developer check: the same frame rendered many times must give the same bits every time (the unit-to-wave
assignment, the order in which waiting paths are taken up and the route of every path's radiance differ from run
to run; the sums are integers, so nothing of that may show).  usage: python tools/soak.py [--reps 40] [--scene rtiow]"""
import argparse
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import craytracer_amd as ca  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--scene", default="rtiow")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=6)
args = ap.parse_args()
fs = {"rtiow": lambda: ca.scene_rtiow(1), "perlin": lambda: ca.scene_perlin(1), "stress": lambda: ca.scene_stress(3000, 2)}[args.scene]()
cam = ca.default_camera(args.width / args.height)
gpu = ca.CasterHip(0)
gpu.upload_scene(fs)
bad = 0
for strict in (True, False):
    for vector in (False, True):
        seen = {}
        for k in range(args.reps):
            s, rays, _ = gpu.render_image(cam, args.width, args.height, args.spp, 50, 99, strict=strict, vector_loop=vector)
            seen.setdefault((rays, hashlib.sha256(s.tobytes()).hexdigest()[:16]), []).append(k)
        print(f"{args.scene} {'strict' if strict else 'fast'} {'vector' if vector else 'matrix'} loop: {args.reps} renders, {len(seen)} distinct result(s): {list(seen)[:3]}", flush=True)
        bad += len(seen) - 1
print("distinct results beyond the first:", bad)
sys.exit(1 if bad else 0)
