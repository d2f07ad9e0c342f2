"""Developer probe: frames rendered back to back on one context against frames alternating between two contexts
(two streams), so that the end of one frame -- waves that still hold a few long paths -- overlaps with the start of the next.
usage: python tools/pipeline_probe.py [--spp 62] [--frames 12]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (streams only)

import craytracer_amd as ca  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=62)
ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
args = ap.parse_args()
fs = ca.scene_rtiow(1)
cam = ca.default_camera(args.width / args.height)
n = args.width * args.height * 16
ctxs = [ca.CasterHip(0), ca.CasterHip(0)]
streams = [torch.cuda.Stream(device=0), torch.cuda.Stream(device=0)]
accs = []
for g, st in zip(ctxs, streams):
    g.set_stream(st.cuda_stream)
    g.upload_scene(fs)
    accs.append(g.malloc(n))
    g.memset_zero(accs[-1], n)
    g.render(cam, args.width, args.height, 0, args.spp, 50, 1, accs[-1], strict=True)
    g.synchronize()
for label, pick in (("one context", lambda k: 0), ("two contexts", lambda k: k & 1), ("one context", lambda k: 0), ("two contexts", lambda k: k & 1)):
    for g in ctxs:
        g.synchronize()
    t0 = time.perf_counter()
    for k in range(args.frames):
        c = pick(k)
        ctxs[c].memset_zero(accs[c], n)
        ctxs[c].render(cam, args.width, args.height, 0, args.spp, 50, 1 + k, accs[c], strict=True)
    for g in ctxs:
        g.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label}: {args.frames} frames of {args.spp} spp in {dt * 1e3:.2f} ms = {dt * 1e3 / args.frames:.3f} ms per frame", flush=True)
