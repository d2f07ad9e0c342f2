"""Not collected by pytest (run it by name on a GPU box: python tests/big_frame_check.py [samples per pixel, default 2]).
Developer check: strict-mode frames at benchmark size against the CPU restatement (same random
streams): ray counts must be equal and every channel sum within 1e-4 relative."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import craytracer_amd as ca
from oracle import oracle as orc
SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for name, fs in (("rtiow", ca.scene_rtiow(1)), ("perlin", ca.scene_perlin(1)), ("stress1500", ca.scene_stress(1500, 2))):
    w, h, spp, depth, seed = (1920, 1080, SPP, 50, 0xC0FFEE) if name != "stress1500" else (640, 360, SPP, 50, 5)
    cam = ca.default_camera(w / h)
    gpu = ca.CasterHip(0); gpu.upload_scene(fs)
    g, g_rays, ms = gpu.render_image(cam, w, h, spp, depth, seed, strict=True)
    t0 = time.time()
    o, _, o_rays = orc.dispatch(orc.Scene.from_flat(fs), cam, w, h, spp, depth, seed, 15)
    err = np.abs(g[..., :3].astype(np.float64) - o[..., :3].astype(np.float64))
    tol = 1e-4 * np.abs(o[..., :3].astype(np.float64)) + 1e-7
    print(name, f"{w}x{h}x{spp}spp", "rays", g_rays, o_rays, "equal" if g_rays == o_rays else "DIFFER", "bad", int((err > tol).sum()), "max err", err.max(), "cpu s %.1f" % (time.time() - t0), flush=True)
    gpu.close()
