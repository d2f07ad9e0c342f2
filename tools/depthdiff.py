"""Rays shot as a function of max_depth, strict vs fast: shows at which bounce the two modes part."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import craytracer_amd as ca
gpu = ca.CasterHip(0)
w, h, spp, seed = 192, 108, 32, 7
cam = ca.default_camera(w / h)
for name, fs in (("rtiow", ca.scene_rtiow(1)), ("scene_0", ca.scene_0())):
    gpu.upload_scene(fs)
    prev = (0, 0)
    for depth in (1, 2, 3, 4, 6, 10, 50):
        _, rs, _ = gpu.render_image(cam, w, h, spp, depth, seed, strict=True)
        _, rf, _ = gpu.render_image(cam, w, h, spp, depth, seed, strict=False)
        print(name, 'depth', depth, 'strict', rs, 'fast', rf, 'delta', rf - rs, 'incr strict', rs - prev[0], 'incr fast', rf - prev[1])
        prev = (rs, rf)
