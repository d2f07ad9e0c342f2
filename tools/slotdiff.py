import sys, numpy as np
sys.path.insert(0, '/root/repo')
import craytracer_amd as ca
gpu = ca.CasterHip(0)
fs = ca.scene_rtiow(1)
gpu.upload_scene(fs)
w, h, depth, seed = 128, 72, 50, 5
cam = ca.default_camera(w / h)
for strict in (True, False):
    gpu.set_chunk(2)
    both, rb, _ = gpu.render_image(cam, w, h, 2, depth, seed, strict=strict, sample_begin=10)
    a, ra, _ = gpu.render_image(cam, w, h, 1, depth, seed, strict=strict, sample_begin=10)
    b, rb1, _ = gpu.render_image(cam, w, h, 1, depth, seed, strict=strict, sample_begin=11)
    d = both[..., :3] - a[..., :3]
    bad = np.abs(d - b[..., :3]).max(axis=2) > 1e-5
    print('strict' if strict else 'fast', 'rays', rb, ra + rb1, 'pixels whose slot-B path differs from the slot-A path:', bad.sum(), 'of', w * h)
