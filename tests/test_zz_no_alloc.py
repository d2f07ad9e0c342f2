"""caster_hip_render_noalloc, the render entry point that allocates nothing (its scratch is a static
device array), against caster_hip_render through the C ABI: same bits, same ray count, on two calls
in a row, the second on a smaller frame (the scratch keeps the first call's contents)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(gpu, cam, w, h, s0, spp, depth, seed, strict, vector_loop=False):
    nb = w * h * 16
    a, b = gpu.malloc(nb), gpu.malloc(nb)
    try:
        gpu.memset_zero(a, nb)
        gpu.memset_zero(b, nb)
        gpu.render_noalloc(cam, w, h, s0, spp, depth, seed, a, strict=strict, vector_loop=vector_loop)
        ia = gpu.render_info()
        gpu.render(cam, w, h, s0, spp, depth, seed, b, strict=strict, vector_loop=vector_loop)
        ib = gpu.render_info()
        ha = gpu.download(a, (h, w, 4), np.float32)
        hb = gpu.download(b, (h, w, 4), np.float32)
    finally:
        gpu.free(a)
        gpu.free(b)
    return ha, ia, hb, ib


def test_no_alloc_render_equals_render(gpu, ca):
    gpu.upload_scene(ca.scene_rtiow(1))
    # first a frame that fills most of the scratch, then a smaller one, then an odd-sized one
    for (w, h, s0, spp, strict) in ((1024, 576, 0, 3, True), (200, 100, 3, 5, True), (37, 21, 0, 9, False)):
        cam = ca.default_camera(w / h)
        ha, ia, hb, ib = _both(gpu, cam, w, h, s0, spp, 50, 77, strict)
        assert ia["rays"] == ib["rays"] and ia["rays"] > w * h * spp
        assert ia["wave_trips"] > 0 and ia["loop"] == ib["loop"] == "matrix"
        assert np.array_equal(ha, hb)
        assert np.all(ha[..., 3] == spp)
    # the vector loop and a scene with rects through the same entry point
    gpu.upload_scene(ca.scene_0())
    cam = ca.default_camera(64 / 36)
    ha, ia, hb, ib = _both(gpu, cam, 64, 36, 0, 6, 50, 5, True)
    assert ia["rays"] == ib["rays"] and np.array_equal(ha, hb) and ia["loop"] == "vector"


def test_no_alloc_limits(gpu, ca):
    gpu.upload_scene(ca.scene_0())
    acc = gpu.malloc(16)
    try:
        with pytest.raises(ca.CasterHipError, match="exceed"):
            gpu.render_noalloc(ca.default_camera(2.0), 2048, 1024, 0, 1, 5, 1, acc, strict=True)     # 2 Mi pixels
        # the largest frame it takes: exactly the scratch size, zero samples (nothing is written)
        gpu.render_noalloc(ca.default_camera(1.0), 1024, 1024, 0, 0, 5, 1, acc, strict=True)
        rays, _ = gpu.render_stats()
        assert rays == 0
    finally:
        gpu.free(acc)
