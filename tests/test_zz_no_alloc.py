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


def test_no_alloc_render_allocates_nothing(gpu, ca):
    """The property the entry point exists for: the GPU's free memory is the same before and after
    calls of caster_hip_render_noalloc, the first of them on a frame larger than anything
    caster_hip_render has seen on this context (a fresh one, so no scratch of an earlier frame covers it)."""
    with ca.CasterHip(0) as fresh:
        fresh.upload_scene(ca.scene_rtiow(1))
        w, h = 1024, 1000                                   # 1 024 000 pixels: just under the limit
        acc = fresh.malloc(w * h * 16)
        try:
            fresh.memset_zero(acc, w * h * 16)
            fresh.render(ca.default_camera(1.0), 64, 64, 0, 1, 5, 1, acc, strict=True)      # warms the library up on a small frame
            fresh.synchronize()
            free0, total = fresh.mem_info()
            assert 0 < free0 <= total
            for (fw, fh, spp) in ((w, h, 1), (300, 200, 2), (w, h, 1)):
                fresh.render_noalloc(ca.default_camera(fw / fh), fw, fh, 0, spp, 50, 9, acc, strict=True)
                info = fresh.render_info()
                assert info["rays"] >= fw * fh * spp
                assert fresh.mem_info()[0] == free0
            # the allocating entry point on the large frame does take memory (the check can see it): its scratch grows
            fresh.render(ca.default_camera(w / h), w, h, 0, 1, 50, 9, acc, strict=True)
            fresh.synchronize()
            assert fresh.mem_info()[0] < free0
        finally:
            fresh.free(acc)


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
