"""Camera rays are traced against a per-tile list of the spheres their beam can touch (a conservative
bound over the lens disk and the tile's patch of the focus plane, src/camera.c:32-42) instead of the
whole sphere list that hit_list_hit_all walks (src/caster.c:198-236).  A sphere missing from a list
would silently drop a hit, so whole frames are held against the same frames rendered with every
camera ray meeting every sphere (CASTER_HIP_NO_TILE_LISTS): same bits, same ray counts, for cameras
that stress the bound -- wide lenses, a lens inside the sphere field, grazing views, a focus plane
close to the lens, a scene thousands of units from the origin -- in both arithmetic modes and both
sphere loops."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CAMERAS = {
    # name: (position, look-at, fov, aperture, focus distance)
    "benchmark": ((13, 2, 3), (0, 0, 0), 20.0, 0.1, 10.0),
    "wide_lens": ((13, 2, 3), (0, 0, 0), 20.0, 3.0, 10.0),            # a lens 3 units across: every tile's beam is fat
    "inside_the_field": ((0.6, 0.25, 0.55), (4, 0.3, 0.4), 70.0, 0.2, 1.5),     # lens between the small spheres, near focus plane
    "grazing": ((-12, 0.21, 9), (8, 0.2, -6), 35.0, 0.05, 30.0),      # along the ground: long lists
    "top_down": ((0.3, 25, 0.2), (0, 0, 0), 50.0, 0.5, 25.0),
    "close_focus": ((6, 1.2, 2), (4, 1, 0), 40.0, 0.4, 0.05),         # focus plane 5 cm from a 40 cm lens: beam wider than long
    "pinhole": ((13, 2, 3), (0, 0, 0), 20.0, 0.0, 10.0),
}


def frames_equal(gpu, cam, w, h, spp, depth, seed, strict, vector_loop):
    a, ra, _ = gpu.render_image(cam, w, h, spp, depth, seed, strict=strict, vector_loop=vector_loop)
    info = gpu.render_info()
    b, rb, _ = gpu.render_image(cam, w, h, spp, depth, seed, strict=strict, vector_loop=vector_loop, tile_lists=False)
    assert info["primary_batches"] > 0
    assert ra == rb, (ra, rb)
    assert np.array_equal(a, b)
    return ra


@pytest.mark.parametrize("name", sorted(CAMERAS))
def test_tile_lists_change_nothing(gpu, ca, name):
    pos, at, fov, aperture, focus = CAMERAS[name]
    fs = ca.scene_rtiow(1)
    gpu.upload_scene(fs)
    w, h, spp, depth = 200, 112, 8, 50
    cam = ca.calculate_cam_info(pos, at, (0, 1, 0), fov, np.float32(w / h), aperture, focus)
    for strict in (True, False):
        for vector_loop in (False, True):
            rays = frames_equal(gpu, cam, w, h, spp, depth, 0xBEA3 + len(name), strict, vector_loop)
            assert rays >= w * h * spp


def test_tile_lists_on_large_and_distant_scenes(gpu, ca):
    """10 002 spheres (lists of dozens of spheres, exact operands from device memory) and the benchmark
    layout 3 000 units from the origin (the bound's slack scales with the distances involved)."""
    fs = ca.scene_stress(10000, 2)
    gpu.upload_scene(fs)
    w, h = 160, 90
    cam = ca.default_camera(w / h)
    for strict in (True, False):
        frames_equal(gpu, cam, w, h, 4, 50, 11, strict, False)
    frames_equal(gpu, cam, w, h, 2, 50, 12, True, True)
    fs = ca.scene_rtiow(5)
    objs = fs.objects.copy()
    shift = np.array([3000.0, -40.0, 1500.0], dtype=np.float32)
    objs["g"][:, :3] += shift
    gpu.upload_scene(ca.FlatScene(objs, fs.materials, fs.images, fs.noise))
    cam = ca.calculate_cam_info(tuple(np.array([13, 2, 3], np.float32) + shift), tuple(shift), (0, 1, 0), 20.0, np.float32(w / h), 0.1, 10.0)
    for strict in (True, False):
        frames_equal(gpu, cam, w, h, 8, 50, 13, strict, False)
