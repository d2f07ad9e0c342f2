"""The bound behind the per-tile sphere lists of the camera rays, restated in numpy and checked on the
CPU against brute force: for random cameras (src/camera.c:3-30 through the C host), tiles and scenes,
every sphere that a camera ray of the tile hits -- rays drawn as src/dispatcher.c:21-24 and
src/camera.c:32-42 draw them, lens points on the rim of the disk and pixel positions on the tile's
edges included, intersections in float64 -- must be among the spheres the bound keeps.  The GPU test
(test_gpu_tile_lists.py) checks the implementation end to end; this one checks the geometry it rests
on, without a GPU.  build_tile_list in craytracer_amd/csrc/caster_render.hpp is the device form of
`spheres_in_reach` below (same quantities, same inflation factors)."""
import numpy as np
import pytest


def spheres_in_reach(cam, width, height, x0, y0, tw, th, sph):
    """Mask of the spheres [n, 4] = (cx, cy, cz, r) the beam of tile (x0, y0, tw, th) can touch; None when the
    bound does not apply (the kernel then tests every sphere)."""
    o, llc, h, v, u0, v0, lr = cam[0:3], cam[3:6], cam[6:9], cam[9:12], cam[12:15], cam[15:18], cam[18]
    if width < 2 or height < 2:
        return None
    iw, ih = 1.0 / (width - 1), 1.0 / (height - 1)
    hu, hv = 0.5 * tw * iw, 0.5 * th * ih
    uc, vc = x0 * iw + hu, y0 * ih + hv
    ax = (llc + (h * uc + v * vc)) - o
    L = np.sqrt(ax @ ax)
    rho_q = (hu * np.sqrt(h @ h) + hv * np.sqrt(v @ v)) * 1.002
    rho_l = abs(lr) * 1.002 * np.sqrt(max(u0 @ u0, v0 @ v0) + abs(u0 @ v0))
    kk = rho_l + rho_q
    if not (L - kk > 1.0e-3 * L) or not (L < 1.0e30):
        return None
    e = ax / L
    inv_lo, inv_hi = 0.999 / (L + kk), 1.001 / (L - kk)
    vv = sph[:, :3] - o
    z = vv @ e
    w = vv - np.outer(z, e)
    dperp = np.sqrt((w * w).sum(1))
    r = np.abs(sph[:, 3])
    slack = 1.0e-5 * ((np.abs(z) + dperp) + L)
    R0 = (r + rho_l) * 1.001 + slack
    s_hi = (z + R0) * inv_hi
    s_lo = np.maximum((z - R0) * inv_lo, 0.0)
    f_lo = np.abs(1.0 - s_lo) * rho_l + s_lo * rho_q
    f_hi = np.abs(1.0 - s_hi) * rho_l + s_hi * rho_q
    bound = (r + np.maximum(f_lo, f_hi)) * 1.001 + slack
    out = (z + R0 < 0.0) | (dperp > bound)
    return ~out


def camera_rays(rng, cam, width, height, x0, y0, tw, th, n):
    """n camera rays of the tile: pixel + jitter over the frame size, lens point in the unit disk times the lens
    radius; a share of them on the extremes (tile edges, lens rim)."""
    o, llc, h, v, u0, v0, lr = cam[0:3], cam[3:6], cam[6:9], cam[9:12], cam[12:15], cam[15:18], cam[18]
    px = x0 + rng.uniform(0, tw, n)
    py = y0 + rng.uniform(0, th, n)
    edge = rng.random(n) < 0.3
    px[edge] = np.where(rng.random(edge.sum()) < 0.5, x0, x0 + tw * (1 - 1e-7))
    edge = rng.random(n) < 0.3
    py[edge] = np.where(rng.random(edge.sum()) < 0.5, y0, y0 + th * (1 - 1e-7))
    u, vv = px / (width - 1), py / (height - 1)
    ang = rng.uniform(0, 2 * np.pi, n)
    rad = np.sqrt(rng.random(n))
    rad[rng.random(n) < 0.3] = 1.0 - 1e-9
    rdx, rdy = rad * np.cos(ang) * lr, rad * np.sin(ang) * lr
    off = np.outer(rdx, u0) + np.outer(rdy, v0)
    org = o + off
    d = llc + (np.outer(u, h) + ((np.outer(vv, v) - o) - off))
    return org, d


def hits(org, d, sph):
    """[rays, spheres] True where the ray meets the sphere at some t > 0 (float64)."""
    oc = org[:, None, :] - sph[None, :, :3]
    a = (d * d).sum(1)[:, None]
    hb = (oc * d[:, None, :]).sum(2)
    c = (oc * oc).sum(2) - sph[None, :, 3] ** 2
    disc = hb * hb - a * c
    ok = disc >= 0
    sq = np.sqrt(np.where(ok, disc, 0.0))
    return ok & ((-hb + sq) / a > 0.0)


@pytest.mark.parametrize("case", range(12))
def test_every_sphere_a_camera_ray_hits_is_in_reach_of_its_tile(case):
    import craytracer_amd as ca

    rng = np.random.default_rng(9000 + case)
    width, height = int(rng.integers(40, 400)), int(rng.integers(30, 300))
    n = int(rng.integers(50, 600))
    extent = float(np.exp(rng.uniform(np.log(0.1), np.log(1e3))))
    centre = rng.normal(size=3) * extent * float(rng.choice([0.0, 1.0, 30.0]))
    sph = np.empty((n, 4))
    sph[:, :3] = centre + rng.uniform(-extent, extent, (n, 3))
    sph[:, 3] = extent * np.exp(rng.uniform(np.log(2e-3), np.log(0.2), n))
    eye = centre + rng.normal(size=3) * extent * float(rng.choice([0.2, 1.5, 5.0]))
    dist = float(np.linalg.norm(eye - centre))
    aperture = extent * float(rng.choice([0.0, 0.01, 0.3]))
    focus = max(dist, 1e-3 * extent) * float(rng.choice([0.3, 1.0, 3.0]))
    cam = ca.calculate_cam_info(tuple(eye), tuple(centre), (0, 1, 0), float(rng.choice([15.0, 40.0, 90.0])),
                                np.float32(width / height), aperture, focus).astype(np.float64)
    checked = kept = applied = 0
    for _ in range(25):
        x0, y0 = int(rng.integers(0, width) // 8 * 8), int(rng.integers(0, height) // 8 * 8)
        tw, th = min(8, width - x0), min(8, height - y0)
        reach = spheres_in_reach(cam, width, height, x0, y0, tw, th, sph)
        if reach is None:
            continue
        applied += 1
        org, d = camera_rays(rng, cam, width, height, x0, y0, tw, th, 400)
        hit = hits(org, d, sph)
        assert not (hit & ~reach[None, :]).any(), (case, x0, y0, int((hit & ~reach[None, :]).sum()))
        checked += int(hit.sum())
        kept += int(reach.sum())
    if applied:
        assert kept < applied * n          # the bound does cull: the lists are not simply everything


def test_the_bound_is_tight_enough_to_matter():
    """On the benchmark scene and camera a tile's list holds a few spheres of the 485 that go through it."""
    import craytracer_amd as ca

    fs = ca.scene_rtiow(1)
    g = fs.objects["g"][fs.objects["type"] == 0][:, :4].astype(np.float64)
    small = g[g[:, 3] < 50]
    cam = ca.default_camera(1920 / 1080).astype(np.float64)
    sizes = []
    for y0 in range(0, 1080, 72):
        for x0 in range(0, 1920, 120):
            reach = spheres_in_reach(cam, 1920, 1080, x0, y0, 8, 8, small)
            assert reach is not None
            sizes.append(int(reach.sum()))
    assert np.mean(sizes) < 6 and max(sizes) < 60
