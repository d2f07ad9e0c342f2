"""Per-function parity of the HIP device functions (strict arithmetic mode) with the CPU
restatement of the reference functions, called through the C ABI's array entry points.

Bar: t, normal and scatter direction within 2 ULP-scaled FP32 on
identical inputs.  Everything that involves only +,-,*,/ and sqrt is in fact expected to be
bit-identical; libm-backed results (Schlick powf, uv atan2/acos) may differ in the last place.
"""
import numpy as np
import pytest

from conftest import random_rays, scaled_ulp, ulp_distance

pytestmark = pytest.mark.gpu

ULP_TOL = 2.0


def test_rng_stream_bit_exact(gpu, orc):
    for seed, pixel, sample in [(0, 0, 0), (0xC0FFEE, 12345, 7), (0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF), (1, 1920 * 1080 - 1, 499)]:
        a = gpu.test_rng(seed, pixel, sample, 257)
        b = orc.rng(seed, pixel, sample, 257)
        assert np.array_equal(a, b)
        assert a.max() < (1 << 24)


def _sphere_cases(rng, n):
    """Spheres and rays that exercise: misses, two-root hits, origin inside, grazing, ground r=1000, sky r=1e5."""
    sph = np.zeros((n, 4), np.float32)
    rays = random_rays(rng, n)
    kind = rng.integers(0, 6, size=n)
    sph[:, :3] = rng.uniform(-12, 12, size=(n, 3))
    sph[:, 3] = rng.uniform(0.1, 1.5, size=n)
    # aim most rays at their sphere (with an offset up to ~1.3 r => hits, grazes and near misses)
    aim = kind != 0
    tgt = sph[:, :3] + rng.normal(size=(n, 3)).astype(np.float32) * (sph[:, 3:4] * rng.uniform(0, 1.3, size=(n, 1)).astype(np.float32))
    d = (tgt - rays[:, :3]).astype(np.float32) * rng.uniform(0.1, 3.0, size=(n, 1)).astype(np.float32)
    rays[aim, 3:] = d[aim]
    inside = kind == 2
    rays[inside, :3] = sph[inside, :3] + (rng.uniform(-0.5, 0.5, size=(inside.sum(), 3)) * sph[inside, 3:4]).astype(np.float32)
    ground = kind == 3
    sph[ground] = np.array([0, -1000, 0, 1000], np.float32)
    sky = kind == 4
    sph[sky] = np.array([0, 0, 0, 100000], np.float32)
    tr = np.zeros((n, 2), np.float32)
    tr[:, 0] = 0.001
    tr[:, 1] = 16777215.0
    shrink = rng.random(n) < 0.3   # closest-so-far smaller than the far root, sometimes than both
    tr[shrink, 1] = rng.uniform(0.01, 3.0, size=shrink.sum())
    return sph, rays, tr


def test_hit_sphere_parity(gpu, orc):
    rng = np.random.default_rng(101)
    sph, rays, tr = _sphere_cases(rng, 6000)
    g = gpu.test_hit_sphere(sph, rays, tr, strict=True)
    o = orc.hit_sphere(sph, rays, tr)
    assert np.array_equal(g["hit"], o["hit"])
    assert 0.2 < o["hit"].mean() < 0.95
    h = o["hit"] == 1
    assert np.array_equal(g["face"][h], o["face"][h])
    assert ulp_distance(g["t"][h], o["t"][h]).max() <= ULP_TOL
    assert scaled_ulp(g["p"][h], o["p"][h], np.abs(o["p"][h]).max(axis=1, keepdims=True)).max() <= ULP_TOL
    assert scaled_ulp(g["n"][h], o["n"][h], 1.0).max() <= ULP_TOL
    # +,-,*,/ and sqrt only: expected identical bits
    assert np.array_equal(g["t"][h], o["t"][h])
    assert np.array_equal(g["n"][h], o["n"][h])


def test_hit_xy_rect_parity(gpu, orc):
    rng = np.random.default_rng(102)
    n = 3000
    rect = np.zeros((n, 5), np.float32)
    rect[:, :3] = rng.uniform(-6, 6, size=(n, 3))
    rect[:, 3:] = rng.uniform(0.5, 4, size=(n, 2))
    rays = random_rays(rng, n)
    tgt = rect[:, :3].copy()
    tgt[:, :2] += (rng.uniform(-0.7, 0.7, size=(n, 2)) * rect[:, 3:]).astype(np.float32)
    rays[:, 3:] = (tgt - rays[:, :3]) * rng.uniform(0.2, 2.0, size=(n, 1)).astype(np.float32)
    tr = np.tile(np.array([0.001, 16777215.0], np.float32), (n, 1))
    tr[rng.random(n) < 0.2, 1] = 0.5
    # edge cases: ray parallel to the plane (dz = 0 -> inf), ray starting on the plane
    rays[:5, 5] = 0.0
    rays[5:10, 2] = rect[5:10, 2]
    g = gpu.test_hit_xy_rect(rect, rays, tr, strict=True)
    o = orc.hit_xy_rect(rect, rays, tr)
    assert np.array_equal(g["hit"], o["hit"])
    assert 0.1 < o["hit"].mean() < 0.9
    h = o["hit"] == 1
    for key in ("t", "p", "n", "uv"):
        assert np.array_equal(g[key][h], o[key][h]), key
    assert np.array_equal(g["face"][h], o["face"][h])


@pytest.mark.parametrize("scene_name", ["scene_0", "rtiow", "ties"])
def test_hit_list_parity(gpu, orc, ca, scene_name):
    rng = np.random.default_rng(103)
    if scene_name == "scene_0":
        fs = ca.scene_0()
    elif scene_name == "rtiow":
        fs = ca.scene_rtiow(1)
    else:
        # duplicate objects: equal t for two list entries -> the later entry must win (src/caster.c:114,204)
        base = ca.scene_0()
        objs = np.concatenate([base.objects, base.objects[1:2], base.objects[3:4], base.objects[2:3]])
        mats = np.concatenate([base.materials, base.materials[:1]])
        objs = objs.copy()
        objs["material"][4] = 4   # duplicate of the glass sphere with another material
        objs["material"][5] = 0   # duplicate of the rect
        objs["material"][6] = 1   # duplicate of the third sphere, after a rect in list order
        fs = ca.FlatScene(objs, mats, [])
    gpu.upload_scene(fs)
    sc = orc.Scene.from_flat(fs)
    n = 4000
    rays = random_rays(rng, n)
    # aim at objects of the scene
    pick = rng.integers(0, len(fs.objects), size=n)
    tgt = fs.objects["g"][pick, :3] + rng.normal(size=(n, 3)).astype(np.float32) * 0.6
    aimed = rng.random(n) < 0.8
    rays[aimed, 3:] = (tgt - rays[:, :3])[aimed]
    g = gpu.test_hit_list(rays, strict=True)
    o = orc.hit_list(sc, rays)
    assert np.array_equal(g["hit"], o["hit"])
    h = o["hit"] == 1
    assert h.mean() > 0.5
    assert np.array_equal(g["mat"][h], o["mat"][h])
    assert np.array_equal(g["face"][h], o["face"][h])
    assert np.array_equal(g["t"][h], o["t"][h])
    assert np.array_equal(g["p"][h], o["p"][h])
    assert np.array_equal(g["n"][h], o["n"][h])
    if scene_name == "ties":
        assert (o["mat"][h] == 4).any() and not (o["mat"][h] == 1).all()


def _scatter_inputs(rng, ca, n):
    mats = np.zeros(4, ca.MATERIAL_DTYPE)
    mats["type"] = [0, 1, 2, 1]
    mats["color0"] = [[0.5, 0.5, 0.5], [0.8, 0.7, 0.6], [0, 0, 0], [0.9, 0.9, 0.9]]
    mats["fuzziness"] = [0, 0.3, 0, 0.0]
    mats["n_refract"] = [0, 0, 1.5, 0]
    mats["image"] = -1
    objs = np.zeros(1, ca.OBJECT_DTYPE)
    objs["g"][0, :4] = [0, 0, 0, 1]
    fs = ca.FlatScene(objs, mats, [])
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32) * rng.uniform(0.1, 10, size=(n, 1)).astype(np.float32)
    # incoming direction opposes the (face-corrected) normal, as hit_record_set_face guarantees
    flip = (d * nrm).sum(axis=1) > 0
    d[flip] = -d[flip]
    # some grazing directions so that total internal reflection occurs for back-face dielectric hits
    graze = rng.random(n) < 0.3
    d[graze] -= (0.97 * (d[graze] * nrm[graze]).sum(axis=1, keepdims=True) * nrm[graze]).astype(np.float32)
    p = rng.uniform(-5, 5, size=(n, 3)).astype(np.float32)
    rays = np.concatenate([p - d, d], axis=1).astype(np.float32)
    face = rng.integers(0, 2, size=n).astype(np.int32)
    mat = rng.integers(0, 4, size=n).astype(np.int32)
    return fs, rays, p, nrm, face, mat


def test_scatter_parity(gpu, orc, ca):
    rng = np.random.default_rng(104)
    n = 8000
    fs, rays, p, nrm, face, mat = _scatter_inputs(rng, ca, n)
    gpu.upload_scene(fs)
    sc = orc.Scene.from_flat(fs)
    g = gpu.test_scatter(rays, p, nrm, face, mat, seed=77, strict=True)
    o = orc.scatter(sc, rays, p, nrm, face, mat, seed=77)
    # the number of random draws is part of the contract (metal always draws, dielectric only when not TIR)
    assert np.array_equal(g["draws"], o["draws"])
    assert np.array_equal(g["ok"], o["ok"])
    assert set(np.unique(o["draws"][mat == 2])) == {0, 1}
    assert scaled_ulp(g["ray"][:, 3:], o["ray"][:, 3:], np.abs(o["ray"][:, 3:]).max(axis=1, keepdims=True)).max() <= ULP_TOL
    assert np.array_equal(g["ray"][:, :3], o["ray"][:, :3])
    # lambertian and metal use only +,-,*,/ and sqrt -> identical bits
    lm = mat != 2
    assert np.array_equal(g["ray"][lm], o["ray"][lm])
    # dielectric: identical except where the Schlick term's powf rounding decides the branch
    dm = mat == 2
    same = np.all(g["ray"][dm] == o["ray"][dm], axis=1)
    assert same.mean() > 0.999


def test_camera_ray_parity(gpu, orc, ca):
    rng = np.random.default_rng(105)
    cam = ca.default_camera()
    uv = rng.uniform(0, 1, size=(5000, 2)).astype(np.float32)
    g = gpu.test_camera_ray(cam, uv, seed=5, strict=True)
    o = orc.camera_ray(cam, uv, seed=5)
    assert np.array_equal(g["draws"], o["draws"])
    assert (o["draws"] % 2 == 0).all() and o["draws"].max() > 2
    assert np.array_equal(g["ray"], o["ray"])


def test_texture_parity(gpu, orc, ca):
    rng = np.random.default_rng(106)
    fs = ca.scene_0(textured=True)
    gpu.upload_scene(fs)
    sc = orc.Scene.from_flat(fs)
    n = 6000
    mat = rng.choice([0, 2], size=n).astype(np.int32)      # checker ground, image globe
    uv = rng.uniform(-0.1, 1.1, size=(n, 2)).astype(np.float32)
    uv[:10] = [[0, 0], [1, 1], [1, 0], [0, 1], [0.5, 0.5], [0.999999, 0.999999], [-1, 2], [2, -1], [1e-9, 1e-9], [0.25, 0.75]]
    p = rng.uniform(-20, 20, size=(n, 3)).astype(np.float32)
    g = gpu.test_texture(mat, uv, p, strict=True)
    o = orc.texture(sc, mat, uv, p)
    img = mat == 2
    assert np.array_equal(g[img], o[img])
    # the checker picks a colour by the sign of a product of sines: equal except on a sign tie
    chk = mat == 0
    assert (np.all(g[chk] == o[chk], axis=1)).mean() > 0.9995


def test_noise_texture_parity(gpu, orc, ca):
    """The marble texture (Perlin noise; an extension, the reference only holds a placeholder for it):
    the noise arithmetic is bit-exact against the CPU restatement, the final sine is evaluated in double
    and rounded, so values agree to the last place or two."""
    rng = np.random.default_rng(108)
    fs = ca.scene_perlin(1)
    gpu.upload_scene(fs)
    sc = orc.Scene.from_flat(fs)
    n = 20000
    noisy = np.flatnonzero(fs.materials["tex_type"] == 4)
    mat = rng.choice(noisy, size=n).astype(np.int32)
    p = rng.uniform(-30, 30, size=(n, 3)).astype(np.float32)
    p[:8] = [[0, 0, 0], [1, 2, 3], [-1, -2, -3], [0.5, 0.5, 0.5], [255.5, -255.5, 256], [1e-6, -1e-6, 0], [1000, -1000, 4], [3.999999, 4, 4.000001]]
    uv = np.zeros((n, 2), np.float32)
    g = gpu.test_texture(mat, uv, p, strict=True)
    o = orc.texture(sc, mat, uv, p)
    assert np.isfinite(g).all() and g.min() >= 0.0 and g.max() <= 1.0
    assert (g == o).all(axis=1).mean() > 0.98
    assert np.abs(g - o).max() < 3e-7
    f = gpu.test_texture(mat, uv, p, strict=False)
    assert np.abs(f - o).max() < 2e-4
    # without tables the scene cannot be rendered: a loud error, not black pixels
    bare = ca.FlatScene(fs.objects, fs.materials, fs.images, None)
    gpu.upload_scene(bare)
    acc = gpu.malloc(16 * 16 * 16)
    try:
        with pytest.raises(ca.CasterHipError, match="noise"):
            gpu.render(ca.default_camera(1.0), 16, 16, 0, 1, 5, 1, acc, strict=True)
    finally:
        gpu.free(acc)


def test_fast_mode_functions_close(gpu, orc, ca):
    """The fast arithmetic mode (FMA allowed) stays within a few ULP on well-conditioned inputs."""
    rng = np.random.default_rng(107)
    sph, rays, tr = _sphere_cases(rng, 4000)
    small = sph[:, 3] < 10
    g = gpu.test_hit_sphere(sph[small], rays[small], tr[small], strict=False)
    o = orc.hit_sphere(sph[small], rays[small], tr[small])
    agree = g["hit"] == o["hit"]
    assert agree.mean() > 0.995
    h = agree & (o["hit"] == 1)
    rel = np.abs(g["t"][h] - o["t"][h]) / np.maximum(np.abs(o["t"][h]), 1e-3)
    assert np.median(rel) < 1e-6 and np.quantile(rel, 0.99) < 1e-3


def _grazing_rays(rng, fs, n):
    """Rays aimed at the silhouettes of the scene's spheres (discriminant near zero), rays that start
    on a sphere's surface (self-intersection at t ~ 0), and rays from inside spheres."""
    objs = fs.objects[fs.objects["type"] == 0]
    objs = objs[objs["g"][:, 3] < 50]                       # not the ground / sky spheres
    pick = rng.integers(0, len(objs), size=n)
    c = objs["g"][pick, :3].astype(np.float64)
    r = objs["g"][pick, 3].astype(np.float64)
    o = rng.uniform(-14, 14, size=(n, 3))
    o[:, 1] = np.abs(o[:, 1]) * 0.25 + 0.02
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    kind = rng.integers(0, 4, size=n)
    # 0: tangent-ish (offset ~ r(1 +- 1e-3)); 1: origin on the surface, any direction;
    # 2: origin inside; 3: generic aimed
    toward = c - o
    dist = np.linalg.norm(toward, axis=1, keepdims=True)
    w = np.cross(toward, u)
    w /= np.maximum(np.linalg.norm(w, axis=1, keepdims=True), 1e-9)
    scale = np.where(kind == 0, 1 + rng.normal(size=n) * 1e-3, rng.uniform(0, 1.2, size=n))
    tgt = c + w * (r * scale)[:, None]
    d = (tgt - o) * rng.uniform(0.05, 3.0, size=(n, 1))
    on = kind == 1
    o[on] = c[on] + u[on] * r[on, None]
    d[on] = rng.normal(size=(on.sum(), 3)) * rng.uniform(0.05, 3.0, size=(on.sum(), 1))
    ins = kind == 2
    o[ins] = c[ins] + u[ins] * (r[ins, None] * rng.uniform(0, 0.98, size=(ins.sum(), 1)))
    return np.concatenate([o, d], axis=1).astype(np.float32)


@pytest.mark.parametrize("scene_name", ["rtiow", "stress_2600", "stress_4500", "scene_0", "offset_scene", "mixed_scales"])
def test_production_sphere_loop_parity(gpu, orc, ca, scene_name):
    """hit_list_hit_all through the render kernel's own loop (LDS tiles, candidate filter, packed slots):
    every hit/miss decision, object and t must equal the reference arithmetic, in particular where the
    conservative filter is under the most strain (grazing rays, rays leaving a surface)."""
    rng = np.random.default_rng(108)
    if scene_name == "rtiow":
        fs = ca.scene_rtiow(1)
    elif scene_name == "stress_2600":
        fs = ca.scene_stress(2600, 2)                      # three LDS tiles
    elif scene_name == "stress_4500":
        fs = ca.scene_stress(4500, 2)                      # more than 128 blocks of 32 spheres in the matrix loop
    elif scene_name == "scene_0":
        fs = ca.scene_0()
    elif scene_name == "mixed_scales":
        # radii over a factor of 30 (more would make the large ones "always candidates"), centres spread
        # over 80 units: the half-precision operands of the matrix filter at their widest range
        fs = ca.scene_stress(400, 7)
        objs = fs.objects.copy()
        small = objs["g"][:, 3] < 50
        k = int(small.sum())
        objs["g"][small, 3] = np.exp(rng.uniform(np.log(0.01), np.log(0.3), size=k)).astype(np.float32)
        objs["g"][small, 0] = rng.uniform(-40, 40, size=k).astype(np.float32)
        objs["g"][small, 2] = rng.uniform(-40, 40, size=k).astype(np.float32)
        fs = ca.FlatScene(objs, fs.materials, [])
    else:
        fs = ca.scene_rtiow(5)                             # the whole scene far from the origin
        objs = fs.objects.copy()
        objs["g"][:, 0] += 3000.0
        objs["g"][:, 2] -= 1500.0
        fs = ca.FlatScene(objs, fs.materials, [])
    gpu.upload_scene(fs)
    sc = orc.Scene.from_flat(fs)
    n = 60000
    rays = _grazing_rays(rng, fs, n)
    if scene_name == "offset_scene":
        rays[:, 0] += 3000.0
        rays[:, 2] -= 1500.0
    # rays from very far away, aimed at spheres: beyond the range of the matrix filter's half-precision
    # operands, which must then hand every sphere to the exact test
    far = rng.choice(n - 1, size=400, replace=False)
    tgt = rays[far, :3] + rays[far, 3:] * 0.5
    dirs = rng.normal(size=(400, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dist = np.exp(rng.uniform(np.log(3e3), np.log(3e6), size=(400, 1)))
    rays[far, :3] = (tgt - dirs * dist).astype(np.float32)
    rays[far, 3:] = (dirs * rng.uniform(0.1, 3.0, size=(400, 1))).astype(np.float32)
    rays[-1] = [np.nan, 0, 0, 0, 1, 0]                     # non-finite ray: NaN passes every range test (src/caster.c:107,114)
    g = gpu.test_trace(rays, strict=True)
    # the filter on the vector unit and on the matrix cores hands the same candidates to the same exact test
    gv = gpu.test_trace(rays, strict=True, vector_loop=True)
    for key in ("hit", "mat", "face"):
        assert np.array_equal(g[key], gv[key])
    for key in ("t", "p", "n"):
        assert np.array_equal(g[key], gv[key], equal_nan=True)
    o = orc.hit_list(sc, rays)
    assert np.array_equal(g["hit"], o["hit"])
    h = o["hit"] == 1
    assert 0.3 < h.mean() <= 1.0
    assert np.array_equal(g["mat"][h], o["mat"][h])
    fin = h & np.isfinite(o["t"])
    assert np.array_equal(g["t"][fin], o["t"][fin])
    assert np.array_equal(g["p"][fin], o["p"][fin])
    assert np.array_equal(g["n"][fin], o["n"][fin])
    assert np.array_equal(g["face"][fin], o["face"][fin])
    # fast mode: same decision except within rounding of a silhouette or of t = tmin
    f = gpu.test_trace(rays[:-1], strict=False)
    fv = gpu.test_trace(rays[:-1], strict=False, vector_loop=True)
    assert np.array_equal(f["hit"], fv["hit"]) and np.array_equal(f["mat"], fv["mat"]) and np.array_equal(f["t"], fv["t"], equal_nan=True)
    agree = f["hit"] == o["hit"][:-1]
    assert agree.mean() > 0.985
    both = agree & (o["hit"][:-1] == 1) & (f["mat"] == o["mat"][:-1])
    rel = np.abs(f["t"][both] - o["t"][:-1][both]) / np.maximum(np.abs(o["t"][:-1][both]), 1e-2)
    assert np.quantile(rel, 0.99) < 2e-2


def test_golden_images(gpu, ca):
    """The committed golden images (tests/golden, produced by the CPU restatement): ray counts equal,
    sums within 1e-4 relative, RGBA8 bytes within one code value."""
    import os

    images = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "images.npz"))
    makers = {"scene_0": lambda: ca.scene_0(), "scene_0_textured": lambda: ca.scene_0(textured=True),
              "perlin": lambda: ca.scene_perlin(1),
              "rtiow": lambda: ca.scene_rtiow(1)}
    for name, make in makers.items():
        w, h, spp, depth, seed = (int(v) for v in images[name + "_cfg"])
        gpu.upload_scene(make())
        sums, rays, _ = gpu.render_image(ca.default_camera(w / h), w, h, spp, depth, seed, strict=True)
        ref = images[name + "_sums"]
        assert rays == int(images[name + "_rays"])
        err = np.abs(sums[..., :3].astype(np.float64) - ref[..., :3])
        assert (err <= 1e-4 * np.abs(ref[..., :3]) + 1e-7).all()
        acc = gpu.malloc(sums.nbytes)
        try:
            gpu.upload(acc, sums)
            rgba = gpu.resolve(acc, w, h, spp)
        finally:
            gpu.free(acc)
        assert (np.abs(rgba.astype(int) - images[name + "_rgba"].astype(int)) <= 1).all()
