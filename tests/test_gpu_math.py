"""Known-answer vectors for the small vector and optics functions of src/common.c and for
get_uv_coord of src/material.c, HIP (strict mode) against the CPU restatement through the C ABI.

Bar: 2 ULP-scaled FP32.  reflect, refract, to_unit and near_zero use only +, -, *, / and sqrt and
are expected bit-identical; reflectance (powf) and the sphere uv (atan2f, acosf) go through libm on
the CPU and through double-then-round on the GPU and may differ in the last place.
"""
import numpy as np
import pytest

from conftest import scaled_ulp, ulp_distance

pytestmark = pytest.mark.gpu


def _unit(v):
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)


def _in8(*cols):
    n = len(cols[0])
    a = np.zeros((n, 8), np.float32)
    k = 0
    for c in cols:
        c = np.asarray(c, np.float32).reshape(n, -1)
        a[:, k:k + c.shape[1]] = c
        k += c.shape[1]
    return a


def test_reflectance_table(gpu, orc, ca):
    """Schlick's approximation (src/common.c:170-175) over the whole cosine range for the ratios the
    scenes use (1.5 and 1/1.5) and a spread of others, the end points included."""
    cos = np.concatenate([np.linspace(0, 1, 2001), [0.0, 1.0, 1e-7, 0.9999999, 0.5]]).astype(np.float32)
    ratios = np.array([1.5, np.float32(1.0) / np.float32(1.5), 1.0, 1.33, 2.4, 0.41], np.float32)
    cc, rr = np.meshgrid(cos, ratios)
    a = _in8(cc.ravel(), rr.ravel())
    g = gpu.test_math(ca.MATH_REFLECTANCE, a, strict=True)[:, 0]
    o = orc.math(orc.MATH_REFLECTANCE, a)[:, 0]
    assert np.isfinite(g).all() and g.min() >= 0.0 and g.max() <= 1.0
    assert ulp_distance(g, o).max() <= 2
    assert (g == o).mean() > 0.98
    # the fast mode multiplies in float: a few ulp of the result
    f = gpu.test_math(ca.MATH_REFLECTANCE, a, strict=False)[:, 0]
    assert scaled_ulp(f, o, np.maximum(np.abs(o), 1e-3)).max() <= 16


def test_refract_and_reflect_tables(gpu, orc, ca):
    """v3_f32_refract with the cos_theta and ratio ray_scatter_dielectric hands it (src/material.c:22-41),
    and v3_f32_reflect, on random unit directions against random unit normals, front and back faces,
    grazing incidence and normal incidence included."""
    rng = np.random.default_rng(301)
    n = 20000
    nrm = _unit(rng.normal(size=(n, 3)))
    d = _unit(rng.normal(size=(n, 3)))
    flip = (d * nrm).sum(axis=1) > 0
    d[flip] = -d[flip]
    graze = rng.random(n) < 0.25
    d[graze] = _unit(d[graze] - 0.999 * (d[graze] * nrm[graze]).sum(axis=1, keepdims=True) * nrm[graze])
    d[:50] = -nrm[:50]                                                    # normal incidence
    ratio = np.where(rng.random(n) < 0.5, np.float32(1.0) / np.float32(1.5), np.float32(1.5)).astype(np.float32)
    cos_theta = np.minimum((-d * nrm).sum(axis=1, dtype=np.float32), np.float32(1.0)).astype(np.float32)
    a = _in8(d, nrm, ratio, cos_theta)
    g = gpu.test_math(ca.MATH_REFRACT, a, strict=True)
    o = orc.math(orc.MATH_REFRACT, a)
    assert np.array_equal(g, o)
    v = (d * rng.uniform(0.05, 20.0, size=(n, 1))).astype(np.float32)   # reflect takes any length
    a = _in8(v, nrm)
    g = gpu.test_math(ca.MATH_REFLECT, a, strict=True)
    o = orc.math(orc.MATH_REFLECT, a)
    assert np.array_equal(g, o)
    f = gpu.test_math(ca.MATH_REFLECT, a, strict=False)
    assert scaled_ulp(f, o, np.abs(v).max(axis=1, keepdims=True)).max() <= 16     # FMA contraction inside the dot product


def test_uv_coord_of_sphere_normals(gpu, orc, ca):
    """get_uv_coord (src/material.c:45-56) on unit normals: a dense sweep, the poles (acos at +-1), the
    +-pi seam of atan2 (z = +-0, x < 0) and the axis directions."""
    rng = np.random.default_rng(302)
    p = _unit(rng.normal(size=(30000, 3)))
    special = np.array([[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1],
                        [-1, 0, 1e-30], [-1, 0, -1e-30], [-1, 0, 0.0], [-1, 0, -0.0], [-1, 1e-8, 1e-8],
                        [1e-20, 1, 1e-20], [1e-20, -1, -1e-20], [0.6, 0.8, 0.0], [0.0, 0.6, 0.8]], np.float32)
    # normals as hit_sphere produces them are only unit to rounding: |y| may exceed 1 by an ulp (acos -> NaN in C too)
    near = p[:2000].copy()
    near[:, 1] = np.sign(near[:, 1]) * (1.0 - np.float32(2.0) ** -rng.integers(10, 24, size=2000)).astype(np.float32)
    pts = np.concatenate([special, near, p]).astype(np.float32)
    a = _in8(pts)
    g = gpu.test_math(ca.MATH_UV_SPHERE, a, strict=True)[:, :2]
    o = orc.math(orc.MATH_UV_SPHERE, a)[:, :2]
    assert np.isfinite(o).all()
    assert scaled_ulp(g, o, 1.0).max() <= 2.0
    assert (g == o).all(axis=1).mean() > 0.8                 # two libm calls and a division each: most are the same bits
    assert g.min() >= 0.0 and g.max() <= 1.0
    # the seam: z = +0 and z = -0 land on opposite ends of u, as atan2f's sign rule says
    seam = orc.math(orc.MATH_UV_SPHERE, _in8(np.array([[-1, 0, 0.0], [-1, 0, -0.0]], np.float32)))[:, 0]
    gs = gpu.test_math(ca.MATH_UV_SPHERE, _in8(np.array([[-1, 0, 0.0], [-1, 0, -0.0]], np.float32)), strict=True)[:, 0]
    assert np.array_equal(seam, gs) and abs(float(seam[0]) - float(seam[1])) > 0.99
    f = gpu.test_math(ca.MATH_UV_SPHERE, a, strict=False)[:, :2]
    du = np.abs(f[:, 0] - o[:, 0])
    assert np.minimum(du, 1.0 - du).max() < 1e-5 and np.abs(f[:, 1] - o[:, 1]).max() < 2e-4


def test_near_zero_and_to_unit(gpu, orc, ca):
    """v3_f32_near_zero's threshold of 1e-8 per component (src/common.c:152-156: the compare runs in
    double on fabs of the float) and v3_f32_to_unit's three quotients (src/common.c:89-97)."""
    rng = np.random.default_rng(303)
    s = np.float32(1e-8)
    edge = np.array([np.nextafter(s, np.float32(0)), s, np.nextafter(s, np.float32(1)), 0.0, -0.0, 1e-9, -1e-9, 1e-7, -1e-7,
                     1e-45, -1e-45], np.float32)
    xx, yy, zz = np.meshgrid(edge, edge, edge)
    v = np.stack([xx.ravel(), yy.ravel(), zz.ravel()], axis=1)
    a = _in8(v)
    g = gpu.test_math(ca.MATH_NEAR_ZERO, a, strict=True)[:, 0]
    o = orc.math(orc.MATH_NEAR_ZERO, a)[:, 0]
    assert np.array_equal(g, o) and 0.1 < o.mean() < 0.9
    assert np.array_equal(gpu.test_math(ca.MATH_NEAR_ZERO, a, strict=False)[:, 0], o)
    n = 30000
    w = (rng.normal(size=(n, 3)) * np.exp(rng.uniform(np.log(1e-18), np.log(1e18), size=(n, 1)))).astype(np.float32)
    w[:100, 1:] = 0.0                                      # axis-aligned
    w[100:200] *= np.float32(1e-20)                        # squares underflow: outside the shared-reciprocal range
    a = _in8(w)
    g = gpu.test_math(ca.MATH_TO_UNIT, a, strict=True)
    o = orc.math(orc.MATH_TO_UNIT, a)
    assert np.array_equal(g, o, equal_nan=True)
