"""ctypes binding over oracle/liboracle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py, never by the product.  Parity unpinned: see
caster_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

OBJ_DTYPE = np.dtype([("type", "<i4"), ("material", "<i4"), ("g", "<f4", (6,))])
MAT_DTYPE = np.dtype(
    [("type", "<i4"), ("tex_type", "<i4"), ("color0", "<f4", (3,)), ("color1", "<f4", (3,)),
     ("fuzziness", "<f4"), ("n_refract", "<f4"), ("image", "<i4"), ("pad", "<i4")]
)
NOISE_DTYPE = np.dtype([("ranvec", "<f4", (256, 3)), ("perm_x", "u1", (256,)), ("perm_y", "u1", (256,)),
                        ("perm_z", "u1", (256,))])
RNG_COUNTER, RNG_LIBC = 0, 1


class _OImage(C.Structure):
    _fields_ = [("data", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32), ("bpp", C.c_uint32)]


class _OScene(C.Structure):
    _fields_ = [("count", C.c_int32), ("nmats", C.c_int32), ("nimages", C.c_int32), ("pad", C.c_int32),
                ("objs", C.c_void_p), ("mats", C.c_void_p), ("images", C.c_void_p), ("noise", C.c_void_p)]


class _OV3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class _OCam(C.Structure):
    _fields_ = [("f", C.c_float * 19)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        l = C.CDLL(_LIB)
        l.orc_render_rows.restype = C.c_int64
        l.orc_dispatch.restype = C.c_int64
        l.orc_dispatch_rng.restype = C.c_int64
        l.orc_bmp_bytes.restype = C.c_int64
        l.orc_calculate_cam_info.restype = _OCam
        l.orc_calculate_cam_info.argtypes = [_OV3, _OV3, _OV3, C.c_float, C.c_float, C.c_float, C.c_float]
        l.orc_reflectance.restype = C.c_float
        l.orc_reflectance.argtypes = [C.c_float, C.c_float]
        _lib = l
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.reshape(shape) if shape is not None else a


class Scene:
    """Holds numpy copies of the flat tables and the C struct that points at them."""

    def __init__(self, objects, materials, images=(), noise=None):
        self.objects = np.ascontiguousarray(objects, dtype=OBJ_DTYPE)
        self.materials = np.ascontiguousarray(materials, dtype=MAT_DTYPE)
        self.images = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        self._imgs = (_OImage * max(1, len(self.images)))()
        for k, im in enumerate(self.images):
            self._imgs[k] = _OImage(im.ctypes.data, im.shape[1], im.shape[0], im.shape[2])
        self.noise = None if noise is None else np.ascontiguousarray(noise, dtype=NOISE_DTYPE).reshape(1)
        self.c = _OScene(len(self.objects), len(self.materials), len(self.images), 0,
                         self.objects.ctypes.data, self.materials.ctypes.data, C.addressof(self._imgs),
                         None if self.noise is None else self.noise.ctypes.data)

    @classmethod
    def from_flat(cls, fs):
        return cls(fs.objects, fs.materials, fs.images, getattr(fs, "noise", None))


def calculate_cam_info(pos, look_at, up, fov, aspect, aperture, focus) -> np.ndarray:
    ci = lib().orc_calculate_cam_info(_OV3(*pos), _OV3(*look_at), _OV3(*up), fov, aspect, aperture, focus)
    return np.array(list(ci.f), dtype=np.float32)


def default_camera(aspect=16.0 / 9.0) -> np.ndarray:
    return calculate_cam_info((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, np.float32(aspect), 0.1, 10.0)


def render(scene: Scene, cam, width, height, spp, max_depth, seed, sample_begin=0, rng_mode=RNG_COUNTER,
           rows=None):
    """Returns (sums [h, w, 4] float32 with row j = 0 at the image bottom, rays)."""
    cam = _f32(cam, (19,))
    sums = np.zeros((height, width, 4), dtype=np.float32)
    j0, j1 = rows if rows is not None else (0, height)
    rays = lib().orc_render_rows(C.byref(scene.c), _p(cam), C.c_int(width), C.c_int(height), C.c_int(j0), C.c_int(j1),
                                 C.c_int(sample_begin), C.c_int(spp), C.c_int(max_depth), C.c_uint32(seed),
                                 C.c_int(rng_mode), _p(sums))
    return sums, int(rays)


def dispatch(scene: Scene, cam, width, height, spp, max_depth, seed, threads, blocks=None, rng_mode=RNG_COUNTER):
    """pthread port of the reference dispatcher; returns (sums, rgba [h, w, 4] uint8, rays).
    rng_mode RNG_LIBC: every thread draws from the one libc rand(), as the reference ships."""
    cam = _f32(cam, (19,))
    sums = np.zeros((height, width, 4), dtype=np.float32)
    rgba = np.zeros((height, width, 4), dtype=np.uint8)
    rays = lib().orc_dispatch_rng(C.byref(scene.c), _p(cam), C.c_int(width), C.c_int(height), C.c_int(spp), C.c_int(max_depth),
                                  C.c_uint32(seed), C.c_int(threads), C.c_int(blocks if blocks else threads), C.c_int(rng_mode),
                                  _p(sums), _p(rgba))
    return sums, rgba, int(rays)


def resolve(sums, spp) -> np.ndarray:
    sums = _f32(sums)
    h, w = sums.shape[:2]
    rgba = np.zeros((h, w, 4), dtype=np.uint8)
    lib().orc_resolve(_p(sums), C.c_int(w), C.c_int(h), C.c_int(spp), _p(rgba))
    return rgba


def bmp_bytes(rgba) -> bytes:
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = rgba.shape[:2]
    row = w * 3
    out = np.zeros(54 + (row + (4 - row % 4) % 4) * h, dtype=np.uint8)
    n = lib().orc_bmp_bytes(_p(rgba), C.c_int(w), C.c_int(h), _p(out))
    return out[:n].tobytes()


def hit_sphere(sph4, ray6, trange2):
    sph4, ray6, trange2 = _f32(sph4, (-1, 4)), _f32(ray6, (-1, 6)), _f32(trange2, (-1, 2))
    n = len(sph4)
    hit, face = np.zeros(n, np.int32), np.zeros(n, np.int32)
    t, p3, n3, uv = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32)
    lib().orc_batch_hit_sphere(C.c_int(n), _p(sph4), _p(ray6), _p(trange2), _p(hit), _p(t), _p(p3), _p(n3), _p(face), _p(uv))
    return dict(hit=hit, t=t, p=p3, n=n3, face=face, uv=uv)


def hit_xy_rect(rect5, ray6, trange2):
    rect5, ray6, trange2 = _f32(rect5, (-1, 5)), _f32(ray6, (-1, 6)), _f32(trange2, (-1, 2))
    n = len(rect5)
    hit, face = np.zeros(n, np.int32), np.zeros(n, np.int32)
    t, p3, n3, uv = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32)
    lib().orc_batch_hit_xy_rect(C.c_int(n), _p(rect5), _p(ray6), _p(trange2), _p(hit), _p(t), _p(p3), _p(n3), _p(face), _p(uv))
    return dict(hit=hit, t=t, p=p3, n=n3, face=face, uv=uv)


def hit_list(scene: Scene, ray6):
    ray6 = _f32(ray6, (-1, 6))
    n = len(ray6)
    hit, face, mat = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    t, p3, n3 = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    lib().orc_batch_hit_list(C.byref(scene.c), C.c_int(n), _p(ray6), _p(hit), _p(t), _p(p3), _p(n3), _p(face), _p(mat))
    return dict(hit=hit, t=t, p=p3, n=n3, face=face, mat=mat)


def scatter(scene: Scene, ray6, p3, n3, face, mat, seed):
    ray6, p3, n3 = _f32(ray6, (-1, 6)), _f32(p3, (-1, 3)), _f32(n3, (-1, 3))
    face, mat = np.ascontiguousarray(face, np.int32), np.ascontiguousarray(mat, np.int32)
    n = len(ray6)
    ok, draws, out = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros((n, 6), np.float32)
    lib().orc_batch_scatter(C.byref(scene.c), C.c_int(n), _p(ray6), _p(p3), _p(n3), _p(face), _p(mat), C.c_uint32(seed),
                            _p(ok), _p(out), _p(draws))
    return dict(ok=ok, ray=out, draws=draws)


def camera_ray(cam, uv2, seed):
    cam, uv2 = _f32(cam, (19,)), _f32(uv2, (-1, 2))
    n = len(uv2)
    out, draws = np.zeros((n, 6), np.float32), np.zeros(n, np.int32)
    lib().orc_batch_camera_ray(_p(cam), C.c_int(n), _p(uv2), C.c_uint32(seed), _p(out), _p(draws))
    return dict(ray=out, draws=draws)


def texture(scene: Scene, mat, uv2, p3):
    mat, uv2, p3 = np.ascontiguousarray(mat, np.int32), _f32(uv2, (-1, 2)), _f32(p3, (-1, 3))
    n = len(mat)
    rgb = np.zeros((n, 3), np.float32)
    lib().orc_batch_texture(C.byref(scene.c), C.c_int(n), _p(mat), _p(uv2), _p(p3), _p(rgb))
    return rgb


def noise_value(noise, p3):
    """orc_noise_value for an array of points (pure-Python loop over a C call: small inputs only)."""
    noise = np.ascontiguousarray(noise, dtype=NOISE_DTYPE).reshape(1)
    f = lib().orc_noise_value
    f.restype = C.c_float
    f.argtypes = [C.c_void_p, _OV3]
    return np.array([f(noise.ctypes.data, _OV3(*map(float, p))) for p in _f32(p3, (-1, 3))], dtype=np.float32)


def noise_turb(noise, p3, depth=7):
    noise = np.ascontiguousarray(noise, dtype=NOISE_DTYPE).reshape(1)
    f = lib().orc_noise_turb
    f.restype = C.c_float
    f.argtypes = [C.c_void_p, _OV3, C.c_int]
    return np.array([f(noise.ctypes.data, _OV3(*map(float, p)), depth) for p in _f32(p3, (-1, 3))], dtype=np.float32)


def rng(seed, pixel, sample, n):
    out = np.zeros(n, np.uint32)
    lib().orc_batch_rng(C.c_uint32(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_int(n), _p(out))
    return out


MATH_REFLECTANCE, MATH_REFRACT, MATH_REFLECT, MATH_UV_SPHERE, MATH_NEAR_ZERO, MATH_TO_UNIT = range(6)


def math(fn, in8):
    """orc_batch_math: item k reads in8[k, :8] and returns out[k, :3] (see caster_oracle.c)."""
    in8 = _f32(in8, (-1, 8))
    out = np.zeros((len(in8), 3), np.float32)
    lib().orc_batch_math(C.c_int(fn), C.c_int(len(in8)), _p(in8), _p(out))
    return out


def reflectance(cosine, ratio) -> float:
    return float(lib().orc_reflectance(C.c_float(cosine), C.c_float(ratio)))
