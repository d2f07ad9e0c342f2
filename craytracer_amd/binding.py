"""ctypes binding over libcaster_hip.so and libcraytracer_host.so."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_DIR = os.path.join(_HERE, "lib")

STRICT = 1
VECTOR_LOOP = 2      # evaluate the sphere filter on the vector unit (A/B against the matrix cores)
NO_TILE_LISTS = 4    # camera rays meet every sphere instead of their tile's list (tests)

OBJECT_DTYPE = np.dtype([("type", "<i4"), ("material", "<i4"), ("g", "<f4", (6,))])
MATERIAL_DTYPE = np.dtype(
    [
        ("type", "<i4"),
        ("tex_type", "<i4"),
        ("color0", "<f4", (3,)),
        ("color1", "<f4", (3,)),
        ("fuzziness", "<f4"),
        ("n_refract", "<f4"),
        ("image", "<i4"),
        ("pad", "<i4"),
    ]
)
NOISE_DTYPE = np.dtype([("ranvec", "<f4", (256, 3)), ("perm_x", "u1", (256,)), ("perm_y", "u1", (256,)),
                        ("perm_z", "u1", (256,))])      # caster_noise
assert OBJECT_DTYPE.itemsize == 32 and MATERIAL_DTYPE.itemsize == 48


class CasterHipError(RuntimeError):
    pass


class _RenderInfo(C.Structure):      # caster_render_info
    _fields_ = [("loop_kind", C.c_int32), ("mfma_blocks", C.c_int32), ("n_spheres", C.c_int32), ("kernel_ms", C.c_float),
                ("rays", C.c_uint64), ("wave_trips", C.c_uint64), ("primary_batches", C.c_uint64)]


NOALLOC_MAX_PIXELS = 1 << 20         # CASTER_HIP_NOALLOC_MAX_PIXELS
MATH_REFLECTANCE, MATH_REFRACT, MATH_REFLECT, MATH_UV_SPHERE, MATH_NEAR_ZERO, MATH_TO_UNIT = range(6)
LOOP_NAMES = {0: "vector", 1: "vector_tiled", 2: "matrix", 3: "matrix_large"}


class _Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32), ("bpp", C.c_uint32)]


def lib_paths():
    return {
        "caster": os.path.join(_LIB_DIR, "libcaster_hip.so"),
        "host": os.path.join(_LIB_DIR, "libcraytracer_host.so"),
    }


_caster = None
_host = None


def load_caster_lib():
    """Loads libcaster_hip.so; raises if it has not been built (no fallback)."""
    global _caster
    if _caster is None:
        path = lib_paths()["caster"]
        if not os.path.exists(path):
            raise CasterHipError(
                f"{path} is missing: build it with `make -C craytracer_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback"
            )
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        lib.caster_hip_last_error.restype = C.c_char_p
        _caster = lib
    return _caster


def load_host_lib():
    global _host
    if _host is None:
        load_caster_lib()
        path = lib_paths()["host"]
        if not os.path.exists(path):
            raise CasterHipError(f"{path} is missing: build it with `make -C craytracer_amd/host`")
        lib = C.CDLL(path)
        for name in ("crt_scene_0", "crt_scene_rtiow", "crt_scene_stress", "crt_flatten", "crt_make_test_texture",
                     "new_image_buffer", "crt_flat_objects", "crt_flat_materials", "crt_flat_images",
                     "crt_flat_noise", "crt_scene_perlin", "new_noise_perlin"):
            getattr(lib, name).restype = C.c_void_p
        lib.crt_bmp_encode.restype = C.c_long
        _host = lib
    return _host


def _check(rc: int, what: str):
    if rc != 0:
        msg = load_caster_lib().caster_hip_last_error().decode("utf-8", "replace")
        raise CasterHipError(f"{what} failed (status {rc}): {msg}")


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


@dataclass
class FlatScene:
    """A hit_list flattened into the tables libcaster_hip uploads."""

    objects: np.ndarray                      # OBJECT_DTYPE, list order
    materials: np.ndarray                    # MATERIAL_DTYPE
    images: List[np.ndarray] = field(default_factory=list)   # uint8 [h, w, bpp]
    noise: Optional[np.ndarray] = None       # NOISE_DTYPE, the tables the TTYPE_NOISE materials share

    @property
    def n_spheres(self) -> int:
        return int((self.objects["type"] == 0).sum())

    @property
    def n_rects(self) -> int:
        return int((self.objects["type"] == 1).sum())


# ----------------------------------------------------------------- host side --
class _ImgBuffer(C.Structure):
    _fields_ = [("data", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32), ("bpp", C.c_uint8)]


class _HitList(C.Structure):
    _fields_ = [("head", C.c_void_p), ("types", C.c_void_p), ("count", C.c_int32)]


class _Scene(C.Structure):
    _fields_ = [
        ("world", _HitList),
        ("spheres", C.c_void_p), ("n_spheres", C.c_int32),
        ("rects", C.c_void_p), ("n_rects", C.c_int32),
        ("materials", C.c_void_p), ("n_materials", C.c_int32),
        ("texture_image", C.c_void_p),
        ("noise", C.c_void_p),
    ]


def _flatten_scene_ptr(scene_ptr) -> FlatScene:
    host = load_host_lib()
    sc = C.cast(scene_ptr, C.POINTER(_Scene)).contents
    fs = host.crt_flatten(C.byref(sc.world))
    if not fs:
        raise CasterHipError("crt_flatten failed")
    try:
        n_obj, n_mat, n_img = C.c_int(), C.c_int(), C.c_int()
        host.crt_flat_counts(C.c_void_p(fs), C.byref(n_obj), C.byref(n_mat), C.byref(n_img))
        objs = np.frombuffer(
            C.string_at(host.crt_flat_objects(C.c_void_p(fs)), n_obj.value * OBJECT_DTYPE.itemsize), dtype=OBJECT_DTYPE
        ).copy()
        mats = np.frombuffer(
            C.string_at(host.crt_flat_materials(C.c_void_p(fs)), n_mat.value * MATERIAL_DTYPE.itemsize),
            dtype=MATERIAL_DTYPE,
        ).copy()
        images = []
        if n_img.value:
            arr = C.cast(host.crt_flat_images(C.c_void_p(fs)), C.POINTER(_Image))
            for k in range(n_img.value):
                im = arr[k]
                raw = C.string_at(im.data, im.w * im.h * im.bpp)
                images.append(np.frombuffer(raw, dtype=np.uint8).reshape(im.h, im.w, im.bpp).copy())
        noise = None
        pn = host.crt_flat_noise(C.c_void_p(fs))
        if pn:
            noise = np.frombuffer(C.string_at(pn, NOISE_DTYPE.itemsize), dtype=NOISE_DTYPE).copy().reshape(())
        return FlatScene(objs, mats, images, noise)
    finally:
        host.crt_flat_free(C.c_void_p(fs))


def make_test_texture(w: int = 256, h: int = 128) -> np.ndarray:
    host = load_host_lib()
    p = host.crt_make_test_texture(C.c_uint32(w), C.c_uint32(h))
    buf = C.cast(p, C.POINTER(_ImgBuffer)).contents
    out = np.frombuffer(C.string_at(buf.data, w * h * 3), dtype=np.uint8).reshape(h, w, 3).copy()
    host.free_image_buffer(C.c_void_p(p))
    return out


def scene_0(textured: bool = False) -> FlatScene:
    """src/main.c:14-55; the earth texture is generated (the reference's bitmap is not in its repository)."""
    host = load_host_lib()
    earth = host.crt_make_test_texture(C.c_uint32(256), C.c_uint32(128)) if textured else None
    p = host.crt_scene_0(C.c_void_p(earth))
    try:
        return _flatten_scene_ptr(p)
    finally:
        host.crt_scene_free(C.c_void_p(p))


def scene_rtiow(seed: int = 1) -> FlatScene:
    host = load_host_lib()
    p = host.crt_scene_rtiow(C.c_uint32(seed))
    try:
        return _flatten_scene_ptr(p)
    finally:
        host.crt_scene_free(C.c_void_p(p))


def scene_perlin(seed: int = 1) -> FlatScene:
    """The RTIOW layout with marble (Perlin noise) textures on the ground and the diffuse spheres."""
    host = load_host_lib()
    p = host.crt_scene_perlin(C.c_uint32(seed))
    try:
        return _flatten_scene_ptr(p)
    finally:
        host.crt_scene_free(C.c_void_p(p))


def make_noise(seed: int = 1) -> np.ndarray:
    """new_noise_perlin(seed) as a NOISE_DTYPE record."""
    host = load_host_lib()
    p = host.new_noise_perlin(C.c_uint32(seed))
    try:
        return np.frombuffer(C.string_at(p, NOISE_DTYPE.itemsize), dtype=NOISE_DTYPE).copy().reshape(())
    finally:
        host.free_noise_perlin(C.c_void_p(p))


def scene_stress(n: int, seed: int = 2) -> FlatScene:
    host = load_host_lib()
    p = host.crt_scene_stress(C.c_int32(n), C.c_uint32(seed))
    try:
        return _flatten_scene_ptr(p)
    finally:
        host.crt_scene_free(C.c_void_p(p))


class _V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class _CamInfo(C.Structure):
    _fields_ = [("f", C.c_float * 19)]


def calculate_cam_info(pos, look_at, up, fov, aspect, aperture, focus) -> np.ndarray:
    """src/camera.c:3-30 through the C host; returns the 19 floats of cam_info."""
    host = load_host_lib()
    host.calculate_cam_info.restype = _CamInfo
    host.calculate_cam_info.argtypes = [_V3, _V3, _V3, C.c_float, C.c_float, C.c_float, C.c_float]
    ci = host.calculate_cam_info(_V3(*pos), _V3(*look_at), _V3(*up), fov, aspect, aperture, focus)
    return np.array(list(ci.f), dtype=np.float32)


def default_camera(aspect: float = 16.0 / 9.0) -> np.ndarray:
    """The camera of src/main.c:71-78."""
    return calculate_cam_info((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, np.float32(aspect), 0.1, 10.0)


def bmp_encode(rgba: np.ndarray) -> bytes:
    """Bytes the host's write_img_buffer_to_file() produces for an RGBA8 [h, w, 4] image."""
    host = load_host_lib()
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = rgba.shape[:2]
    buf = _ImgBuffer(rgba.ctypes.data, w, h, 4)
    row = w * 3
    out = np.zeros(54 + (row + (4 - row % 4) % 4) * h, dtype=np.uint8)
    n = host.crt_bmp_encode(C.byref(buf), _ptr(out))
    return out[:n].tobytes()


# ------------------------------------------------------------------ GPU side --
class CasterHip:
    """One libcaster_hip context (one GPU)."""

    def __init__(self, device: int = 0):
        self._lib = load_caster_lib()
        self._ctx = C.c_void_p()
        _check(self._lib.caster_hip_create(C.c_int(device), C.byref(self._ctx)), "caster_hip_create")
        self.device = device
        self._keep = None

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.caster_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- setup
    def set_stream(self, stream_handle: int):
        _check(self._lib.caster_hip_set_stream(self._ctx, C.c_void_p(stream_handle)), "caster_hip_set_stream")

    def set_chunk(self, samples_per_unit: int):
        _check(self._lib.caster_hip_set_chunk(self._ctx, C.c_int(samples_per_unit)), "caster_hip_set_chunk")

    def upload_scene(self, fs: FlatScene):
        objs = np.ascontiguousarray(fs.objects, dtype=OBJECT_DTYPE)
        mats = np.ascontiguousarray(fs.materials, dtype=MATERIAL_DTYPE)
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in fs.images]
        arr = (_Image * max(1, len(imgs)))()
        for k, im in enumerate(imgs):
            arr[k] = _Image(im.ctypes.data, im.shape[1], im.shape[0], im.shape[2])
        _check(
            self._lib.caster_hip_upload_scene(
                self._ctx, C.c_int(len(objs)), _ptr(objs), C.c_int(len(mats)), _ptr(mats), C.c_int(len(imgs)), arr
            ),
            "caster_hip_upload_scene",
        )
        if fs.noise is not None:
            noise = np.ascontiguousarray(fs.noise, dtype=NOISE_DTYPE).reshape(1)
            _check(self._lib.caster_hip_set_noise(self._ctx, _ptr(noise)), "caster_hip_set_noise")

    # -- device memory
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _check(self._lib.caster_hip_malloc(self._ctx, C.c_size_t(nbytes), C.byref(p)), "caster_hip_malloc")
        return p.value

    def free(self, ptr: int):
        _check(self._lib.caster_hip_free(self._ctx, C.c_void_p(ptr)), "caster_hip_free")

    def memset_zero(self, ptr: int, nbytes: int):
        _check(self._lib.caster_hip_memset_zero(self._ctx, C.c_void_p(ptr), C.c_size_t(nbytes)), "caster_hip_memset_zero")

    def download(self, ptr: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        _check(self._lib.caster_hip_download(self._ctx, _ptr(out), C.c_void_p(ptr), C.c_size_t(out.nbytes)), "caster_hip_download")
        return out

    def upload(self, ptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        _check(self._lib.caster_hip_upload(self._ctx, C.c_void_p(ptr), _ptr(arr), C.c_size_t(arr.nbytes)), "caster_hip_upload")

    def synchronize(self):
        _check(self._lib.caster_hip_synchronize(self._ctx), "caster_hip_synchronize")

    def mem_info(self):
        """(free bytes, total bytes) of the context's GPU."""
        free, total = C.c_size_t(), C.c_size_t()
        _check(self._lib.caster_hip_mem_info(self._ctx, C.byref(free), C.byref(total)), "caster_hip_mem_info")
        return free.value, total.value

    def add_from_device(self, dst_ptr: int, src_device: int, src_ptr: int, n_floats: int):
        """dst += src where src lives on another context's device (its work must be complete)."""
        _check(
            self._lib.caster_hip_add_from_device(self._ctx, C.c_void_p(dst_ptr), C.c_int(src_device), C.c_void_p(src_ptr), C.c_size_t(n_floats)),
            "caster_hip_add_from_device",
        )

    # -- the hot path
    def render(self, cam, width, height, sample_begin, sample_count, max_depth, seed, accum_ptr: int, strict=False, vector_loop=False,
               tile_lists=True):
        cam = _f32(cam, (19,))
        _check(
            self._lib.caster_hip_render(
                self._ctx, _ptr(cam), C.c_int(width), C.c_int(height), C.c_int(sample_begin), C.c_int(sample_count),
                C.c_int(max_depth), C.c_uint32(seed),
                C.c_uint32((STRICT if strict else 0) | (VECTOR_LOOP if vector_loop else 0) | (0 if tile_lists else NO_TILE_LISTS)), C.c_void_p(accum_ptr),
            ),
            "caster_hip_render",
        )

    def render_noalloc(self, cam, width, height, sample_begin, sample_count, max_depth, seed, accum_ptr: int, strict=False, vector_loop=False):
        """caster_hip_render_noalloc: the same render out of the library's static scratch (no device allocation)."""
        cam = _f32(cam, (19,))
        _check(
            self._lib.caster_hip_render_noalloc(
                self._ctx, _ptr(cam), C.c_int(width), C.c_int(height), C.c_int(sample_begin), C.c_int(sample_count),
                C.c_int(max_depth), C.c_uint32(seed), C.c_uint32((STRICT if strict else 0) | (VECTOR_LOOP if vector_loop else 0)), C.c_void_p(accum_ptr),
            ),
            "caster_hip_render_noalloc",
        )

    def render_info(self) -> dict:
        """caster_hip_render_info of the last render: loop kind, blocks, wave trips, rays, kernel ms."""
        info = _RenderInfo()
        _check(self._lib.caster_hip_render_info(self._ctx, C.byref(info)), "caster_hip_render_info")
        return dict(loop_kind=info.loop_kind, loop=LOOP_NAMES.get(info.loop_kind, "none"), mfma_blocks=info.mfma_blocks,
                    n_spheres=info.n_spheres, kernel_ms=info.kernel_ms, rays=info.rays, wave_trips=info.wave_trips,
                    primary_batches=info.primary_batches)

    def render_stats(self):
        rays, ms = C.c_uint64(), C.c_float()
        _check(self._lib.caster_hip_render_stats(self._ctx, C.byref(rays), C.byref(ms)), "caster_hip_render_stats")
        return rays.value, ms.value

    def resolve(self, accum_ptr: int, width: int, height: int, spp: int) -> np.ndarray:
        out = np.empty((height, width, 4), dtype=np.uint8)
        _check(
            self._lib.caster_hip_resolve(self._ctx, C.c_void_p(accum_ptr), C.c_int(width), C.c_int(height), C.c_int(spp), _ptr(out)),
            "caster_hip_resolve",
        )
        return out

    def resolve_dev(self, accum_ptr: int, width: int, height: int, spp: int, rgba_ptr: int):
        _check(
            self._lib.caster_hip_resolve_dev(self._ctx, C.c_void_p(accum_ptr), C.c_int(width), C.c_int(height), C.c_int(spp), C.c_void_p(rgba_ptr)),
            "caster_hip_resolve_dev",
        )

    def render_image(self, cam, width, height, spp, max_depth, seed, strict=False, sample_begin=0, vector_loop=False, tile_lists=True):
        """Convenience: allocate, render, return (sums [h, w, 4] float32 with j = 0 at the bottom, rays, kernel ms)."""
        nbytes = width * height * 16
        acc = self.malloc(nbytes)
        try:
            self.memset_zero(acc, nbytes)
            self.render(cam, width, height, sample_begin, spp, max_depth, seed, acc, strict=strict, vector_loop=vector_loop,
                        tile_lists=tile_lists)
            rays, ms = self.render_stats()
            sums = self.download(acc, (height, width, 4), np.float32)
        finally:
            self.free(acc)
        return sums, rays, ms

    # -- per-function entry points
    def test_hit_sphere(self, sph4, ray6, trange2, strict=True):
        sph4, ray6, trange2 = _f32(sph4, (-1, 4)), _f32(ray6, (-1, 6)), _f32(trange2, (-1, 2))
        n = len(sph4)
        hit, face = np.zeros(n, np.int32), np.zeros(n, np.int32)
        t, p3, n3 = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        _check(
            self._lib.caster_hip_test_hit_sphere(self._ctx, C.c_uint32(STRICT if strict else 0), C.c_int(n), _ptr(sph4), _ptr(ray6),
                                                 _ptr(trange2), _ptr(hit), _ptr(t), _ptr(p3), _ptr(n3), _ptr(face)),
            "caster_hip_test_hit_sphere",
        )
        return dict(hit=hit, t=t, p=p3, n=n3, face=face)

    def test_hit_xy_rect(self, rect5, ray6, trange2, strict=True):
        rect5, ray6, trange2 = _f32(rect5, (-1, 5)), _f32(ray6, (-1, 6)), _f32(trange2, (-1, 2))
        n = len(rect5)
        hit, face = np.zeros(n, np.int32), np.zeros(n, np.int32)
        t, p3, n3, uv = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32)
        _check(
            self._lib.caster_hip_test_hit_xy_rect(self._ctx, C.c_uint32(STRICT if strict else 0), C.c_int(n), _ptr(rect5), _ptr(ray6),
                                                  _ptr(trange2), _ptr(hit), _ptr(t), _ptr(p3), _ptr(n3), _ptr(face), _ptr(uv)),
            "caster_hip_test_hit_xy_rect",
        )
        return dict(hit=hit, t=t, p=p3, n=n3, face=face, uv=uv)

    def test_hit_list(self, ray6, strict=True):
        ray6 = _f32(ray6, (-1, 6))
        n = len(ray6)
        hit, face, mat = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        t, p3, n3 = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        _check(
            self._lib.caster_hip_test_hit_list(self._ctx, C.c_uint32(STRICT if strict else 0), C.c_int(n), _ptr(ray6), _ptr(hit),
                                               _ptr(t), _ptr(p3), _ptr(n3), _ptr(face), _ptr(mat)),
            "caster_hip_test_hit_list",
        )
        return dict(hit=hit, t=t, p=p3, n=n3, face=face, mat=mat)

    def test_trace(self, ray6, strict=True, vector_loop=False):
        ray6 = _f32(ray6, (-1, 6))
        n = len(ray6)
        hit, face, mat = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        t, p3, n3 = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        _check(
            self._lib.caster_hip_test_trace(self._ctx, C.c_uint32((STRICT if strict else 0) | (VECTOR_LOOP if vector_loop else 0)), C.c_int(n), _ptr(ray6), _ptr(hit),
                                            _ptr(t), _ptr(p3), _ptr(n3), _ptr(face), _ptr(mat)),
            "caster_hip_test_trace",
        )
        return dict(hit=hit, t=t, p=p3, n=n3, face=face, mat=mat)

    def test_scatter(self, ray6, p3, n3, face, mat, seed, strict=True):
        ray6, p3, n3 = _f32(ray6, (-1, 6)), _f32(p3, (-1, 3)), _f32(n3, (-1, 3))
        face, mat = _i32(face), _i32(mat)
        n = len(ray6)
        ok, draws, out = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros((n, 6), np.float32)
        _check(
            self._lib.caster_hip_test_scatter(self._ctx, C.c_uint32(STRICT if strict else 0), C.c_int(n), _ptr(ray6), _ptr(p3), _ptr(n3),
                                              _ptr(face), _ptr(mat), C.c_uint32(seed), _ptr(ok), _ptr(out), _ptr(draws)),
            "caster_hip_test_scatter",
        )
        return dict(ok=ok, ray=out, draws=draws)

    def test_camera_ray(self, cam, uv2, seed, strict=True):
        cam, uv2 = _f32(cam, (19,)), _f32(uv2, (-1, 2))
        n = len(uv2)
        out, draws = np.zeros((n, 6), np.float32), np.zeros(n, np.int32)
        _check(
            self._lib.caster_hip_test_camera_ray(self._ctx, C.c_uint32(STRICT if strict else 0), _ptr(cam), C.c_int(n), _ptr(uv2),
                                                 C.c_uint32(seed), _ptr(out), _ptr(draws)),
            "caster_hip_test_camera_ray",
        )
        return dict(ray=out, draws=draws)

    def test_texture(self, mat, uv2, p3, strict=True):
        mat, uv2, p3 = _i32(mat), _f32(uv2, (-1, 2)), _f32(p3, (-1, 3))
        n = len(mat)
        rgb = np.zeros((n, 3), np.float32)
        _check(
            self._lib.caster_hip_test_texture(self._ctx, C.c_uint32(STRICT if strict else 0), C.c_int(n), _ptr(mat), _ptr(uv2), _ptr(p3), _ptr(rgb)),
            "caster_hip_test_texture",
        )
        return rgb

    def test_math(self, fn, in8, strict=True):
        in8 = _f32(in8, (-1, 8))
        out = np.zeros((len(in8), 3), np.float32)
        _check(self._lib.caster_hip_test_math(self._ctx, C.c_uint32(STRICT if strict else 0), C.c_int(fn), C.c_int(len(in8)), _ptr(in8), _ptr(out)),
               "caster_hip_test_math")
        return out

    def filter_audit(self, ray6, strict=True, vector_loop=False):
        """caster_hip_filter_audit: per ray, spheres the exact test accepts, those of them the filter rejected, smallest slack."""
        ray6 = _f32(ray6, (-1, 6))
        n = len(ray6)
        accepted, missed, slack = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.float32)
        _check(
            self._lib.caster_hip_filter_audit(self._ctx, C.c_uint32((STRICT if strict else 0) | (VECTOR_LOOP if vector_loop else 0)), C.c_int(n),
                                              _ptr(ray6), _ptr(accepted), _ptr(missed), _ptr(slack)),
            "caster_hip_filter_audit",
        )
        return dict(accepted=accepted, missed=missed, slack=slack)

    def test_rng(self, seed, pixel, sample, n):
        out = np.zeros(n, np.uint32)
        _check(self._lib.caster_hip_test_rng(self._ctx, C.c_uint32(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_int(n), _ptr(out)),
               "caster_hip_test_rng")
        return out


def reduce(contexts: Sequence["CasterHip"], accum_ptrs: Sequence[int], n_floats: int):
    """caster_hip_reduce: accum_ptrs[0] receives the sum of all the buffers (one context per GPU) through RCCL."""
    lib = load_caster_lib()
    n = len(contexts)
    ctxs = (C.c_void_p * n)(*[c._ctx.value for c in contexts])
    bufs = (C.c_void_p * n)(*accum_ptrs)
    _check(lib.caster_hip_reduce(ctxs, bufs, C.c_int(n), C.c_size_t(n_floats)), "caster_hip_reduce")
