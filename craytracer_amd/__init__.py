"""craytracer_amd -- MI355X (gfx950) implementation of CRaytracer's per-pixel
path-tracing path.

The product is two shared libraries built in-tree under ``craytracer_amd/lib``:

* ``libcaster_hip.so``      -- hand-written HIP kernels behind a C ABI
  (``include/caster_hip.h``);
* ``libcraytracer_host.so`` -- the C host that mirrors the reference's scene,
  camera, image and dispatcher interfaces (``craytracer_amd/host/craytracer.h``).

This package is a thin ctypes binding over both, used by the tests and by
``bench.py``.  There is no Python or CPU implementation of the path behind it:
importing works without a GPU (so that symbols can be checked), but creating a
:class:`CasterHip` context fails loudly when the library or a HIP device is
missing.
"""
from .binding import (  # noqa: F401
    CasterHip,
    CasterHipError,
    FlatScene,
    OBJECT_DTYPE,
    MATERIAL_DTYPE,
    STRICT,
    default_camera,
    calculate_cam_info,
    lib_paths,
    load_caster_lib,
    load_host_lib,
    scene_0,
    scene_perlin,
    scene_rtiow,
    make_noise,
    NOISE_DTYPE,
    scene_stress,
    make_test_texture,
    bmp_encode,
    reduce,
    NOALLOC_MAX_PIXELS,
    LOOP_NAMES,
    MATH_REFLECTANCE,
    MATH_REFRACT,
    MATH_REFLECT,
    MATH_UV_SPHERE,
    MATH_NEAR_ZERO,
    MATH_TO_UNIT,
)
