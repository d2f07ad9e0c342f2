"""caster_hip_reduce: the RCCL sum of the per-GPU accumulation buffers inside the C ABI.  A one-GPU
box can only form a one-rank communicator: the call must load RCCL, build the communicator, run the
collective on the context's stream behind the render kernel, and leave the sums as they were."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_reduce_on_one_gpu_and_argument_checks(gpu, ca):
    gpu.upload_scene(ca.scene_0())
    w, h, spp = 64, 36, 6
    cam = ca.default_camera(w / h)
    ref, rays, _ = gpu.render_image(cam, w, h, spp, 50, 9, strict=True)
    nb = w * h * 16
    a = gpu.malloc(nb)
    other = ca.CasterHip(0)
    try:
        for _ in range(2):                                     # the second call reuses the communicator
            gpu.memset_zero(a, nb)
            gpu.render(cam, w, h, 0, spp, 50, 9, a, strict=True)
            ca.reduce([gpu], [a], w * h * 4)                   # queued behind the render on the same stream
            got = gpu.download(a, (h, w, 4), np.float32)
            assert np.array_equal(got, ref)
        with pytest.raises(ca.CasterHipError, match="same device"):
            ca.reduce([gpu, other], [a, a], w * h * 4)         # RCCL wants one rank per GPU
        with pytest.raises(ca.CasterHipError):
            ca.reduce([], [], 4)
    finally:
        other.close()
        gpu.free(a)
