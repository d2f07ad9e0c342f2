"""The conservative sphere filter, audited pair by pair on the GPU.

The render kernel runs the reference's sphere test (src/caster.c:99-121) only on the spheres a
cheap filter lets through, where the reference tests all of them (src/caster.c:204).  A filter that
rejected a sphere the exact test accepts would silently lose a hit.  caster_hip_filter_audit runs the
exact test on EVERY (ray, sphere) pair and evaluates the filter exactly as the render kernel does --
the packed vector form, or the real matrix instructions on the real operand table -- and counts the
accepted pairs the filter rejected.  That count must be zero, and the smallest slack
(key - threshold) / (|o'|^2 + |c'|^2 + r^2) over all accepted pairs says how much of the margin g the
arithmetic used up on the most critical pair: the filter's error there is g - slack.
"""
import numpy as np
import pytest

from test_gpu_functions import _grazing_rays

pytestmark = pytest.mark.gpu

# margins of craytracer_amd/csrc/caster_launch.h: vector form strict / fast, and what the matrix form adds
G_STRICT, G_FAST, G_MATRIX_EXTRA = 1.0e-5, 2.5e-6, 3.0e-5


def _scene(ca, name, rng):
    if name == "rtiow":
        return ca.scene_rtiow(1), (0.0, 0.0)
    if name == "stress_2600":
        return ca.scene_stress(2600, 2), (0.0, 0.0)
    if name == "stress_10000":
        return ca.scene_stress(10000, 2), (0.0, 0.0)
    if name == "mixed_scales":
        fs = ca.scene_stress(400, 7)
        objs = fs.objects.copy()
        small = objs["g"][:, 3] < 50
        k = int(small.sum())
        objs["g"][small, 3] = np.exp(rng.uniform(np.log(0.01), np.log(0.3), size=k)).astype(np.float32)
        objs["g"][small, 0] = rng.uniform(-40, 40, size=k).astype(np.float32)
        objs["g"][small, 2] = rng.uniform(-40, 40, size=k).astype(np.float32)
        return ca.FlatScene(objs, fs.materials, []), (0.0, 0.0)
    fs = ca.scene_rtiow(5)
    objs = fs.objects.copy()
    objs["g"][:, 0] += 3000.0
    objs["g"][:, 2] -= 1500.0
    return ca.FlatScene(objs, fs.materials, []), (3000.0, -1500.0)


@pytest.mark.parametrize("name", ["rtiow", "stress_2600", "stress_10000", "mixed_scales", "offset_scene"])
def test_filter_never_rejects_an_accepted_pair(gpu, ca, name, capsys):
    rng = np.random.default_rng(401)
    fs, (dx, dz) = _scene(ca, name, rng)
    gpu.upload_scene(fs)
    n = 40000 if name != "stress_10000" else 12000
    base = fs
    if dx or dz:
        objs = fs.objects.copy()
        objs["g"][:, 0] -= dx
        objs["g"][:, 2] -= dz
        base = ca.FlatScene(objs, fs.materials, [])
    rays = _grazing_rays(rng, base, n)          # silhouettes (discriminant ~ 0), surface starts (t ~ 0), inside starts
    rays[:, 0] += dx
    rays[:, 2] += dz
    # bounce-like rays: unnormalised directions of every length, as the scatter functions leave them
    rays[::7, 3:] *= np.exp(rng.uniform(np.log(1e-3), np.log(1e3), size=(len(rays[::7]), 1))).astype(np.float32)
    report = []
    for strict in (True, False):
        for vector_loop in (False, True):
            a = gpu.filter_audit(rays, strict=strict, vector_loop=vector_loop)
            assert a["accepted"].sum() > n // 4                   # the rays do meet spheres
            assert a["missed"].sum() == 0, (name, strict, vector_loop, int(a["missed"].sum()))
            g = (G_STRICT if strict else G_FAST) + (0.0 if vector_loop else G_MATRIX_EXTRA)
            slack = float(a["slack"][np.isfinite(a["slack"])].min())
            assert slack > 0.0
            report.append(f"{name:13s} {'strict' if strict else 'fast  '} {'vector' if vector_loop else 'matrix'}: "
                          f"{int(a['accepted'].sum()):8d} accepted pairs, 0 missed, min slack {slack:.3e} of margin {g:.2e} "
                          f"(error used {(g - slack) / g * 100:5.1f} %)")
    with capsys.disabled():
        print("\n" + "\n".join(report))
