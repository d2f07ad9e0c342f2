"""The C host end to end on a GPU: command line -> scene -> dispatcher job API -> libcaster_hip ->
image buffer -> output BMP, against the CPU restatement's BMP bytes for the same configuration."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bmp_pixels(data, w, h):
    row = w * 3
    stride = row + (4 - row % 4) % 4
    body = np.frombuffer(data[54:], dtype=np.uint8).reshape(h, stride)[:, :row]
    return body.reshape(h, w, 3)


@pytest.mark.parametrize("scene,make", [("0", lambda ca: ca.scene_0(textured=True)), ("rtiow", lambda ca: ca.scene_rtiow(1)),
                                        ("perlin", lambda ca: ca.scene_perlin(1))])
def test_cli_output_matches_restatement(tmp_path, ca, orc, scene, make):
    exe = os.path.join(os.path.dirname(ca.lib_paths()["host"]), "craytracer")
    out = str(tmp_path / "output.bmp")
    height, spp, depth, seed = 90, 6, 50, 4242
    p = subprocess.run([exe, str(height), str(spp), str(depth), "1", "--scene", scene, "--strict", "--seed", str(seed),
                        "--batch", "4", "--out", out], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "Thread #0: Job finished!" in p.stdout and "App time:" in p.stdout      # src/dispatcher.c:50, src/main.c:111
    width = int(np.float32(height) * (np.float32(16.0) / np.float32(9.0)))
    data = open(out, "rb").read()
    assert data[:2] == b"BM" and int.from_bytes(data[18:22], "little") == width and int.from_bytes(data[22:26], "little") == height
    cam = ca.default_camera(np.float32(16.0) / np.float32(9.0))
    sums, rays = orc.render(orc.Scene.from_flat(make(ca)), cam, width, height, spp, depth, seed)
    want = orc.bmp_bytes(orc.resolve(sums, spp))
    assert len(data) == len(want) and data[:54] == want[:54]
    got_px, want_px = _bmp_pixels(data, width, height), _bmp_pixels(want, width, height)
    assert (np.abs(got_px.astype(int) - want_px.astype(int)) <= 1).all()
    assert (got_px == want_px).mean() > 0.999
    assert f"Rays: {rays} " in p.stdout
