"""Developer tool: wave-level event counts per loop trip from the -DCASTER_STATS build
(make -C craytracer_amd/csrc stats)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import craytracer_amd.binding as b  # noqa: E402

b.lib_paths = lambda: {"caster": os.path.join(ROOT, "craytracer_amd", "lib", "libcaster_hip_stats.so"),
                       "host": os.path.join(ROOT, "craytracer_amd", "lib", "libcraytracer_host.so")}
import craytracer_amd as ca  # noqa: E402

NAMES = {0: "trips", 1: "drain iterations", 2: "accept A", 3: "accept B", 4: "unit-sphere loop iterations",
         5: "disk loop iterations", 6: "bounce executions", 7: "lambertian", 8: "metal", 9: "dielectric", 10: "light",
         11: "path starts", 12: "candidate events", 13: "candidate lanes"}
lib = b.load_caster_lib()
gpu = ca.CasterHip(0)
scene = sys.argv[1] if len(sys.argv) > 1 else "rtiow"
fs = ca.scene_rtiow(1) if scene == "rtiow" else ca.scene_stress(int(scene.split(":")[1]), 2)
gpu.upload_scene(fs)
w, h, spp = 960, 540, 64
cam = ca.default_camera(w / h)
for mode in ("strict", "fast"):
    out = (C.c_ulonglong * 32)()
    getattr(lib, "caster_debug_stats_" + mode)(out, 1)
    _, rays, ms = gpu.render_image(cam, w, h, spp, 50, 1, strict=(mode == "strict"))
    getattr(lib, "caster_debug_stats_" + mode)(out, 1)
    trips = out[0]
    print(mode, "rays", rays, "trips", trips, "slot utilisation %.3f" % (rays / (trips * 128.0)))
    for i, name in NAMES.items():
        print(f"  {name:30s} {out[i]:12d}  per trip {out[i] / trips:8.3f}")
