"""Developer tool: wave-level event counts per loop trip from the -DCASTER_STATS build
(make -C craytracer_amd/csrc stats), or with --timers the
share of the waves' time per region of the render kernel (make -C craytracer_amd/csrc timers)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import craytracer_amd.binding as b  # noqa: E402

TIMERS = "--timers" in sys.argv
if TIMERS:
    sys.argv.remove("--timers")
b.lib_paths = lambda: {"caster": os.path.join(ROOT, "craytracer_amd", "lib",
                                              "libcaster_hip_timers.so" if TIMERS else "libcaster_hip_stats.so"),
                       "host": os.path.join(ROOT, "craytracer_amd", "lib", "libcraytracer_host.so")}
import craytracer_amd as ca  # noqa: E402

NAMES = {0: "trips", 1: "drain iterations", 2: "accept A", 3: "accept B", 4: "unit-sphere loop iterations",
         5: "disk loop iterations", 6: "bounce executions", 7: "lambertian", 8: "metal", 9: "dielectric", 10: "light",
         11: "ring pops", 12: "candidate events", 13: "camera-ray batches", 14: "tile-list spheres tested"}
lib = b.load_caster_lib()
gpu = ca.CasterHip(0)
def opt(name, default):
    if name in sys.argv:
        i = sys.argv.index(name)
        v = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
        return v
    return default


w, h, spp, chunk = opt("--width", 960), opt("--height", 540), opt("--spp", 64), opt("--chunk", 0)
scene = sys.argv[1] if len(sys.argv) > 1 else "rtiow"
fs = ca.scene_rtiow(1) if scene == "rtiow" else ca.scene_stress(int(scene.split(":")[1]), 2)
gpu.upload_scene(fs)
if chunk:
    gpu.set_chunk(chunk)
cam = ca.default_camera(w / h)
for mode in ("strict", "fast"):
    out = (C.c_ulonglong * 32)()
    getattr(lib, "caster_debug_stats_" + mode)(out, 1)
    _, rays, ms = gpu.render_image(cam, w, h, spp, 50, 1, strict=(mode == "strict"))
    getattr(lib, "caster_debug_stats_" + mode)(out, 1)
    trips = out[0]
    print(mode, "rays", rays, "trips", trips, "slot utilisation %.3f" % (rays / (trips * 128.0)), "kernel %.2f ms" % ms)
    if not TIMERS:
        for i, name in NAMES.items():
            print(f"  {name:30s} {out[i]:12d}  per trip {out[i] / trips:8.3f}")
        continue
    total = out[20]
    wl = (C.c_ulonglong * 65536)()
    getattr(lib, "caster_debug_wavelog_" + mode)(wl)
    wl = np.array(wl, dtype=np.uint64).reshape(4096, 16).astype(np.int64)
    t0 = wl[:, 0].min()
    st, en = (wl[:, 0] - t0) * 1e-5, (wl[:, 1] - t0) * 1e-5
    print("  wave start ms: max %.2f; wave end ms: min %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f; units/wave min %d max %d" %
          (st.max(), en.min(), np.percentile(en, 10), np.percentile(en, 50), np.percentile(en, 90), en.max(),
           wl[:, 2].min(), wl[:, 2].max()))
    for i in np.argsort(-en)[:8]:
        print("    wave %4d end %.2f last fetch %.2f (unit %d) queue empty seen %.2f trips after that %d of %d" %
              (i, en[i], (wl[i, 4] - t0) * 1e-5, wl[i, 5], (wl[i, 6] - t0) * 1e-5, wl[i, 3] - wl[i, 7], wl[i, 3]))
        print("      per trip: bounce %.0f refill %.0f isect %.0f (queue %.0f) total %.0f ticks; rays %d, utilisation %.3f" %
              (wl[i, 8] / wl[i, 3], wl[i, 9] / wl[i, 3], wl[i, 10] / wl[i, 3], wl[i, 11] / wl[i, 3], wl[i, 12] / wl[i, 3],
               wl[i, 13] & 0xFFFFFFFF, (wl[i, 13] & 0xFFFFFFFF) / (128.0 * wl[i, 3])))
        nt = max(wl[i, 3] - (wl[i, 13] >> 32), 1)
        print("      the last %d trips (paths gathered), per trip:" % nt, end="")
        print(" isect %.0f (queue %.0f) bounce %.0f refill %.0f ticks" %
              ((wl[i, 14] & 0xFFFFFFFF) / nt, (wl[i, 14] >> 32) / nt, (wl[i, 15] & 0xFFFFFFFF) / nt, (wl[i, 15] >> 32) / nt))
    print("  shader clock while busy %.0f MHz; mean wave busy time %.2f ms of %.2f ms" %
          (100.0 * total / out[22], out[22] / 4096.0 * 1e-5, ms))
    for i, name in enumerate(("schedule", "bounce", "refill", "intersect (incl. queue)", "total", "queue")):
        print(f"  ticks {name:24s} {out[16 + i] / trips:10.1f} per trip  {100.0 * out[16 + i] / max(total, 1):6.2f} %")
