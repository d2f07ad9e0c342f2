"""bench.py's one-line JSON contract on a GPU (reduced size), and the RCCL leg of the multi-GPU
frame on a one-rank process group: the collective runs on torch's stream after the render kernel
that the library put on the same stream."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--width", "320", "--height", "180", "--spp", "24",
                        "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.5", "--workload-steps", "1"], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    # executed work against the unit it runs on: a fraction of a peak, never above it
    assert r["bound"] == "mfma" and r["peak"] == 2500.0 and r["loop"] == "matrix"
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["wave_trips"] > 0 and 0.0 < r["slots_in_use"] <= 1.0     # a frame this small leaves most of the 4 096 waves nearly idle
    assert r["traffic"] is None and r["counters"] is None    # the recorded counters belong to the full-size workload only
    v = r["vector_loop"]                              # second leg: the filter on the vector unit, against the FP32 vector peak
    assert v["bound"] == "valu_fp32" and v["peak"] == 157.3 and v["loop"] == "vector"
    assert 0.0 < v["frac"] <= 1.0 and abs(v["frac"] - v["achieved"] / v["peak"]) < 1e-3
    assert v["algorithmic_equivalent"]["flops_per_ray"] == 18 * 487 + 100
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mrays/s"
    # the same code drawing from the one libc rand(), as the reference ships
    assert c["as_shipped"]["value"] > 0 and c["as_shipped"]["cores"] == c["cores"] and "rand()" in c["as_shipped"]["sample"]
    # recorded counters are quoted only for the kernel sources they were taken on
    assert len(r["kernel_source_hash"]) == 16 and r["counters"] is None
    # where the frame's time goes on the stream, and who took part
    k = d["ranks"]
    assert k["n_ranks_seen"] == 1 and k["devices"] == [0] and k["kernel_ms_per_rank"] == [k["kernel_ms_max"]]
    assert k["clear_ms"] > 0 and k["render_ms_max"] >= k["kernel_ms_max"] * 0.99 and k["resolve_ms"] > 0 and 0 <= k["reduce_ms"] < 0.5
    # the other baseline workloads, each with the executed-work roofline of its own kernel
    wl = d["workloads"]
    assert set(wl) == {"stress_10k_matrix", "stress_10k_vector", "marble_perlin", "rtiow_4k"}
    assert wl["stress_10k_matrix"]["objects"] == 10002 and wl["stress_10k_matrix"]["roofline"]["loop"] == "matrix_large"
    assert wl["stress_10k_vector"]["roofline"]["loop"] == "vector_tiled" and wl["stress_10k_vector"]["roofline"]["bound"] == "valu_fp32"
    assert wl["stress_10k_matrix"]["rays_per_frame"] == wl["stress_10k_vector"]["rays_per_frame"]      # two loops, one set of paths
    assert wl["marble_perlin"]["roofline"]["kernel"].endswith("true>") and wl["rtiow_4k"]["workload"].startswith("rtiow_4k 3840x2160")
    for e in wl.values():
        assert 0.0 < e["roofline"]["frac"] <= 1.0 and e["kernel_ms"] > 0 and e["mrays_per_s"] > 0
    # value = rays of all steps / wall time
    assert abs(d["value"] - d["rays_per_frame"] / (d["ms_per_step"] * 1e-3) * 1e-6) / d["value"] < 1e-3


_RCCL_WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import craytracer_amd as ca
from craytracer_amd.frame import render_frame
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[2])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
w, h, spp, depth, seed = 96, 54, 12, 50, 31
fs = ca.scene_rtiow(1)
cam = ca.default_camera(w / h)
gpu = ca.CasterHip(0)
gpu.upload_scene(fs)
gpu.set_stream(torch.cuda.current_stream(dev).cuda_stream)
accum = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
# world is passed as 2 to the helper so that it calls the collective; the group has one rank,
# so the rank renders the whole sample range itself
render_frame(lambda b, c: gpu.render(cam, w, h, 0, spp, depth, seed, accum.data_ptr(), strict=True),
             lambda: dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM), spp, 2, 0)
torch.cuda.synchronize(dev)
got = accum.cpu().numpy()
ref, rays, _ = gpu.render_image(cam, w, h, spp, depth, seed, strict=True)
assert np.array_equal(got, ref), float(np.abs(got - ref).max())
dist.destroy_process_group()
print("rccl-ok", rays)
"""


def test_rccl_reduce_follows_the_render_kernel(tmp_path):
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script), ROOT, "29517"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "rccl-ok" in p.stdout
