"""bench.py's multi-rank leg rehearsed on a one-GPU box.  Collected late: it starts a process group of
its own, and a failure here must not hide the parity tests from a run that stops at the first failure."""
import json
import os
import signal
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` exactly as a driver without a launcher would call it: the script starts
    its two ranks itself.  On the one GPU of the test box both ranks use device 0 and the sum over ranks goes
    through gloo on host copies of the buffers (RCCL refuses two ranks on one GPU; gloo's own path for device
    tensors is not used: with two persistent render grids time-slicing one GPU it stalled in about one
    run of four inside the collective, a configuration no real run has).  Checks what the N > 1 line must
    hold: whole-job rays (the frame's, not a rank's), the max-over-ranks time, strong scaling, no CPU
    baseline, and the per-rank diagnostics.  A rank that stops making progress dumps its Python frames after
    80 s (CASTER_BENCH_TRACE), the launcher ends the job at 90 s, and the test ends the process group at 100 s."""
    env = dict(os.environ, CASTER_BENCH_BACKEND="gloo", CASTER_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    common = ["--width", "320", "--height", "180", "--spp", "24", "--steps", "2", "--warmup", "1"]
    p1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *common, "--no-cpu-baseline", "--vector-steps", "0",
                         "--workload-steps", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert p1.returncode == 0, p1.stderr[-2000:]
    one = json.loads([ln for ln in p1.stdout.splitlines() if ln.startswith("{")][0])
    assert one["ranks"]["n_ranks_seen"] == 1 and one["ranks"]["reduce_ms"] < 0.5

    env["CASTER_BENCH_TRACE"] = "1"
    out_path, err_path = tmp_path / "two_rank.out", tmp_path / "two_rank.err"
    with open(out_path, "w") as fo, open(err_path, "w") as fe:
        proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *common, "--launch-timeout", "90"],
                                stdout=fo, stderr=fe, env=env, start_new_session=True)
        try:
            rc = proc.wait(timeout=100)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)              # the launcher and both ranks
            proc.wait()
            raise AssertionError("two-rank bench did not finish in 100 s; trace:\n" + open(err_path).read()[-6000:])
    stdout, stderr = open(out_path).read(), open(err_path).read()
    assert rc == 0, (stdout + stderr)[-6000:]
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 alone prints
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and "cpu_baseline" not in two and "workloads" not in two
    assert two["rays_per_frame"] == one["rays_per_frame"]    # the same paths whatever the split
    assert abs(two["value"] - two["rays_per_frame"] / (two["ms_per_step"] * 1e-3) * 1e-6) / two["value"] < 1e-3
    assert 0.0 < two["roofline"]["frac"] <= 1.0 and "vector_loop" not in two["roofline"]
    assert "x2" in two["config"]["parallelism"]
    r = two["ranks"]
    assert r["n_ranks_seen"] == 2 and len(r["kernel_ms_per_rank"]) == 2 and r["devices"] == [0, 0]
    assert sum(r["rays_per_rank"]) == two["rays_per_frame"] and min(r["rays_per_rank"]) > 0
    assert r["kernel_ms_min"] <= r["kernel_ms_max"] == two["kernel_ms_per_frame"]
    assert r["reduce_ms"] > 0.0 and r["resolve_ms"] > 0.0 and r["clear_ms"] > 0.0
