"""bench.py's multi-rank leg rehearsed on a one-GPU box.  Collected late: it starts a process group of
its own, and a failure here must not hide the parity tests from a run that stops at the first failure."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line_on_one_gpu(tmp_path):
    """bench.py's multi-rank leg rehearsed on the one GPU of the test box: two ranks under
    torch.distributed.run, both on device 0, the sum over ranks through gloo on host copies of the buffers
    (RCCL refuses two ranks on one GPU; gloo's own path for device tensors stalled in about one run of
    four with two ranks sharing the GPU, inside the collective, while 3 000 concurrent frames of two
    processes without a process group never did).  Checks what the N > 1 line must hold: whole-job rays (the frame's, not a rank's), the
    max-over-ranks time, strong scaling, no CPU baseline."""
    env = dict(os.environ, CASTER_BENCH_BACKEND="gloo", CASTER_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--width", "320", "--height", "180", "--spp", "24", "--steps", "2", "--warmup", "1"]
    p1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *common, "--no-cpu-baseline", "--vector-steps", "0"],
                        capture_output=True, text=True, timeout=600, env=env)
    assert p1.returncode == 0, p1.stderr[-2000:]
    one = json.loads([ln for ln in p1.stdout.splitlines() if ln.startswith("{")][0])
    import signal

    port = str(29600 + os.getpid() % 300)
    env["CASTER_BENCH_TRACE"] = "1"
    out_path, err_path = tmp_path / "two_rank.out", tmp_path / "two_rank.err"
    with open(out_path, "w") as fo, open(err_path, "w") as fe:
        proc = subprocess.Popen([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                                 "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", *common],
                                stdout=fo, stderr=fe, env=env, start_new_session=True)
        try:
            rc = proc.wait(timeout=100)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)              # the launcher and both ranks
            proc.wait()
            raise AssertionError("two-rank bench did not finish in 100 s; trace:\n" + open(err_path).read()[-3000:])

    class p2:
        returncode, stdout, stderr = rc, open(out_path).read(), open(err_path).read()
    assert p2.returncode == 0, (p2.stdout + p2.stderr)[-3000:]
    lines = [ln for ln in p2.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 alone prints
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and "cpu_baseline" not in two
    assert two["rays_per_frame"] == one["rays_per_frame"]    # the same paths whatever the split
    assert abs(two["value"] - two["rays_per_frame"] / (two["ms_per_step"] * 1e-3) * 1e-6) / two["value"] < 1e-3
    assert 0.0 < two["roofline"]["frac"] <= 1.0 and "vector_loop" not in two["roofline"]
    assert "x2" in two["config"]["parallelism"]
