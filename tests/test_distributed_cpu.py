"""The multi-GPU frame logic (sample partition per rank, sum of the linear radiance buffers onto
rank 0, resolve there) on two gloo ranks.  The renderer plugged into the frame helper here is the
CPU restatement -- the GPU path uses the same helper with libcaster_hip and RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, cfg, out_path):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import craytracer_amd as ca
    from craytracer_amd.frame import render_frame
    from oracle import oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, spp, depth, seed = cfg
    fs = ca.scene_0()
    sc = orc.Scene.from_flat(fs)
    cam = ca.default_camera(w / h)
    accum = torch.zeros((h, w, 4), dtype=torch.float32)
    rays = torch.zeros(1, dtype=torch.int64)

    def render_range(begin, count):
        sums, n = orc.render(sc, cam, w, h, count, depth, seed, sample_begin=begin)
        accum.add_(torch.from_numpy(sums))
        rays[0] = n

    render_frame(render_range, lambda: dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM), spp, world, rank)
    dist.reduce(rays, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.savez(out_path, sums=accum.numpy(), rays=rays.numpy())
    dist.destroy_process_group()


def test_two_rank_sample_partition_matches_single_rank(tmp_path, orc, ca):
    cfg = (32, 18, 9, 30, 21)          # 9 spp over 2 ranks: 4 + 5
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(2, _free_port(), cfg, out), nprocs=2, join=True)
    got = np.load(out)
    w, h, spp, depth, seed = cfg
    ref, ref_rays = orc.render(orc.Scene.from_flat(ca.scene_0()), ca.default_camera(w / h), w, h, spp, depth, seed)
    assert int(got["rays"][0]) == ref_rays                       # the same paths, whatever the rank count
    assert np.array_equal(got["sums"][..., 3], ref[..., 3])
    # float sums grouped per rank differ from the sequential sum only in the last bits
    assert np.allclose(got["sums"][..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)
    rgba = orc.resolve(got["sums"], spp)
    assert (np.abs(rgba.astype(int) - orc.resolve(ref, spp).astype(int)) <= 1).all()


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher starts N ranks itself, each a fresh process with the
    rendezvous variables of torch.distributed.run, before anything imports torch; rank 0's stdout is the
    job's stdout.  Without a GPU the ranks then stop with a loud error, and the job's status says so."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--rank-env"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    mine = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    others = [json.loads(ln) for ln in p.stderr.splitlines() if ln.startswith("{")]
    assert len(mine) == 1 and mine[0]["RANK"] == "0" and len(others) == 2          # rank 0 alone writes to the job's stdout
    ranks = sorted(mine + others, key=lambda d: int(d["RANK"]))
    assert [d["RANK"] for d in ranks] == ["0", "1", "2"] and [d["LOCAL_RANK"] for d in ranks] == ["0", "1", "2"]
    assert all(d["WORLD_SIZE"] == "3" and d["MASTER_ADDR"] == "127.0.0.1" and not d["torch_imported"] for d in ranks)
    assert len({d["MASTER_PORT"] for d in ranks}) == 1
    # a launcher that started another number of ranks than --gpus says is an error, not a silent single-rank run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rank-env"], capture_output=True,
                         text=True, timeout=120, env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE=4" in bad.stderr
    if not torch.cuda.is_available():
        # no GPU here: every rank fails loudly and the job ends promptly with a non-zero status
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode != 0 and "needs a HIP device" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
