#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and ms/frame of the per-pixel path-tracing path on MI355X.

Workload: the "Ray Tracing in One Weekend" final random-spheres
scene (487 spheres incl. ground and an enclosing emissive sky sphere), 1920x1080, 500 samples
per pixel, max depth 50, brute-force closest hit over the whole sphere list as the reference
does.  One step = one frame: every rank renders its sample partition of all pixels into a
float4 accumulation buffer resident in HBM, the buffers are summed onto rank 0 with one RCCL
reduce (N > 1), and rank 0 resolves the RGBA8 pixels.  Scene, camera and accumulation buffer
are resident on the GPU before the timed region starts.

A ray is one ray_color call with depth > 0 (src/caster.c:439), the unit of the reference's own
MRays/s figure (src/display.c:200-217).

Launching: `python bench.py --gpus N` starts its own N ranks (one fresh process per GPU, started
before anything touches a GPU) when no launcher has set WORLD_SIZE; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3      # MI355X vector FP32, 256 CU x 256 flop/clk x 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X dense FP16 matrix peak (no sparsity)


MFMA_FLOPS = 2 * 32 * 32 * 16      # one v_mfma_f32_32x32x16_f16
VECTOR_FILTER_FLOPS_PER_PAIR = 13  # the packed filter: 1 multiply + 6 fused multiply-adds per (ray, sphere)


def flops_per_ray(n_spheres: int, n_rects: int) -> int:
    """Algorithmic work of one ray: 18 flop per ray-sphere test (oc 3, oc.d 5, oc.oc 5, r*r and
    subtract 2, discriminant 3; FMA counts 2), 12 per rect test, and a nominal 100 for the
    closest-hit finish, scatter, random draws and texture.  Every ray meets every object."""
    return 18 * n_spheres + 12 * n_rects + 100


# ------------------------------------------------------------------ launcher --
def launch_ranks(n: int, argv, timeout_s: float) -> int:
    """Starts n ranks of this script, one per GPU, and relays rank 0's output.  The parent never
    imports torch and never touches a GPU; every rank is a fresh process that finds RANK,
    LOCAL_RANK, WORLD_SIZE and the rendezvous address in its environment, exactly as under
    torch.distributed.run.  Returns the exit status for the whole job: 0 only if every rank
    returned 0; on the first failure (or at the time limit) the remaining ranks are ended by PID."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs between processes on this driver
        # rank 0's stdout is this job's stdout (the JSON line); the other ranks' goes to stderr
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if rank == 0 else sys.stderr))
    deadline = time.monotonic() + timeout_s
    status = 0
    live = list(range(n))
    while live:
        for rank in list(live):
            rc = procs[rank].poll()
            if rc is None:
                continue
            live.remove(rank)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                print(f"bench.py: rank {rank} exited with status {rc}; ending the other ranks", file=sys.stderr, flush=True)
        if live and (status != 0 or time.monotonic() > deadline):
            if status == 0:
                status = 124
                print(f"bench.py: ranks {live} still running after {timeout_s:.0f} s; ending them", file=sys.stderr, flush=True)
            for rank in live:
                procs[rank].terminate()
            t_end = time.monotonic() + 10
            for rank in live:
                try:
                    procs[rank].wait(timeout=max(0.1, t_end - time.monotonic()))
                except subprocess.TimeoutExpired:
                    procs[rank].kill()
                    procs[rank].wait()
            live = []
        if live:
            time.sleep(0.05)
    return status


def cpu_baseline(fs, cam, width, height, depth, seed, seconds_target):
    """The CPU restatement of the reference path under a pthread port of its row-block dispatcher,
    timed on this host's cores over a bounded sample of the same workload (whole frame, few spp).
    Two legs: the port on the keyed generator (threads independent), and the same code on libc
    rand() as the reference ships it (src/common.c:10-13: one hidden state, serialised by glibc's lock)."""
    from oracle import oracle as orc

    sc = orc.Scene.from_flat(fs)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)), 64)
    # the reference's row split drops height % blocks rows: use a thread count that divides the height
    threads = max(t for t in range(1, cores + 1) if height % t == 0)
    # calibrate with a small frame, then size the sample so that the timed run lasts ~seconds_target
    t0 = time.perf_counter()
    _, _, rays = orc.dispatch(sc, cam, width // 8, threads * max(1, height // 8 // threads), 1, depth, seed, threads)
    dt = time.perf_counter() - t0
    rate = rays / max(dt, 1e-6)
    rays_per_spp = rays * 64.0
    spp = int(max(1, min(8, round(seconds_target * rate / rays_per_spp))))
    t0 = time.perf_counter()
    _, _, rays = orc.dispatch(sc, cam, width, height, spp, depth, seed, threads)
    dt = time.perf_counter() - t0
    out = {
        "value": round(rays / dt * 1e-6, 4),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"same scene and camera, {width}x{height}, {spp} spp, depth {depth}: {rays} rays in {dt:.1f} s "
                  f"on {threads} pthreads (row-block split of src/dispatcher.c)",
        "ms_per_frame_scaled": round(dt * 1e3 * 500.0 / spp, 1),
    }
    # as shipped: a quarter-size frame, one sample, same thread count, every draw through libc rand()
    w4, h4 = width // 2, threads * max(1, height // 2 // threads)
    t0 = time.perf_counter()
    _, _, rays_s = orc.dispatch(sc, cam, w4, h4, 1, depth, seed, threads, rng_mode=orc.RNG_LIBC)
    dt_s = time.perf_counter() - t0
    out["as_shipped"] = {
        "value": round(rays_s / dt_s * 1e-6, 4),
        "unit": "Mrays/s",
        "cores": threads,
        "sample": f"{w4}x{h4}, 1 spp, depth {depth}: {rays_s} rays in {dt_s:.1f} s on {threads} pthreads drawing from the one "
                  "libc rand() (src/common.c:10-13; glibc serialises the calls on its lock)",
        "note": "the reference's rdtsc block profiler in ray_color (src/caster.c:427, src/profiler.c) is not reproduced: "
                "its cost is not in this figure",
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=["strict", "fast"], default="strict",
                    help="strict: reference operation order (the parity-checked path); fast: FMA/approximate div")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0xC0FFEE)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vector-steps", type=int, default=2,
                    help="frames of the second leg (sphere filter on the vector unit), timed outside the main region; 0 = skip")
    ap.add_argument("--workload-steps", type=int, default=2,
                    help="frames per leg of the other baseline workloads (10 002 spheres on both loops, marble, 4K), "
                         "timed outside the main region at N = 1; 0 = skip")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds the self-launched ranks may take (N > 1)")
    ap.add_argument("--rank-env", action="store_true",
                    help="every rank prints the rendezvous variables it was started with as one JSON line and exits (launcher check, no GPU)")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # no launcher: become one.  Nothing has touched a GPU in this process, and nothing will.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout))
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.rank_env:
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
                         | {"torch_imported": "torch" in sys.modules}), flush=True)
        return

    trace = os.environ.get("CASTER_BENCH_TRACE") == "1"
    if trace:
        # a rank that stops making progress writes where it is (Python frames of every thread) before anyone ends it
        import faulthandler

        faulthandler.enable(file=sys.stderr)
        faulthandler.dump_traceback_later(float(os.environ.get("CASTER_BENCH_TRACE_AFTER", "80")), repeat=False, file=sys.stderr)

    import torch
    import torch.distributed as dist

    import craytracer_amd as ca
    from craytracer_amd.frame import render_frame

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    # Rehearsal switches for a one-GPU box (tests only): all ranks on device 0 and a backend that accepts
    # several ranks per GPU.  A real run has one rank per GPU over RCCL.
    backend = os.environ.get("CASTER_BENCH_BACKEND", "nccl")
    if os.environ.get("CASTER_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but the node shows {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    W, H, SPP = args.width, args.height, args.spp
    fs = ca.scene_rtiow(1)
    cam = ca.default_camera(W / H)
    gpu = ca.CasterHip(local_rank)
    gpu.upload_scene(fs)
    if args.chunk:
        gpu.set_chunk(args.chunk)
    stream = torch.cuda.current_stream(dev)
    gpu.set_stream(stream.cuda_stream)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    strict = args.mode == "strict"

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    kernel_ms = []
    spans = []                 # per frame: events at clear / render / reduce / resolve boundaries, on the frame's stream
    rays_step = [0]
    info_step = [None]

    def note(what):
        if trace:
            print(f"[rank {rank}] {what}", file=sys.stderr, flush=True)

    def reduce_sum():
        if backend == "nccl":
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)          # RCCL, on the stream the library renders on
        else:                                                        # rehearsal: the sum through host memory
            host = accum.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            accum.copy_(host)

    def mark():
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)       # the library launches on this same stream (set_stream above)
        return e

    def one_step(vector_loop=False):
        # everything of a frame is queued on one stream without a host wait in between: clear, render,
        # sum over ranks (RCCL), resolve; the host then waits once for the frame
        note("step: clear")
        e = [mark()]
        accum.zero_()
        e.append(mark())
        begin, count = render_frame(
            lambda b, c: gpu.render(cam, W, H, b, c, args.depth, args.seed, accum.data_ptr(), strict=strict,
                                    vector_loop=vector_loop),
            None, SPP, world, rank)
        e.append(mark())
        if world > 1:
            reduce_sum()
        e.append(mark())
        note("step: rendered and reduced (queued)")
        if rank == 0:
            gpu.resolve_dev(accum.data_ptr(), W, H, SPP, rgba.data_ptr())
        e.append(mark())
        info = gpu.render_info()               # waits for this rank's frame: the frame boundary
        note("step: frame complete")
        rays_step[0] = info["rays"]
        info["paths"] = W * H * count          # camera rays of this rank's share: one per path
        info_step[0] = info
        kernel_ms.append(info["kernel_ms"])    # HIP events on the kernel's own stream, around the kernel alone
        spans.append(e)

    for _ in range(args.warmup):
        one_step()
    kernel_ms.clear()
    spans.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    main_info = info_step[0]
    main_kernel_ms = float(np.mean(kernel_ms))
    # the frame's phases on this rank, averaged over the timed frames (the barrier above completed every event)
    phase_ms = [float(np.mean([sp[k].elapsed_time(sp[k + 1]) for sp in spans])) for k in range(4)]   # clear, render, reduce, resolve

    # Second leg, outside the timed region, one GPU only: the same frame with the sphere filter on the
    # vector unit (CASTER_HIP_VECTOR_LOOP), the form whose bound is the FP32 vector peak.
    vector_leg = None
    if world == 1 and args.vector_steps > 0:
        one_step(vector_loop=True)
        kernel_ms.clear()
        torch.cuda.synchronize(dev)
        tv = time.perf_counter()
        for _ in range(args.vector_steps):
            one_step(vector_loop=True)
        torch.cuda.synchronize(dev)
        tv = time.perf_counter() - tv
        vector_leg = (info_step[0], float(np.mean(kernel_ms)), tv / args.vector_steps * 1e3)
        rays_step[0] = main_info["rays"]

    stat_dev = dev if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([elapsed], dtype=torch.float64, device=stat_dev)
    r = torch.tensor([float(rays_step[0])], dtype=torch.float64, device=stat_dev)
    mine = torch.tensor([main_kernel_ms, *phase_ms, float(local_rank), float(main_info["rays"])], dtype=torch.float64, device=stat_dev)
    per_rank = [mine]
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        per_rank = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(per_rank, mine)
    elapsed = float(t.item())
    rays_frame = int(r.item())
    per_rank = [[float(x) for x in p.tolist()] for p in per_rank]
    kernel_ms_avg = max(p[0] for p in per_rank)

    if rank == 0:
        ns, nr = fs.n_spheres, fs.n_rects
        all_solid = bool((fs.materials["tex_type"] == 0).all())     # such scenes run the kernel instances without texture code
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_frame * args.steps / elapsed * 1e-6
        workload = f"rtiow_final_scene {W}x{H} {SPP}spp depth{args.depth}"
        roof = roofline_of(main_info, main_kernel_ms, args.mode, ns, nr, all_solid, main_info["paths"])
        recorded(roof, workload, args.mode, world, main_info["loop"], with_traffic=True)
        if vector_leg is not None:
            v_info, v_ms, v_wall_ms = vector_leg
            vr = roofline_of(v_info, v_ms, args.mode, ns, nr, all_solid, v_info["paths"])
            vr["ms_per_frame"] = round(v_wall_ms, 3)
            vr["mrays_per_s"] = round(v_info["rays"] / (v_wall_ms * 1e-3) * 1e-6, 2)
            vr["steps"] = args.vector_steps
            recorded(vr, workload, args.mode, world, v_info["loop"], with_traffic=False)
            roof["vector_loop"] = vr
        out = {
            "metric": "Mrays/s at 1080p 500spp (RTIOW final scene, depth 50, brute-force closest hit)",
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "objects": int(len(fs.objects)),
                "arithmetic_mode": args.mode,
                "parallelism": f"sample-partition x{world}" + ((" + RCCL sum-reduce to rank 0" if backend == "nccl" else
                                                                f" + {backend} sum-reduce to rank 0 (rehearsal)") if world > 1 else ""),
                "seed": args.seed,
            },
            "ms_per_frame": round(ms_per_step, 3),
            "rays_per_frame": rays_frame,
            "kernel_ms_per_frame": round(kernel_ms_avg, 3),
            "roofline": roof,
            # where a frame's time goes on the stream (HIP events between the phases), and who took part
            "ranks": {
                "n_ranks_seen": len(per_rank) if world == 1 else int(dist.get_world_size()),
                "kernel_ms_max": round(max(p[0] for p in per_rank), 3),
                "kernel_ms_min": round(min(p[0] for p in per_rank), 3),
                "kernel_ms_per_rank": [round(p[0], 3) for p in per_rank],
                "rays_per_rank": [int(p[6]) for p in per_rank],
                "devices": [int(p[5]) for p in per_rank],
                "clear_ms": round(per_rank[0][1], 3),
                "render_ms_max": round(max(p[2] for p in per_rank), 3),
                "reduce_ms": round(per_rank[0][3], 3),          # on rank 0: includes waiting for the slowest rank's render
                "reduce_ms_min": round(min(p[3] for p in per_rank), 3),   # the rank that arrived last waited least: the collective itself
                "resolve_ms": round(per_rank[0][4], 3),
                "host_ms_per_frame_beyond_stream": round(ms_per_step - sum(per_rank[0][1:5]), 3),
            },
        }
        if world == 1 and args.workload_steps > 0:
            out["workloads"] = other_workloads(ca, torch, dev, args, strict)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fs, cam, W, H, args.depth, args.seed, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    gpu.close()
    if world > 1:
        dist.destroy_process_group()
    if trace:
        import faulthandler

        faulthandler.cancel_dump_traceback_later()


def other_workloads(ca, torch, dev, args, strict):
    """The other workloads of the baseline list, one GPU, outside the timed region: a warm-up frame and
    `--workload-steps` timed frames each, sized to a fraction of a second per frame.  Each entry carries
    the executed-work roofline of its kernel (roofline_of) and the HIP-event kernel time."""
    legs = [
        # name, scene, width, height, spp, vector loop
        ("stress_10k_matrix", lambda: ca.scene_stress(10000, 2), 1920, 1080, 16, False),
        ("stress_10k_vector", lambda: ca.scene_stress(10000, 2), 1920, 1080, 16, True),
        ("marble_perlin", lambda: ca.scene_perlin(1), 1920, 1080, 64, False),
        ("rtiow_4k", lambda: ca.scene_rtiow(1), 3840, 2160, 16, False),
    ]
    out = {}
    for name, make, w, h, spp, vector in legs:
        fs = make()
        cam = ca.default_camera(w / h)
        g = ca.CasterHip(dev.index)
        try:
            g.upload_scene(fs)
            acc = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
            torch.cuda.synchronize(dev)
            ms, info = [], None
            for k in range(1 + args.workload_steps):
                acc.zero_()
                torch.cuda.synchronize(dev)
                g.render(cam, w, h, 0, spp, args.depth, args.seed, acc.data_ptr(), strict=strict, vector_loop=vector)
                info = g.render_info()
                if k:
                    ms.append(info["kernel_ms"])
            kms = float(np.mean(ms))
            all_solid = bool((fs.materials["tex_type"] == 0).all())
            roof = roofline_of(info, kms, args.mode, fs.n_spheres, fs.n_rects, all_solid, w * h * spp)
            roof.pop("note", None)
            out[name] = {
                "workload": f"{name} {w}x{h} {spp}spp depth{args.depth}", "objects": int(len(fs.objects)), "steps": args.workload_steps,
                "rays_per_frame": int(info["rays"]), "kernel_ms": round(kms, 3),
                "mrays_per_s": round(info["rays"] / (kms * 1e-3) * 1e-6, 1), "roofline": roof,
            }
            del acc
        finally:
            g.close()
    return out


def roofline_of(info, kernel_ms, mode, n_spheres, n_rects, all_solid, paths=0):
    """Roofline entry of one launch of the render kernel, from what the launch itself reports
    (caster_hip_render_info: which sphere loop ran, wave trips, rays) and its HIP-event duration.

    Executed work, so the fraction cannot exceed 1:
      * matrix loop: every wave trip meets 4 blocks of 32 rays with every block of 32 spheres through two
        v_mfma_f32_32x32x16_f16 -> trips x blocks x 4 x 2 x 32768 flop, against the dense FP16 matrix peak.
        Only rays from a path's second on go through the loop: the camera rays of a tile are traced against
        the tile's short sphere list in batches of 64 (primary_batches), so a frame executes about a third
        fewer trips, and matrix flops, than it has rays / 128;
      * vector loop: every trip evaluates the packed filter (1 multiply + 6 FMAs = 13 flop) for 128 ray
        slots and every sphere of the padded list, against the FP32 vector peak.
    The work of the reference's loop as written (18 flop per ray-sphere test) is reported beside it as
    algorithmic_equivalent; it is not a fraction of any unit of this kernel."""
    sec = kernel_ms * 1e-3
    trips = info["wave_trips"]
    alg = info["rays"] * flops_per_ray(n_spheres, n_rects) / sec * 1e-12
    if info["loop_kind"] >= 2:
        executed = trips * info["mfma_blocks"] * 4 * 2 * MFMA_FLOPS / sec * 1e-12
        bound, peak, kind = "mfma", F16_MFMA_PEAK_TFLOPS, info["loop_kind"]
        note = ("executed flops of the conservative sphere filter on the matrix cores (f16 operands, f32 accumulate) "
                "against the dense f16 matrix peak; camera rays do not go through the filter (per-tile sphere lists), so the "
                "executed flops cover the rays from a path's second on; the kernel as a whole is bound by vector-instruction "
                "issue (one instruction per 4 SIMD cycles: see counters.simd_cycles_per_valu_inst)")
    else:
        padded = (n_spheres + 7) // 8 * 8
        executed = trips * 128 * padded * VECTOR_FILTER_FLOPS_PER_PAIR / sec * 1e-12
        bound, peak, kind = "valu_fp32", FP32_VALU_PEAK_TFLOPS, info["loop_kind"]
        note = ("executed flops of the packed FP32 sphere filter (13 per ray-sphere pair) against the FP32 vector peak; "
                "algorithmic_equivalent counts the 18 flop of the reference's test per pair")
    return {
        "bound": bound,
        "achieved": round(float(executed), 3),
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": round(float(executed) / peak, 4),
        "traffic": None,
        "kernel": f"caster::render_kernel_{mode}<{kind}, {'false' if all_solid else 'true'}>",
        "loop": info["loop"],
        "kernel_ms": round(kernel_ms, 3),
        "wave_trips": int(trips),
        "primary_batches": int(info.get("primary_batches", 0)),
        # rays traced by the loop (all but one camera ray per batch lane) over the loop's ray slots
        "slots_in_use": round(max(0, info["rays"] - paths) / max(1, trips * 128), 4),
        "algorithmic_equivalent": {
            "tflops": round(float(alg), 3), "flops_per_ray": flops_per_ray(n_spheres, n_rects),
            "of_fp32_vector_peak": round(float(alg) / FP32_VALU_PEAK_TFLOPS, 4),
        },
        "note": note,
    }


def _latest_profile(pattern):
    import glob

    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pattern)))
    for path in reversed(paths):
        try:
            with open(path) as f:
                yield json.load(f)
        except (OSError, ValueError):
            continue


def recorded(roof, workload, mode, world, loop, with_traffic):
    """Adds the hardware-counter figures of this workload's render kernel (vector-instruction issue, active
    lanes, FP32 flops executed, matrix-pipe busy; HBM bytes per launch) to a roofline entry.  Counters cannot be
    read inside a timed run: they are the ones recorded with rocprofv3 --pmc for this exact workload, mode and
    loop (tools/pmc_summary.py -> profiles/*pmc_summary*.json, *hbm_traffic.json).  A record carries the hash of
    the kernel sources it was taken on; figures from other sources than the ones in the tree are not quoted as
    current: `counters` / `traffic` stay null, `counters_stale` is true and `counters_recorded_on` names the hash."""
    from craytracer_amd.buildinfo import kernel_source_hash

    roof["counters"] = None
    if world != 1:
        return
    here = kernel_source_hash()
    roof["kernel_source_hash"] = here
    for rec in _latest_profile("*pmc_summary*.json"):
        for entry in rec.get("kernels", []):
            if entry.get("workload") == workload and entry.get("arithmetic_mode") == mode and entry.get("loop") == loop:
                if rec.get("kernel_source_hash") == here:
                    roof["counters"] = entry.get("figures")
                    if with_traffic and entry.get("figures", {}).get("hbm_bytes_per_launch") is not None:
                        roof["traffic"] = entry["figures"]["hbm_bytes_per_launch"]
                    valu = entry.get("figures", {}).get("valu_insts")
                    if valu:
                        # The unit that bounds the kernel: a SIMD issues one vector instruction of a 64-lane wave per 4
                        # cycles.  Instructions of this workload's launch (recorded counter, same sources) over this
                        # run's kernel time, against 1 024 SIMDs at the 2.4 GHz peak clock; under load the part runs
                        # at about 2.2 GHz and a few instructions issue in fewer than 4 cycles.
                        peak = 256 * 4 * 2.4e9 / 4.0
                        rate = valu / (roof["kernel_ms"] * 1e-3)
                        roof["valu_issue"] = {"bound": "vector-instruction issue", "achieved": round(rate * 1e-9, 2), "peak": round(peak * 1e-9, 2),
                                              "unit": "G instructions/s", "frac_of_peak_clock": round(rate / peak, 4)}
                else:
                    roof["counters_stale"] = True
                    roof["counters_recorded_on"] = rec.get("kernel_source_hash", "unstamped (an earlier build)")
                return


if __name__ == "__main__":
    main()
