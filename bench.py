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


def cpu_baseline(fs, cam, width, height, depth, seed, seconds_target):
    """The CPU restatement of the reference path under a pthread port of its row-block dispatcher,
    timed on this host's cores over a bounded sample of the same workload (whole frame, few spp)."""
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
    return {
        "value": round(rays / dt * 1e-6, 4),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"same scene and camera, {width}x{height}, {spp} spp, depth {depth}: {rays} rays in {dt:.1f} s "
                  f"on {threads} pthreads (row-block split of src/dispatcher.c)",
        "ms_per_frame_scaled": round(dt * 1e3 * 500.0 / spp, 1),
    }


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
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import craytracer_amd as ca
    from craytracer_amd.frame import render_frame, sample_partition

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    # Rehearsal switches for a one-GPU box (tests only): all ranks on device 0 and a backend that accepts
    # several ranks per GPU.  A real run has one rank per GPU over RCCL.
    backend = os.environ.get("CASTER_BENCH_BACKEND", "nccl")
    if os.environ.get("CASTER_BENCH_SHARE_GPU") == "1":
        local_rank = 0
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
    rays_step = [0]
    info_step = [None]

    trace = os.environ.get("CASTER_BENCH_TRACE") == "1"

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

    def one_step(vector_loop=False):
        # everything of a frame is queued on one stream without a host wait in between: clear, render,
        # sum over ranks (RCCL), resolve; the host then waits once for the frame
        note("step: clear")
        accum.zero_()
        begin, count = render_frame(
            lambda b, c: gpu.render(cam, W, H, b, c, args.depth, args.seed, accum.data_ptr(), strict=strict,
                                    vector_loop=vector_loop),
            reduce_sum if world > 1 else None,
            SPP, world, rank)
        note("step: rendered and reduced (queued)")
        if rank == 0:
            gpu.resolve_dev(accum.data_ptr(), W, H, SPP, rgba.data_ptr())
        info = gpu.render_info()               # waits for this rank's frame: the frame boundary
        note("step: frame complete")
        rays_step[0] = info["rays"]
        info_step[0] = info
        kernel_ms.append(info["kernel_ms"])    # HIP events on the kernel's own stream, around the kernel alone

    for _ in range(args.warmup):
        one_step()
    kernel_ms.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    main_info = info_step[0]
    main_kernel_ms = float(np.mean(kernel_ms))

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
    k = torch.tensor([main_kernel_ms], dtype=torch.float64, device=stat_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    rays_frame = int(r.item())
    kernel_ms_avg = float(k.item())

    if rank == 0:
        ns, nr = fs.n_spheres, fs.n_rects
        all_solid = bool((fs.materials["tex_type"] == 0).all())     # such scenes run the kernel instances without texture code
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_frame * args.steps / elapsed * 1e-6
        workload = f"rtiow_final_scene {W}x{H} {SPP}spp depth{args.depth}"
        roof = roofline_of(main_info, main_kernel_ms, args.mode, ns, nr, all_solid)
        roof["traffic"] = recorded_traffic(workload, args.mode, world)
        roof["counters"] = recorded_counters(workload, args.mode, world, main_info["loop"])
        if vector_leg is not None:
            v_info, v_ms, v_wall_ms = vector_leg
            vr = roofline_of(v_info, v_ms, args.mode, ns, nr, all_solid)
            vr["ms_per_frame"] = round(v_wall_ms, 3)
            vr["mrays_per_s"] = round(v_info["rays"] / (v_wall_ms * 1e-3) * 1e-6, 2)
            vr["steps"] = args.vector_steps
            vr["counters"] = recorded_counters(workload, args.mode, world, v_info["loop"])
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
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fs, cam, W, H, args.depth, args.seed, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    gpu.close()
    if world > 1:
        dist.destroy_process_group()


def roofline_of(info, kernel_ms, mode, n_spheres, n_rects, all_solid):
    """Roofline entry of one launch of the render kernel, from what the launch itself reports
    (caster_hip_render_info: which sphere loop ran, wave trips, rays) and its HIP-event duration.

    Executed work, so the fraction cannot exceed 1:
      * matrix loop: every wave trip meets 4 blocks of 32 rays with every block of 32 spheres through two
        v_mfma_f32_32x32x16_f16 -> trips x blocks x 4 x 2 x 32768 flop, against the dense FP16 matrix peak;
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
                "against the dense f16 matrix peak; the kernel as a whole is co-bound by vector-instruction issue "
                "(exact tests, shading, path starts): see counters")
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
        "slots_in_use": round(info["rays"] / max(1, trips * 128), 4),
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


def recorded_counters(workload, mode, world, loop):
    """Hardware-counter figures of this workload's render kernel (vector-instruction issue, active
    lanes, FP32 flops executed, matrix-pipe busy).  Counters cannot be read inside a timed run: these
    are the ones recorded with rocprofv3 --pmc for this exact workload, mode and loop
    (tools/pmc_summary.py -> profiles/*pmc_summary*.json); None for anything else."""
    if world != 1:
        return None
    for rec in _latest_profile("*pmc_summary*.json"):
        for entry in rec.get("kernels", []):
            if entry.get("workload") == workload and entry.get("arithmetic_mode") == mode and entry.get("loop") == loop:
                return entry.get("figures")
    return None


def recorded_traffic(workload, mode, world):
    """HBM bytes per launch of the render kernel.  Counters cannot be read from inside a timed run:
    the figure is the one measured with rocprofv3 --pmc passes for this exact workload, kept under
    profiles/ together with how it was taken; None for any other workload."""
    if world != 1:
        return None
    for rec in _latest_profile("*hbm_traffic.json"):
        if rec.get("workload") == workload and rec.get("arithmetic_mode") == mode:
            return rec.get("bytes_per_launch")
        return None
    return None


if __name__ == "__main__":
    main()
