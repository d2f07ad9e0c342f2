"""Turns the rocprofv3 --pmc passes of tools/pmc.sh into the figures bench.py quotes beside its
roofline (profiles/*pmc_summary*.json) and the HBM traffic record (profiles/*hbm_traffic.json).

usage: python tools/pmc_summary.py <pmc dir> <workload string> <mode> <loop> [<pmc dir> <workload> <mode> <loop> ...] --out <prefix>
(the kernel's duration and its wave trips are read from the first pass's log, <pmc dir>/p1.log, a tools/quick_bench.py line)

Every figure is a ratio of counters summed over the dispatch records of the one render-kernel launch
of a pass (the 8 XCDs report separately):
  simd_cycles            GRBM_GUI_ACTIVE (summed over XCDs) x 32 CUs x 4 SIMDs
  valu_issue_busy        SQ_ACTIVE_INST_VALU x 4 / simd_cycles' wave-level equivalent: the share of SIMD cycles in
                         which a vector instruction was being issued (SQ_ACTIVE_INST_VALU counts quad-cycles per SIMD)
  valu_insts_per_trip    SQ_INSTS_VALU / wave trips (a trip traces 128 ray slots from the second ray of a path on; the
                         instructions of the camera-ray batches are in the numerator as well)
  valu_insts_per_ray     SQ_INSTS_VALU / rays of the launch (wave-level instructions: 64 lanes each)
  active_lane_frac       SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
  fp32_share_of_valu     (ADD + MUL + FMA + TRANS) / SQ_INSTS_VALU
  fp32_executed_tflops   SQ_INSTS_VALU_FLOPS_FP32 x 64 lanes / kernel seconds (an upper bound: lanes masked off count)
  mfma_busy_frac         SQ_VALU_MFMA_BUSY_CYCLES / simd_cycles
  lds_conflict_share     SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS
  hbm bytes              2 x FETCH_SIZE KB (gfx950 reports half of wide reads) + WRITE_SIZE KB, per launch
Both output files carry the hash of the kernel sources in the tree (craytracer_amd/buildinfo.py).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from craytracer_amd.buildinfo import kernel_source_hash  # noqa: E402


def read_pass_dir(path):
    tot = defaultdict(float)
    names = set()
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "render_kernel" not in row["Kernel_Name"]:
                    continue
                names.add(row["Kernel_Name"].split("(")[0].replace("void ", ""))
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
    return dict(tot), sorted(names)


def figures(c, kernel_ms, trips, rays=0.0, batches=0.0):
    simd_cycles = c["GRBM_GUI_ACTIVE"] * 32 * 4
    valu = c["SQ_INSTS_VALU"]
    fp32 = c.get("SQ_INSTS_VALU_ADD_F32", 0) + c.get("SQ_INSTS_VALU_MUL_F32", 0) + c.get("SQ_INSTS_VALU_FMA_F32", 0) + c.get("SQ_INSTS_VALU_TRANS_F32", 0)
    out = {
        "simd_cycles": simd_cycles,
        "shader_clock_ghz": round(c["GRBM_GUI_ACTIVE"] / 8 / (kernel_ms * 1e-3) * 1e-9, 3),
        "valu_insts": valu,
        "valu_insts_per_trip": round(valu / trips, 1) if trips else None,
        "valu_insts_per_ray": round(valu / rays, 3) if rays else None,
        "rays": rays,
        "camera_ray_batches": batches,
        "simd_cycles_per_valu_inst": round(simd_cycles / valu, 3),
        "active_lane_frac": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 4),
        "fp32_share_of_valu": round(fp32 / valu, 4),
        "fp32_executed_tflops": round(c["SQ_INSTS_VALU_FLOPS_FP32"] * 64 / (kernel_ms * 1e-3) * 1e-12, 2),
        "fp32_executed_of_vector_peak": round(c["SQ_INSTS_VALU_FLOPS_FP32"] * 64 / (kernel_ms * 1e-3) * 1e-12 / 157.3, 4),
        "salu_insts": c.get("SQ_INSTS_SALU"),
        "lds_insts": c.get("SQ_INSTS_LDS"),
        "lds_conflict_share": round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_ACTIVE_INST_LDS"], 4) if c.get("SQ_ACTIVE_INST_LDS") else None,
        "waves": c.get("SQ_WAVES"),
    }
    if c.get("SQ_INSTS_MFMA"):
        out["mfma_insts"] = c["SQ_INSTS_MFMA"]
        out["mfma_insts_per_trip"] = round(c["SQ_INSTS_MFMA"] / trips, 1) if trips else None
        out["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4)
        out["mfma_executed_tflops"] = round(c["SQ_INSTS_MFMA"] * 64 * 512 / (kernel_ms * 1e-3) * 1e-12, 1)   # 32768 flop each
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["hbm_bytes_per_launch"] = int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)
    return out


def main():
    args = sys.argv[1:]
    prefix = args[args.index("--out") + 1]
    args = args[:args.index("--out")]
    kernels = []
    import re

    for k in range(0, len(args), 4):
        path, workload, mode, loop = args[k:k + 4]
        line = [ln for ln in open(path + "/p1.log") if " kernel " in ln and " trips " in ln][-1]
        kernel_ms = re.search(r"kernel ([0-9.]+) ms", line).group(1)
        trips = re.search(r"trips ([0-9]+)", line).group(1)
        rays = re.search(r"rays ([0-9]+)", line).group(1)
        batches = re.search(r"batches ([0-9]+)", line)
        c, names = read_pass_dir(path)
        fig = figures(c, float(kernel_ms), float(trips), float(rays), float(batches.group(1)) if batches else 0.0)
        kernels.append({"workload": workload, "arithmetic_mode": mode, "loop": loop, "kernel": names, "kernel_ms": float(kernel_ms),
                        "wave_trips": float(trips), "figures": fig, "raw": c})
        if loop.startswith("matrix") and "hbm_bytes_per_launch" in fig:
            with open(prefix + "_hbm_traffic.json", "w") as f:
                json.dump({"kernel_source_hash": kernel_source_hash(), "workload": workload, "arithmetic_mode": mode, "kernel": names, "FETCH_SIZE_KB": c["FETCH_SIZE"],
                           "WRITE_SIZE_KB": c["WRITE_SIZE"], "bytes_per_launch": fig["hbm_bytes_per_launch"],
                           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc.sh), one launch each, summed "
                                  "over XCDs; bytes = 2 x FETCH_SIZE x 1024 (gfx950 reports half of wide reads) + WRITE_SIZE x 1024; the writes are "
                                  "64-bit integer atomics, which WRITE_SIZE tallies as 8 bytes each when a wave's lanes add to consecutive words "
                                  "and as a 32-byte request otherwise (tools/atomic_traffic.hip)"}, f, indent=1)
    with open(prefix + "_pmc_summary.json", "w") as f:
        json.dump({"how": "rocprofv3 --pmc, separate passes (tools/pmc.sh), one launch of the render kernel per pass; "
                          "figures derived by tools/pmc_summary.py; kernel_source_hash = craytracer_amd.buildinfo.kernel_source_hash() "
                          "of the tree the passes ran on (bench.py quotes the figures only for that tree)",
                   "kernel_source_hash": kernel_source_hash(), "kernels": kernels}, f, indent=1)
    for k in kernels:
        print(k["workload"], k["arithmetic_mode"], k["loop"], json.dumps(k["figures"]))


if __name__ == "__main__":
    main()
