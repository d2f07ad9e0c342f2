"""Developer tool: static instruction budget of one kernel, by source function.

Reads an assembly listing made with line tables, e.g.
    hipcc <flags of the Makefile> --cuda-device-only -gline-tables-only -S caster_strict.hip -o strict.s
and attributes every instruction of the named kernel to the source function whose body holds the
line of its innermost .loc.  Prints vector (VALU), matrix, scalar, LDS, memory and scratch
instruction counts per function.  Static counts: multiply by how often a region runs per wave trip
(tools/stats.py gives those) to get the dynamic budget.

usage: python tools/asm_budget.py strict.s 'render_kernel_strictILi2ELb0' [--lines]
"""
import collections
import os
import re
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "craytracer_amd", "csrc")


def function_starts(path):
    """(line, name) of every function definition in a source file, by a simple pattern."""
    out = []
    pat = re.compile(r"^(?:template\s*<[^>]*>\s*)?(?:CDEV|static inline|__global__|__device__)[^;(]*?\b([A-Za-z_][A-Za-z_0-9]*)\s*\(")
    pending_template = False
    with open(path) as f:
        for n, line in enumerate(f, 1):
            s = line.strip()
            if s.startswith("template"):
                pending_template = True
            m = pat.match(s)
            if m:
                out.append((n, m.group(1)))
            elif s.startswith("void CASTER_NAME(") or s.startswith("hipError_t CASTER_NAME("):
                out.append((n, re.sub(r"\W+", "_", s.split("(")[1])))
            pending_template = False
    return out


def classify(mn):
    if mn.startswith("v_mfma") or mn.startswith("v_smfma"):
        return "mfma"
    if mn.startswith("v_"):
        return "valu"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith("scratch_"):
        return "scratch"
    if mn.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    return "other"


def main():
    asm, kernel = sys.argv[1], sys.argv[2]
    by_line = "--lines" in sys.argv
    files = {}
    starts = {}
    counts = collections.defaultdict(lambda: collections.Counter())
    cur = None
    inside = False
    loc_re = re.compile(r"^\s*\.loc\s+(\d+)\s+(\d+)")
    file_re = re.compile(r'^\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"')
    ins_re = re.compile(r"^\s+([a-z][a-z0-9_]+)\b")
    with open(asm) as f:
        for line in f:
            m = file_re.match(line)
            if m:
                files[int(m.group(1))] = m.group(3)
                continue
            if not inside:
                if line.startswith("_Z") and kernel in line and line.rstrip().split(":")[0].endswith("E") or \
                        (line.startswith("_Z") and kernel in line.split(":")[0]):
                    inside = True
                continue
            if line.startswith(".Lfunc_end"):
                break
            m = loc_re.match(line)
            if m:
                cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
                continue
            m = ins_re.match(line)
            if not m or line.lstrip().startswith((".", ";")):
                continue
            kind = classify(m.group(1))
            if kind == "other":
                continue
            fname, ln = cur if cur else ("?", 0)
            if by_line:
                key = f"{fname}:{ln}"
            else:
                if fname not in starts:
                    p = os.path.join(CSRC, fname)
                    starts[fname] = function_starts(p) if os.path.exists(p) else []
                name = "?"
                for s_line, s_name in starts[fname]:
                    if s_line <= ln:
                        name = s_name
                    else:
                        break
                key = f"{fname}:{name}" if starts[fname] else fname
            counts[key][kind] += 1
            if m.group(1) in ("v_mul_lo_u32", "v_mul_hi_u32", "v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f64", "v_rcp_f64", "v_rsq_f64",
                              "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32"):
                counts[key]["quarter"] += 1
            if "_f64" in m.group(1):
                counts[key]["f64"] += 1
    kinds = ["valu", "mfma", "salu", "lds", "vmem", "scratch", "quarter", "f64"]
    print(f"{'where':58s}" + "".join(f"{k:>9s}" for k in kinds))
    tot = collections.Counter()
    for key, c in sorted(counts.items(), key=lambda kv: -kv[1]["valu"]):
        print(f"{key:58s}" + "".join(f"{c[k]:9d}" for k in kinds))
        tot.update(c)
    print(f"{'total':58s}" + "".join(f"{tot[k]:9d}" for k in kinds))


if __name__ == "__main__":
    main()
