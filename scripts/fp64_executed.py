"""Executed FP64 work of the headline kernel from MEASURED instruction counts: the ISA histogram of the
kernel's loop body gives the share of FP64 instructions and their flop weights (fma 2, add / mul / min / max / rsq / rcp ...
1 per lane), rocprofv3's SQ_INSTS_VALU gives the wave-instructions the launch really issued; every wave-instruction is 64
lane-operations whether the lanes carry useful data or not, so two numbers come out:
    issued  = SQ_INSTS_VALU x f64 share x mean flop weight x 64      (what the FP64 pipes were asked to do)
    useful  = the same x the active-lane share of the stages (estimated from the stage structure, stated below)
usage (build container): python scripts/fp64_executed.py profiles/r04_fast800_w8_summary.json  ->  profiles/r04_fast800_fp64_executed.json
(compiles csrc/family_fast.hip with -save-temps into /tmp to get the listing of st_fast_800_kernel<400,0,1,8>)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP64_PEAK = 78.6e12

FLOPS = {"v_fma_f64": 2, "v_fmac_f64": 2}      # every other v_*_f64 arithmetic instruction: 1 per lane
NO_FLOP = ("v_cmp", "v_cndmask", "v_mov", "v_cvt", "v_readlane", "v_frexp", "v_ldexp", "v_trunc", "v_floor", "v_rndne", "v_ceil")


def loop_body(lines, key):
    start = end = None
    for n, l in enumerate(lines):
        if start is None and re.match(r"^(_Z\w*%s\w*):" % re.escape(key), l):
            start = n
        elif start is not None and l.startswith(".Lfunc_end"):
            end = n
            break
    body = lines[start:end]
    labels = {m.group(1): n for n, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    best = (0, 0, 0)
    for n, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < n and n - labels[m.group(1)] > best[0]:
            best = (n - labels[m.group(1)], labels[m.group(1)], n)
    return body[best[1]:best[2] + 1]


def main(summary_path, out_path):
    summ = json.load(open(summary_path))
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(ROOT, "pyaudioanalysis_amd", "csrc", "family_fast.hip")
        sys.path.insert(0, ROOT)
        from pyaudioanalysis_amd import _build          # the unit's own flags (the ISA histogram must be the shipped kernel's)
        subprocess.run(["hipcc"] + _build.BASE_FLAGS + _build.UNIT_FLAGS.get("family_fast.hip", []) + ["-O3",
                        "-c", src, "-o", os.path.join(d, "x.o"), "-save-temps"], cwd=d, check=True, capture_output=True)
        lines = open(os.path.join(d, "family_fast-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    body = loop_body(lines, "st_fast_800_kernelILi400ELi0ELi1ELi8E")
    ops = collections.Counter()
    for l in body:
        m = re.match(r"^\s+([a-z_0-9]+)\b", l)
        if m and not l.strip().startswith((".", ";")):
            ops[m.group(1)] += 1
    valu = sum(n for op, n in ops.items() if op.startswith("v_"))
    f64 = {op: n for op, n in ops.items() if op.startswith("v_") and "f64" in op}
    flop_lane = 0
    f64_arith = 0
    for op, n in f64.items():
        base = re.sub(r"_e(32|64)$|_dpp$", "", op)
        if base.startswith(NO_FLOP):
            continue
        f64_arith += n
        flop_lane += n * FLOPS.get(base, 1)
    pm = summ["pmc_feature_kernel"]
    insts = pm["SQ_INSTS_VALU"]["per_dispatch"]
    frames = summ["bench_line_under_trace"]["config"]["frames_per_step_rank0"]
    k_ms = summ["bench_line_under_trace"]["roofline"]["kernel_avg_ms"]
    # static loop body = one quad of one wave; the dynamic count per quad (halo quads run a shorter path) comes from the counter
    quads = insts / valu
    issued = quads * flop_lane * 64
    out = {
        "kernel": "st_fast_800_kernel<400,0,1,8>", "source_summary": os.path.basename(summary_path),
        "static_loop_body": {"valu": valu, "valu_f64": sum(f64.values()), "valu_f64_arithmetic": f64_arith,
                             "flop_per_lane_per_quad": flop_lane, "top_f64": dict(collections.Counter(f64).most_common(8))},
        "SQ_INSTS_VALU_per_launch": insts, "loop_bodies_per_launch_equivalent": quads, "frames": frames,
        "issued_fp64_flop_per_launch": issued, "issued_kflop_per_frame": issued / frames / 1e3,
        "kernel_avg_ms": k_ms, "issued_tflops": issued / (k_ms * 1e-3) / 1e12,
        "issued_frac_of_fp64_peak": issued / (k_ms * 1e-3) / FP64_PEAK,
        "note": "issued = wave-instructions x 64 lanes: idle lanes included (13 of 16 lanes work in pass 2, 50 of 64 in the "
                "time-domain stage); SURVEY 8d's algorithmic estimate is 35-55 kflop per frame",
    }
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
