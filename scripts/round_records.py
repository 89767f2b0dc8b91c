#!/usr/bin/env python
"""Turn the output of scripts/rounds/r06/gpu_round_r06.sh (gpurun_out/<tag>/) into the tracked records of the round
(profiles/<round>_*): the bench line, the headline kernel's summary + latest_traffic.json, the executed-FP64 record, the clock
record from the wave trace, the device-code record (only when the GPU suite of that pass was green) and the register table of
the shipped library.        python scripts/round_records.py r06h [--round r06]"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--round", default="r06")
    args = ap.parse_args()
    rnd, src, prof = args.round, os.path.join(ROOT, "gpurun_out", args.tag), os.path.join(ROOT, "profiles")
    tests = open(os.path.join(src, "tests.log")).read()
    m = re.search(r"(\d+) passed", tests)
    green = bool(m) and " failed" not in tests and " error" not in tests.lower()
    print("GPU suite:", tests.strip().splitlines()[-1] if tests.strip() else "?", "-> green" if green else "-> NOT green")
    line = open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(prof, "bench_%s_n1_local.json" % rnd), "w").write(line + "\n")
    full = os.path.join(src, "bench_full_n1.json")
    if os.path.exists(full):
        shutil.copy(full, os.path.join(prof, "bench_%s_n1_full.json" % rnd))
    summ = os.path.join(ROOT, "gpurun_out", "%s_fast800_w8_summary.json" % rnd)
    shutil.copy(summ, os.path.join(prof, "%s_fast800_w8_summary.json" % rnd))
    traffic = json.load(open(os.path.join(ROOT, "gpurun_out", "latest_traffic.json")))
    traffic["round"] = rnd
    json.dump(traffic, open(os.path.join(prof, "latest_traffic.json"), "w"), indent=1)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "fp64_executed.py"),
                           os.path.join(prof, "%s_fast800_w8_summary.json" % rnd),
                           os.path.join(prof, "%s_fast800_fp64_executed.json" % rnd)], stdout=subprocess.DEVNULL)
    wt = os.path.join(src, "wave_trace.txt")
    if os.path.exists(wt) and os.path.getsize(wt) > 200:
        text = open(wt).read()
        shutil.copy(wt, os.path.join(prof, "%s_fast800_wave_trace.txt" % rnd))
        life = [float(v) for v in re.search(r"wave life us: min ([\d.]+) median ([\d.]+) p90 [\d.]+ max ([\d.]+)", text).groups()]
        cyc = re.search(r"wave cycles: min (\d+) median (\d+) max (\d+)\s+-> clock ([\d.]+) GHz", text).groups()
        waves = int(re.search(r"waves traced (\d+)", text).group(1))
        ck = {"what": "engine clock the headline kernel (st_fast_800_w8, 1-hour clip, 800/400) runs at: median of cycles / life time over "
                      "the waves of one launch (scripts/phase_timing.py with the -DPAA_F800_TRACE build) after 3 s of untimed launches, "
                      "scripts/rounds/%s/gpu_round_%s.sh" % (rnd, rnd),
              "source": "profiles/%s_fast800_wave_trace.txt" % rnd, "kernel": "st_fast_800_w8",
              "sustained_clock_ghz": float(cyc[3]), "data_sheet_clock_ghz": 2.4,
              "wave_cycles_median": int(cyc[1]), "wave_cycles_min": int(cyc[0]), "wave_cycles_max": int(cyc[2]),
              "wave_life_us_min": life[0], "wave_life_us_median": life[1], "wave_life_us_max": life[2], "waves": waves,
              "frames": traffic["frames"], "measured_in_round": rnd}
        json.dump(ck, open(os.path.join(prof, "%s_fast800_clock.json" % rnd), "w"), indent=1)
    if green:
        shutil.copy(os.path.join(src, "device_code.json"), os.path.join(prof, "%s_device_code.json" % rnd))
    else:
        print("device-code record NOT updated: the GPU suite of this pass was not green")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "resource_usage.py"), "--json",
                           os.path.join(prof, "%s_resource_usage.json" % rnd)], stdout=subprocess.DEVNULL)
    print("records written for", rnd)


if __name__ == "__main__":
    main()
