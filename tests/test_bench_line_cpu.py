"""bench.py's ONE JSON line must fit the driver's 8 018-character stdout tail whole (BENCH_r05.parsed was null --
the 20.5 KB line was cut).  These tests build FULL synthetic records -- every key the N = 1 and the N = 8 runs produce, verbose
`config.others` for every SHAPES entry included -- and push them through the same formatter bench.py prints with."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _others(n_extra=0):
    out = {}
    names = ["cfg3", "cfg4_shard", "step800_68rows", "cfg5_features", "cfg5_spectrogram", "cfg5_chromagram"] + \
        ["shape_%s_with_a_long_name" % k for k in bench.SHAPES] + ["extra_%d" % i for i in range(n_extra)]
    for i, k in enumerate(names):
        out[k] = {"frames_per_step": 143999 + i, "ms_per_step": 0.123456789 + i, "frames_per_s": 5.123456789e8 / (i + 1),
                  "algorithmic_bytes_per_frame": 1072, "achieved_GBps": 595.123456, "kernel": "spectrogram_tri_r19x29x2_%d" % i,
                  "hbm_frac": 0.0743123456, "workload": "x" * 140, "fs": 44100, "window": 1102, "step": 441,
                  "samples": "interleaved stereo int16"}
    return out


def _full_record(world):
    others = _others()
    res = {
        "metric": "short-term frames/sec (34-feat, 16 kHz, 50 ms/25 ms) + HBM GB/s vs peak",
        "value": 5.123456789e8 * world * 0.6, "unit": "frames/s", "n_gpus": world, "ranks": world, "steps": 200, "warmup": 100,
        "ms_per_step": 0.2801234567, "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None,
        "prewarm_seconds": 0.3, "dtype": "f64", "data": "synthetic (oracle/synth.py; SURVEY 8d seeds)",
        "config": {"workload": "cfg4: 100000 clips x 10 s in the job (64 distinct seeded clips, tiled), 12500 on this rank, 800/400, 34 rows",
                   "frames_per_step_job": 39900000, "frames_per_step_rank0": 4987500, "clips_in_job": 100000, "window": 800,
                   "step": 400, "rows": 34, "kernel": "st_fast_800_w8", "rows_computed_per_rank": 34, "input_buffers_rotated": 3,
                   "devices": ["0000:%02x:00.0" % (0x10 + 8 * r) for r in range(world)], "distinct_devices": world,
                   "rccl_ranks": world if world > 1 else None, "launched_by": "bench.py itself (no launcher environment)",
                   "multi_gpu": "contiguous clip ranges per rank (partition_by_frames), RCCL gather of the slabs to rank 0 "
                                "overlapped with the next step",
                   "others": others},
        "roofline": {"bound": "hbm", "achieved": 594.7199064173598, "peak": 8000.0, "unit": "GB/s", "frac": 0.07433998830216997,
                     "traffic": 162279539.61952803, "kernel": "st_fast_800_w8", "kernel_avg_ms": 0.2595624029636383,
                     "launches_timed": 50, "event_pair_every_nth_launch": 4, "algorithmic_bytes_per_frame": 1072,
                     "note": "n" * 120, "traffic_round": "r06", "traffic_stale": False, "traffic_from_profile": 162279539.6,
                     "traffic_over_algorithmic": 1.0512584639860685, "traffic_source": "s" * 150,
                     "fp64_valu": {"algorithmic_kflop_per_frame": 45.0, "achieved_tflops": 24.96492144475857, "peak_tflops": 78.6,
                                   "frac": 0.3176198657093966, "issued_kflop_per_frame_measured": 43.84342746198599,
                                   "issued_tflops": 24.323282721276684, "issued_frac": 0.3094565231714591, "issued_source": "i" * 100,
                                   "sustained_clock_ghz": 2.15, "peak_tflops_at_sustained_clock": 70.4125,
                                   "frac_at_sustained_clock": 0.354552408233745, "issued_frac_at_sustained_clock": 0.3454398398193032,
                                   "clock_source": "c" * 80}},
        "sustained_frames_per_s": 5.16e8, "sustained": {"seconds": 2.51, "steps": 9000, "frames_per_s": 5.16e8},
        "parity_check": {"status": "ok", "violations": 0, "entries": 4895966, "checker": "k" * 230,
                         "gate": "|d| <= 1e-4 |ref| + 1e-6 scale(row) + 1e-9", "max_abs_diff": 3.552713678800501e-14},
        "parity_spot_check": "ok",
        "host_to_host": {"10_min_clip": {"frames": 23999, "ms": 0.599431004957296, "frames_per_s": 40036300.7, "pcie_GBps": 42.92},
                         "1_hour_clip": {"frames": 143999, "ms": 2.99, "frames_per_s": 48139157.8, "pcie_GBps": 51.6},
                         "note": "h" * 110},
        "cpu_baseline": {"value": 45416.796355186554, "unit": "frames/s", "cores": 1, "kind": "port", "numpy_port": 18828.97,
                         "c_port": 45416.79, "cpu_model": "AMD EPYC 9575F 64-Core Processor", "host_cores": 256,
                         "sample": "s" * 150, "ports_vs_reference": "p" * 160,
                         "reference_cost_port": {"value": 2600.1234, "frames": 4000, "seconds": 1.54, "what": "w" * 60},
                         "reference": {"where": "measured in the build container, not on this host", "cpu_model": "Intel(R) Xeon(R) Processor @ 2.10GHz",
                                       "numpy": "2.2.6", "scipy": "1.15.3", "cores": 1, "source": "profiles/reference_cpu_r03.json",
                                       "frames_per_s": {"cfg2_60s_34rows": 2453.43, "cfg2_60s_68rows": 2564.0, "cfg3_sample": 2286.5},
                                       "ports_on_that_core": {"numpy_port_frames_per_s": 8176.0, "c_port_frames_per_s": 25335.9}},
                         "all_cores": {"value": 747936.0032377123, "unit": "frames/s", "cores": 16, "host_logical_cpus": 256,
                                       "cores_note": "n" * 70, "kind": "port (oracle/paa_oracle.c, one single-threaded process per core)",
                                       "frames": 3836784, "wall_seconds": 5.13, "sample": "s" * 100}},
    }
    if world > 1:
        res["config"].update({
            "frames_per_s": res["value"], "frames_per_s_without_gather": 4.0e9, "gather_bytes_per_step_into_root": 9496200000,
            "frames_per_s_mid_gather": 3.9e9, "mid_gather": {"frames_per_s": 3.9e9, "ms_per_step": 10.2, "what": "m" * 130},
            "gather_rows": "g" * 120,
            "expected_speedup_short_gather": {"speedup_over_one_gpu": 4.61, "ideal": world, "compute_s_per_step_one_gpu": 0.0771,
                                              "link_s_per_step": 0.0177, "link_GBps": 76.8, "bound": "xgmi link into the root",
                                              "note": "n" * 200}})
    configs = {"cfg2" if world == 1 else "cfg4_job": [5.1e8, 0.2801, 0.0743, "st_fast_800_w8"]}
    for k, e in others.items():
        configs[k] = bench.compact(e)
    res["configs_columns"] = ["frames_per_s", "ms_per_step", "hbm_frac", "kernel"]
    res["configs"] = configs
    return res


def test_n1_line_fits_the_driver_tail_and_keeps_the_contract():
    rec = _full_record(1)
    assert len(json.dumps(rec)) > 15000                       # the record itself is the size that broke BENCH_r05
    text = bench.format_line(rec, "gpurun_out/bench_full_n1.json")
    assert "\n" not in text and len(text) < 8000
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["config"]["workload"] and line["dtype"] == "f64" and line["n_gpus"] == 1
    assert "others" not in line["config"] and line["full_record"] == "gpurun_out/bench_full_n1.json"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert line["parity_check"]["status"] == "ok"
    # the table that ends the line still holds every BASELINE configuration
    assert text.rstrip("}").endswith("]") and list(line)[-1] == "configs"
    for k in ("cfg2", "cfg3", "cfg4_shard", "cfg5_features", "cfg5_spectrogram", "cfg5_chromagram"):
        assert len(line["configs"][k]) == 4


def test_n8_line_with_every_scaling_field_fits():
    rec = _full_record(8)
    text = bench.format_line(rec, "gpurun_out/bench_full_n8.json")
    assert len(text) < 8000
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    sc = line["scale"]
    for k in ("frames_per_s", "frames_per_s_without_gather", "frames_per_s_mid_gather", "rccl_ranks", "distinct_devices"):
        assert k in sc, k
    assert sc["expected_speedup_short_gather"]["speedup_over_one_gpu"] == 4.61
    assert line["n_gpus"] == 8 and line["ranks"] == 8 and len(line["config"]["devices"]) == 8
    # `scale` sits in the tail (after everything but the configs table)
    assert list(line)[-3:] == ["scale", "configs_columns", "configs"]


def test_an_oversized_record_is_trimmed_not_truncated():
    rec = _full_record(1)
    extra = _others(n_extra=120)
    for k, e in extra.items():
        rec["configs"][k] = bench.compact(e)
    text = bench.format_line(rec, None)
    assert len(text) < 8000
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    for k in ("cfg2", "cfg3", "cfg4_shard", "cfg5_features", "cfg5_spectrogram", "cfg5_chromagram"):
        assert k in line["configs"]
