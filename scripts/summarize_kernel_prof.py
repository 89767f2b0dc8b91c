"""Fold the rocprofv3 (rocpd sqlite) outputs of scripts/profile_kernel.sh into one JSON summary: kernel-trace statistics,
per-dispatch counters of the case's feature kernel, HBM traffic (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
gfx950, WRITE_SIZE as reported) and its ratio to the algorithmic bytes of SURVEY 8d."""
import json
import os
import sqlite3
import sys


def main(prof_dir, out_path):
    out = {"source": os.path.basename(prof_dir.rstrip("/"))}
    line = None
    for ln in open(os.path.join(prof_dir, "bench_trace.log")):
        if ln.startswith("{"):
            line = json.loads(ln)
    if line is None:
        raise SystemExit("no result line in %s/bench_trace.log" % prof_dir)
    out["run_under_trace"] = line
    stem = line["kernel"].replace("_w8", "")
    like = {"st_reg_29x19": "%st_reg_kernel%", "spectrogram_reg_29x19": "%st_reg_kernel%", "chromagram_reg_29x19": "%st_reg_kernel%",
            "st_generic": "%st_generic_kernel%"}.get(stem, "%st_ct_kernel%" if "_ct_" in stem else
                                                    "%st_tri_kernel%" if "_tri_" in stem else "%" + stem + "%")
    if stem.startswith("st_fast_800"):
        like = "%st_fast_800_kernel%"
    if "_blu_" in stem:
        like = "%st_blu_kernel%"
    if line["case"] == "mid_stats":
        like = "%mid_stats_kernel%"
    second = None
    if "_wgr_" in stem:                           # the fused three-pass kernel of the 1 s windows: one launch (+ the delta rows)
        like = "%wgr_kernel%"
    if "wg_lds_fft" in stem:                      # two kernels per step: the spectra of all frames, then their features
        like, second = "%wg_spectrum_kernel%", "%wg_feat_kernel%"
    if "wg_split_fft" in stem:                    # split transforms: sub-transform tasks, then the features (+ a small time-domain kernel)
        like, second = "%wg_split_kernel%", "%wg_feat_kernel%"
    if "_wgs_" in stem:                           # real-input split on register passes (44 100 / 22 050 samples), then the features
        like, second = "%wgs_kernel%", "%wg_feat_kernel%"
    if stem == "big_window_hbm_passes":
        like = "%big_pass_kernel%"
    out["kernel_like"] = like
    if line["case"] == "mid_stats":
        # the profiled kernel is mid_stats_kernel alone: it reads the (68, T) short-term slabs once and writes (136, M)
        # (MidTermFeatures.py:110-126; 1000 clips x 1199 frames, mid-term window and step of 40 frames -> M = 30)
        clips, T, M = 1000, line["frames"] // 1000, -(-(line["frames"] // 1000) // 40)
        line = dict(line, algorithmic_bytes_per_launch=8 * clips * (68 * T + 136 * M),
                    algorithmic_bytes_note="mid_stats_kernel only: 8 B x (68 T + 136 M) per clip")
        out["run_under_trace"] = line
    con = sqlite3.connect(os.path.join(prof_dir, "trace", "trace_results.db"))
    out["kernel_trace_stats"] = [dict(name=r[0][:160], calls=r[1], total_us=r[2], avg_us=r[3], pct=r[4])
                                 for r in con.execute("select * from top_kernels")][:8]
    pm = {}
    for n in sorted(os.listdir(prof_dir)):
        db = os.path.join(prof_dir, n, "pmc_results.db")
        if not os.path.exists(db):
            continue
        con = sqlite3.connect(db)
        nd = con.execute("select count(distinct dispatch_id) from pmc_events where name like ?", (like,)).fetchone()[0]
        for r in con.execute("select counter_name, sum(counter_value) from pmc_events where name like ? group by counter_name", (like,)):
            pm[r[0]] = {"per_dispatch": r[1] / max(nd, 1), "dispatches": nd}
    out["pmc"] = pm
    if second:
        pm2 = {}
        for n in sorted(os.listdir(prof_dir)):
            db = os.path.join(prof_dir, n, "pmc_results.db")
            if not os.path.exists(db):
                continue
            con = sqlite3.connect(db)
            nd = con.execute("select count(distinct dispatch_id) from pmc_events where name like ?", (second,)).fetchone()[0]
            for r in con.execute("select counter_name, sum(counter_value) from pmc_events where name like ? group by counter_name", (second,)):
                pm2[r[0]] = {"per_dispatch": r[1] / max(nd, 1), "dispatches": nd}
        out["pmc_second_kernel"] = {"kernel_like": second, "counters": pm2}
        # HBM traffic of the step = both kernels
        for key in ("FETCH_SIZE", "WRITE_SIZE"):
            if key in pm and key in pm2:
                pm[key + "_both_kernels"] = {"per_dispatch": pm[key]["per_dispatch"] + pm2[key]["per_dispatch"], "dispatches": pm[key]["dispatches"]}
    avg = [k for k in out["kernel_trace_stats"] if like.strip("%") in k["name"]]
    if avg:
        out["kernel_avg_us"] = avg[0]["avg_us"]
        out["achieved_GBps_kernel"] = line["algorithmic_bytes_per_launch"] / (avg[0]["avg_us"] * 1e-6) / 1e9
        out["hbm_frac"] = out["achieved_GBps_kernel"] / 8000.0
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
        fk, wk = ("FETCH_SIZE_both_kernels", "WRITE_SIZE_both_kernels") if "FETCH_SIZE_both_kernels" in pm else ("FETCH_SIZE", "WRITE_SIZE")
        fetch = pm[fk]["per_dispatch"] * 1024.0 * 2.0
        write = pm[wk]["per_dispatch"] * 1024.0
        out["traffic"] = {"fetch_bytes_corrected_x2": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
                          "algorithmic_bytes_per_launch": line["algorithmic_bytes_per_launch"],
                          "traffic_over_algorithmic": (fetch + write) / line["algorithmic_bytes_per_launch"]}
    if "SQ_LDS_BANK_CONFLICT" in pm and "SQ_LDS_IDX_ACTIVE" in pm and pm["SQ_LDS_IDX_ACTIVE"]["per_dispatch"] > 0:
        out["lds_bank_conflict_ratio"] = pm["SQ_LDS_BANK_CONFLICT"]["per_dispatch"] / pm["SQ_LDS_IDX_ACTIVE"]["per_dispatch"]
    if "SQ_ACTIVE_INST_VALU" in pm and "SQ_BUSY_CYCLES" in pm and pm["SQ_BUSY_CYCLES"]["per_dispatch"] > 0:
        # as DESIGN 5: ACTIVE_INST_VALU x 4 / 1024 SIMDs / (BUSY_CYCLES / 32)
        out["valu_issue_fraction"] = pm["SQ_ACTIVE_INST_VALU"]["per_dispatch"] * 4.0 / 1024.0 / (pm["SQ_BUSY_CYCLES"]["per_dispatch"] / 32.0)
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in out if k not in ("pmc", "kernel_trace_stats")}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
