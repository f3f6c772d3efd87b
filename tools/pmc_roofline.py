#!/usr/bin/env python3
"""Reduce the per-entry rocprofv3 --pmc passes of tools/roofline_entry.py (run by tools/run_pmc_roofline.sh) into
  profiles/r03_pmc_mfma.csv  -- matrix-pipe / vector-ALU utilisation and clock of each roofline kernel
  profiles/traffic.json      -- HBM bytes per launch (FETCH_SIZE + WRITE_SIZE)

Corrections (/opt/skills/guides/MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of a wide (16 B/lane) coalesced read stream -- doubled for the kernels whose loads are float4 / LDS-DMA
streams; WRITE_SIZE is exact for 16-B stores and float atomics.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD,
SQ_BUSY_CU_CYCLES cycles per CU (4 SIMDs), SQ_WAVE_CYCLES / SQ_WAIT_* quad-cycles, GRBM_GUI_ACTIVE the sum over 8 XCDs.

usage: pmc_roofline.py <root dir with <entry>/{mfma,fetch,write}> <out csv> <out traffic.json>"""
import csv, glob, json, os, sys

ENTRIES = {   # entry function -> (kernel-name substrings whose dispatches belong to it, FETCH correction, roofline key prefix)
    "conv2_dense_stage4": (["gemm_x3_kernel", "gemm_nt_kernel"], 2.0, "gemm_x3_kernel (conv2 dense half forward"),
    "per_point_stage4": (["gemm_rp_kernel", "gemm_x3_kernel", "gemm_nt_kernel"], 2.0, "gemm_rp_kernel<128> (per-point GEMM"),
    "conv2_dense_dx_stage4": (["gemm_x3_kernel", "gemm_nt_kernel"], 2.0, "gemm_x3_kernel<WT> = pdgn_gemm_nn (conv2 dense half input gradient"),
    "weight_grad_stage4": (["gemm_x3_kernel", "gemm_tn_kernel"], 2.0, "gemm_x3_kernel<AT,WT> = pdgn_gemm_tn_big"),
    "bn_act_backward_stage4": (["cl_bwd_reduce_kernel", "cl_bwd_apply_kernel"], 2.0, "cl_bwd_reduce + cl_bwd_apply"),
    "window_gather_sum_stage4": (["wgs_fwd_xcd_kernel"], 2.0, "wgs_fwd_xcd_kernel"),
    "feature_knn_stage4": (["feat_knn_pc_kernel"], 2.0, "feat_knn_pc_kernel<64>"),
    "knn3_largest": (["knn3_wave4_kernel"], 1.0, "knn3_wave4_kernel"),
}
TIMED = 5        # launches per entry that count (roofline_entry.py: 2 warm-up + 5)


def rows_of(d):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def per_launch(d, subs):
    """{counter: sum over the entry's kernels of the per-launch average over the last TIMED launches}, duration likewise."""
    rows = [r for r in rows_of(d) if any(s in r["Kernel_Name"] for s in subs)]
    by = {}
    for r in rows:
        by.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(
            (int(r["Dispatch_Id"]), float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    tot, dur = {}, {}
    for (kn, cn), vals in by.items():
        vals.sort()
        # an entry may launch its kernel more than once per call (stream-K tail): launches per call = count / (2 + TIMED)
        per_call = max(1, round(len(vals) / (2 + TIMED)))
        last = vals[-TIMED * per_call:]
        tot[cn] = tot.get(cn, 0.0) + sum(v for _, v, _ in last) / TIMED
        dur[cn] = dur.get(cn, 0.0) + sum(t for _, _, t in last) / TIMED
    return tot, dur


def main():
    root, out_csv, out_json = sys.argv[1:4]
    traffic = {"_meta": {"batch": 35, "base_points": 128, "arch": "gfx950", "recorded": "round 6, profiles/r06_pmc_mfma.csv run"}}
    lines = ["entry,kernels,us_per_launch,mfma_busy_frac_of_simd_cycles,valu_insts_per_launch,wave_wait_any_frac,"
             "wave_wait_inst_frac,clock_GHz,fetch_MB,write_MB,traffic_MB"]
    for entry, (subs, corr, key) in ENTRIES.items():
        m, dm = per_launch(os.path.join(root, entry, "mfma"), subs)
        f, _ = per_launch(os.path.join(root, entry, "fetch"), subs)
        w, _ = per_launch(os.path.join(root, entry, "write"), subs)
        us = dm.get("SQ_BUSY_CU_CYCLES", 0.0)
        mfma = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(4.0 * m.get("SQ_BUSY_CU_CYCLES", 1.0), 1.0)
        wc = max(m.get("SQ_WAVE_CYCLES", 1.0), 1.0)
        clock = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / max(us, 1e-9) / 1e3
        fb, wb = f.get("FETCH_SIZE", 0.0) * 1024 * corr, w.get("WRITE_SIZE", 0.0) * 1024
        traffic[key] = fb + wb
        lines.append("%s,%s,%.1f,%.3f,%.4g,%.3f,%.3f,%.2f,%.1f,%.1f,%.1f" % (
            entry, "+".join(subs), us, mfma, m.get("SQ_INSTS_VALU", 0.0), m.get("SQ_WAIT_ANY", 0.0) / wc,
            m.get("SQ_WAIT_INST_ANY", 0.0) / wc, clock, fb / 1e6, wb / 1e6, (fb + wb) / 1e6))
    open(out_csv, "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(out_json, "w"), indent=1)
    print("\n".join(lines))
    # config C5's kernel: vector-ALU issue (SQ_INSTS_VALU per launch against the instruction model of bench.py::EMD_LOOPS,
    # issue utilisation = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, waves parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES)
    ed = os.path.join(root, "emd_cost_c5", "valu")
    if os.path.isdir(ed):
        m, dm = per_launch(ed, ["emd_cost_kernel"])
        us = dm.get("SQ_BUSY_CU_CYCLES", 0.0)
        wc = max(m.get("SQ_WAVE_CYCLES", 1.0), 1.0)
        el = ["kernel,us_per_launch,valu_insts_per_launch,valu_insts_per_dense_element,busy_cu_cycles,wave_cycles_quad,"
              "valu_active_frac_of_wave_cycles,wave_wait_any_frac,wave_wait_inst_frac,clock_GHz",
              "emd_cost_kernel,%.1f,%.5g,%.3f,%.5g,%.5g,%.3f,%.3f,%.3f,%.2f" % (
                  us, m.get("SQ_INSTS_VALU", 0.0), m.get("SQ_INSTS_VALU", 0.0) * 64.0 / (512.0 * 19 * 2048 * 2048),
                  m.get("SQ_BUSY_CU_CYCLES", 0.0), wc, m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, m.get("SQ_WAIT_ANY", 0.0) / wc,
                  m.get("SQ_WAIT_INST_ANY", 0.0) / wc, m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / max(us, 1e-9) / 1e3)]
        open(os.path.join(os.path.dirname(out_csv), "r03_eval_pmc.csv"), "w").write("\n".join(el) + "\n")
        print("\n".join(el))


if __name__ == "__main__":
    main()
