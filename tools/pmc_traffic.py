#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/roofline_only.py into
profiles/traffic.json: HBM bytes per launch of each roofline kernel.

Corrections (/opt/skills/guides/MI355X_MICROARCH.md, section HBM): both counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced read
stream, so it is doubled for the kernels whose loads are float4 streams; WRITE_SIZE is exact
for 16-B streaming stores and float atomics.

usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [<summary.txt>]
"""
import csv, glob, json, os, sys

KERNELS = {   # roofline entry prefix -> (kernel-name substrings, FETCH_SIZE correction)
    "gemm_tn_kernel": (["gemm_tn_kernel"], 2.0),
    "pdgn_gemm_tn, hand-written": (["gemm_tn_kernel"], 2.0),
    "searched library GEMM": (["Cijk_"], 2.0),
    "cl_bwd_reduce + cl_bwd_apply": (["cl_bwd_reduce_kernel", "cl_bwd_apply_kernel"], 2.0),
    "wgs_fwd_xcd_kernel": (["wgs_fwd_xcd_kernel"], 2.0),
    "feat_knn_pc_kernel<128>": (["feat_knn_pc_kernel"], 2.0),
    "knn3_wave_kernel": (["knn3_wave_kernel"], 1.0),
}


def collect(d, counter):
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            per.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return per


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out, lines = {}, []
    for key, (subs, corr) in KERNELS.items():
        fb = wb = 0.0
        for sub in subs:
            for name, vals in fetch.items():
                if sub in name:
                    fb += sum(vals) / len(vals) * 1024 * corr
            for name, vals in write.items():
                if sub in name:
                    wb += sum(vals) / len(vals) * 1024
        out[key] = fb + wb
        lines.append("%-34s FETCH %10.1f MB (x%.0f corr)  WRITE %10.1f MB  total %10.1f MB/launch" % (key, fb / 1e6, corr, wb / 1e6, (fb + wb) / 1e6))
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
