#!/usr/bin/env python3
"""Summarise the counter CSVs written by tools/pmc.sh into a markdown table per kernel + the HBM traffic JSON bench.py reads.
usage: python tools/pmc_summary.py gpurun_out/pmc_<tag> <out.md> [--json kernel_tag out.json particles side model]"""
import collections
import csv
import glob
import json
import os
import sys

TAGS = ("g2p2g_binned", "p2g_wide", "p2g_binned_split", "p2g_binned", "g2p_binned", "tv_scale")


def main():
    root, out_md = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            for tag in TAGS:
                if tag in k:
                    acc[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    break
    lines = []
    summary = {}
    for k in acc:
        m = {c: sum(v) / len(v) for c, v in acc[k].items()}
        # gfx950: FETCH_SIZE tallies 128-B requests as 64 B -> reads doubled; units are KiB (MI355X_MICROARCH.md, HBM section)
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            m["hbm_read_bytes_corrected"] = m["FETCH_SIZE"] * 1024 * 2
            m["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
            m["hbm_bytes_per_launch"] = m["hbm_read_bytes_corrected"] + m["hbm_write_bytes"]
        if "SQ_INSTS_VALU" in m and "GRBM_GUI_ACTIVE" in m:
            simd_cycles = m["GRBM_GUI_ACTIVE"] / 8 * 1024  # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs
            m["valu_busy_frac (SQ_INSTS_VALU x 4 cyc / SIMD cycles)"] = m["SQ_INSTS_VALU"] * 4 / simd_cycles
        summary[k] = m
        lines += ["## %s" % k, "| counter | value |", "|---|---|"]
        lines += ["| %s | %.6g |" % (c, m[c]) for c in sorted(m)]
        lines.append("")
    open(out_md, "a").write("\n".join(lines) + "\n")
    if "--json" in sys.argv:
        i = sys.argv.index("--json")
        tag, out_json, particles, side, model = sys.argv[i + 1:i + 6]
        m = summary[tag]
        json.dump({"kernel": tag, "particles": int(particles), "side": int(side), "model": model, "cache_stress": True,
                   "hbm_bytes_per_launch": m["hbm_bytes_per_launch"], "hbm_read_bytes": m["hbm_read_bytes_corrected"],
                   "hbm_write_bytes": m["hbm_write_bytes"], "source": "tools/pmc.sh + tools/pmc_summary.py (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"},
                  open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
