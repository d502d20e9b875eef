#!/bin/bash
# PMC passes (one counter group per run, kernel trace only) for the gather-style transfer kernels: tools/bench_c2.py --lattice-only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/pmc_c2; rm -rf $O; mkdir -p $O
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  out=$O/$name; mkdir -p $out
  timeout -s KILL 240 rocprofv3 --kernel-trace --kernel-include-regex "p2c2g_cell|p2c2g_node|c2_particle|c2_octant|g2c2p_" --pmc $grp --output-format csv -d $out -o pmc -- python $R/tools/bench_c2.py --lattice-only > $out/out.json 2> $out/stderr.txt
done
python3 - $O <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
TAGS = ("p2c2g_cell8_kernel<8, 0", "p2c2g_node_kernel<8, 0>", "c2_particle_kernel<0, 0>", "c2_octant", "g2c2p_cell", "g2c2p_particle_kernel<8, true")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        for t in TAGS:
            if t in k:
                acc[t][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
lines = ["# r01 PMC counters -- gather-style transfers, tools/pmc_c2.sh (tools/bench_c2.py --lattice-only: 16 777 216 particles, 256^3 grid,", "# 8^3 blocks), per launch; FETCH_SIZE in KiB with the gfx950 x2 read correction applied in hbm_read_bytes_corrected", ""]
for k in TAGS:
    if k not in acc: continue
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        m["hbm_read_bytes_corrected"] = m["FETCH_SIZE"] * 1024 * 2
        m["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
    if "SQ_INSTS_VALU" in m and "GRBM_GUI_ACTIVE" in m:
        m["valu_busy_frac (SQ_INSTS_VALU x 4 cyc / SIMD cycles)"] = m["SQ_INSTS_VALU"] * 4 / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        m["l2_hit_rate"] = m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    lines += ["## %s" % k, "| counter | value |", "|---|---|"] + ["| %s | %.6g |" % (c, m[c]) for c in sorted(m)] + [""]
open(os.path.join(root, "summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $O -name '*.csv' -delete; find $O -name '*.db' -delete
