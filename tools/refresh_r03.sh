#!/bin/bash
# One GPU call that regenerates the measurements profiles/r03_* are written from (default bench = slotted storage, moving column).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r03; rm -rf $O; mkdir -p $O
# 1. kernel-trace stats: default (moving) and at rest
for tag in moving rest; do
  extra=""; [ $tag = rest ] && extra="--drift 0,0,0"
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_$tag -o r -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-at-rest $extra > $O/stats_${tag}_bench.json 2> $O/stats_${tag}_stderr.txt
  db=$(find $O/stats_$tag -name '*.db' | head -1)
  python $R/tools/rocpd_stats.py "$db" $O/kernel_stats_$tag.md > /dev/null
  rm -rf $O/stats_$tag
done
# 2. PMC passes on the fused step's kernels (moving): one counter group per run
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  out=$O/pmc/$name; mkdir -p $out
  timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "g2p2g_slot_kernel|slot_rehome_kernel|slot_commit_kernel" --pmc $grp --output-format csv -d $out -o pmc -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-at-rest > $out/bench.json 2> $out/stderr.txt
  find $out -name '*.csv' -size +8M -delete
done
python3 - $O <<'PY'
import csv, glob, os, sys, collections, json
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(O, "pmc", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = "g2p2g_slot_kernel" if "g2p2g_slot" in r["Kernel_Name"] else ("slot_rehome_kernel" if "slot_rehome" in r["Kernel_Name"] else ("slot_commit_kernel" if "slot_commit" in r["Kernel_Name"] else None))
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
summ = {}
for k in acc:
    m = {c: sum(v[-3:]) / len(v[-3:]) for c, v in acc[k].items()}   # the last launches (steady state of the moving column)
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        m["hbm_read_bytes_corrected"] = m["FETCH_SIZE"] * 1024 * 2   # gfx950: 128-B requests tallied as 64 B (MI355X_MICROARCH.md, HBM)
        m["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
        m["hbm_bytes_per_launch"] = m["hbm_read_bytes_corrected"] + m["hbm_write_bytes"]
    summ[k] = m
    lines += ["## %s" % k, "| counter | value |", "|---|---|"] + ["| %s | %.6g |" % (c, m[c]) for c in sorted(m)] + [""]
open(os.path.join(O, "pmc_fused.md"), "w").write("\n".join(lines) + "\n")
if "g2p2g_slot_kernel" in summ and "hbm_bytes_per_launch" in summ["g2p2g_slot_kernel"]:
    others = ("slot_rehome_kernel", "slot_commit_kernel")
    tot = summ["g2p2g_slot_kernel"]["hbm_bytes_per_launch"] + sum(summ.get(k, {}).get("hbm_bytes_per_launch", 0.0) for k in others)
    json.dump({"kernel": "g2p2g_slot_kernel + slot_rehome_kernel + slot_commit_kernel", "particles": 67108864, "side": 8, "model": "sand", "cache_stress": True,
               "hbm_bytes_per_launch": tot, "main_kernel_bytes": summ["g2p2g_slot_kernel"]["hbm_bytes_per_launch"],
               "rehome_kernel_bytes": summ.get("slot_rehome_kernel", {}).get("hbm_bytes_per_launch"),
               "commit_kernel_bytes": summ.get("slot_commit_kernel", {}).get("hbm_bytes_per_launch"),
               "source": "tools/refresh_r03.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 read correction x2)"},
              open(os.path.join(O, "pmc_g2p2g.json"), "w"), indent=1)
PY
# (copy gpurun_out/r03/pmc_g2p2g.json to profiles/pmc_g2p2g.json by hand: only gpurun_out/ travels back from the GPU box)
cd $R
# 3. bench lines
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --drift 0,0,0 --no-cpu-baseline > $O/bench_n1_at_rest.json 2>/dev/null
python bench.py --compact --drift 0,0,0 --no-cpu-baseline > $O/bench_n1_compact_at_rest.json 2>/dev/null
python bench.py --compact --unfused --drift 0,0,0 --no-cpu-baseline > $O/bench_n1_unfused_at_rest.json 2>/dev/null
python bench.py --cells 100,100,100 --model jello --grid 256 --no-cpu-baseline --no-at-rest > $O/bench_config3_jello_8M.json 2>/dev/null
python tools/bench_prims.py --json $O/prims.json > $O/prims.txt 2>&1
tail -c 400 $O/bench_n1.json
