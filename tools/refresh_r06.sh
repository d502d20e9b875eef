#!/bin/bash
# One GPU call that regenerates the measurements profiles/r06_* and profiles/pmc_*.json are written from (default bench = slotted storage,
# moving column).  Copy gpurun_out/r06/{pmc_g2p2g.json,pmc_p2g.json} to profiles/ by hand afterwards (only gpurun_out/ travels back).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r06; [ -z "$ONLY" ] && rm -rf $O; mkdir -p $O   # ONLY=p2g: just the stand-alone P2G passes (after a change of that kernel alone)
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-at-rest"
P2G="$B --compact --unfused --drift 0,0,0 --steps 8 --warmup 2"
# 1. kernel-trace stats of the default (moving) bench and of the unfused at-rest run
for tag in $([ "$ONLY" = p2g ] && echo unfused || echo moving unfused); do
  cmd="$B --steps 10 --warmup 3"; [ $tag = unfused ] && cmd="$P2G"
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_$tag -o r -- $cmd > $O/stats_${tag}_bench.json 2> $O/stats_${tag}_stderr.txt
  db=$(find $O/stats_$tag -name '*.db' | head -1)
  python $R/tools/rocpd_stats.py "$db" $O/kernel_stats_$tag.md > /dev/null
  rm -rf $O/stats_$tag
done
# 2. PMC passes, one counter group per run: the fused step's kernels (moving) and the stand-alone P2G
pmc() {  # pmc <name> <kernel regex> <command...>
  name=$1; rx=$2; shift 2
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
    g=$(echo $grp | tr ' ' '_' | cut -c1-40)
    out=$O/pmc_$name/$g; mkdir -p $out
    timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "$rx" --pmc $grp --output-format csv -d $out -o pmc -- "$@" > $out/bench.json 2> $out/stderr.txt
    find $out -name '*.csv' -size +8M -delete
  done
}
if [ "$ONLY" != p2g ]; then
pmc fused "g2p2g_slot|slot_rehome_kernel|slot_commit_kernel" $B --steps 20 --warmup 2
pmc fusedrest "g2p2g_slot|slot_rehome_kernel|slot_commit_kernel" $B --steps 10 --warmup 2 --drift 0,0,0
fi
pmc p2g "p2g_tile_kernel" $P2G
python3 - $O $R <<'PY'
import csv, glob, os, sys, collections, json
O, R = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(R, "tools"))
import kernel_hash
def collect(name, keys):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(os.path.join(O, "pmc_" + name, "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((k for k in keys if k in r["Kernel_Name"]), None)
            if k: acc[k][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])   # (rows of one dispatch are summed)
    acc = {k: {c: [per[d] for d in sorted(per)] for c, per in cs.items()} for k, cs in acc.items()}
    if not acc:   # (ONLY=p2g on a fresh box: no passes of this group here -- leave earlier summaries alone)
        return {}
    summ, lines = {}, []
    for k in acc:
        m = {c: sum(v[-3:]) / len(v[-3:]) for c, v in acc[k].items()}   # the last launches (steady state)
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            m["hbm_read_bytes_corrected"] = m["FETCH_SIZE"] * 1024 * 2   # gfx950: 128-B requests tallied as 64 B (MI355X_MICROARCH.md, HBM)
            m["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
            m["hbm_bytes_per_launch"] = m["hbm_read_bytes_corrected"] + m["hbm_write_bytes"]
        summ[k] = m
        lines += ["## %s" % k, "| counter | value |", "|---|---|"] + ["| %s | %.6g |" % (c, m[c]) for c in sorted(m)] + [""]
    open(os.path.join(O, "pmc_%s.md" % name), "w").write("\n".join(lines) + "\n")
    return summ
KS = ("g2p2g_slotblk_kernel", "slot_rehome_kernel", "slot_commit_kernel")
s = collect("fused", KS)
r = collect("fusedrest", KS)
main = s.get("g2p2g_slotblk_kernel", {})
if "hbm_bytes_per_launch" in main:
    rx = r"g2p2g_slotblk_kernel<1, false>"
    obj = "zpc_amd/lib/obj/mpm_slotblk.o"
    tot = sum(s.get(k, {}).get("hbm_bytes_per_launch", 0.0) for k in KS)
    valu = sum(s.get(k, {}).get("SQ_INSTS_VALU", 0.0) for k in KS)
    valu_rest = sum(r.get(k, {}).get("SQ_INSTS_VALU", 0.0) for k in KS) or None
    # shader cycles one SIMD spent per VALU instruction of the main kernel (GRBM_GUI_ACTIVE counts every XCD: / 8; 1024 SIMDs)
    cpi = (main["GRBM_GUI_ACTIVE"] / 8 * 1024) / main["SQ_INSTS_VALU"] if "GRBM_GUI_ACTIVE" in main and main.get("SQ_INSTS_VALU") else None
    json.dump({"kernel": "g2p2g_slotblk_kernel + slot_rehome_kernel + slot_commit_kernel", "particles": 67108864, "side": 8, "model": "sand", "cache_stress": True,
               "hbm_bytes_per_launch": tot, "main_kernel_bytes": main["hbm_bytes_per_launch"],
               "hbm_read_bytes": sum(s.get(k, {}).get("hbm_read_bytes_corrected", 0.0) for k in s), "hbm_write_bytes": sum(s.get(k, {}).get("hbm_write_bytes", 0.0) for k in s),
               "valu_insts_per_launch": valu, "valu_insts_at_rest": valu_rest, "valu_cycles_per_inst_per_simd": cpi,
               "sq_wave_cycles": main.get("SQ_WAVE_CYCLES"), "sq_wait_any": main.get("SQ_WAIT_ANY"), "grbm_gui_active": main.get("GRBM_GUI_ACTIVE"),
               "code_object": obj, "code_regex": rx, "code_hash": kernel_hash.combined(os.path.join(R, obj), rx),
               "source": "tools/refresh_r06.sh (rocprofv3 --pmc, separate passes per counter group; FETCH_SIZE x 2 on gfx950; valu_cycles_per_inst_per_simd = GRBM_GUI_ACTIVE / 8 x 1024 SIMDs / SQ_INSTS_VALU of the main kernel; valu_insts_at_rest = the same kernels on the column at rest)"},
              open(os.path.join(O, "pmc_g2p2g.json"), "w"), indent=1)
s = collect("p2g", ("p2g_tile_kernel",))
if "hbm_bytes_per_launch" in s.get("p2g_tile_kernel", {}):
    m = s["p2g_tile_kernel"]
    rx = r"p2g_tile_kernel<8, 3, 2, true>"
    json.dump({"kernel": "p2g_tile", "particles": 67108864, "side": 8, "model": "sand", "cache_stress": True, "hbm_bytes_per_launch": m["hbm_bytes_per_launch"],
               "hbm_read_bytes": m["hbm_read_bytes_corrected"], "hbm_write_bytes": m["hbm_write_bytes"],
               "code_object": "zpc_amd/lib/obj/mpm_p2g.o", "code_regex": rx, "code_hash": kernel_hash.combined(os.path.join(R, "zpc_amd/lib/obj/mpm_p2g.o"), rx),
               "source": "tools/refresh_r06.sh"}, open(os.path.join(O, "pmc_p2g.json"), "w"), indent=1)
PY
cd $R
[ "$ONLY" = p2g ] && { $B --compact --unfused --drift 0,0,0 > $O/bench_n1_unfused_at_rest.json 2>/dev/null; exit 0; }
# 3. bench lines
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
$B --drift 0,0,0 > $O/bench_n1_at_rest.json 2>/dev/null
$B --compact --drift 0,0,0 > $O/bench_n1_compact_at_rest.json 2>/dev/null
$B --compact --unfused --drift 0,0,0 > $O/bench_n1_unfused_at_rest.json 2>/dev/null
$B --cells 100,100,100 --model jello --grid 256 > $O/bench_config3_jello_8M.json 2>/dev/null
$B --steps 40 --warmup 5 --cells 64,256,64 > $O/eighth_plain.json 2>/dev/null
$B --steps 40 --warmup 5 --cells 64,256,64 --rank-proxy 8 2>/dev/null | grep '^{' > $O/proxy8.json
$B --steps 3000 --warmup 3 --slot-stats 2>/dev/null | grep '^{' > $O/bench_soak3000.json
timeout 600 python tools/bench_prims.py --json $O/prims.json > $O/prims.txt 2>&1
tail -c 400 $O/bench_n1.json
