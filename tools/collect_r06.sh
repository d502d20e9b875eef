#!/bin/bash
# after tools/r06_final.sh (one gpurun call): copy what the round quotes from gpurun_out/r06/ into profiles/ (tracked)
cd "$(dirname "$0")/.."; O=gpurun_out/r06
{ echo "# r06 — rocprofv3 --kernel-trace --stats (tools/refresh_r06.sh, final build)"; echo; echo "## default bench (64 Mi-particle column falling at 0.051 cell/step, slotted storage, one workgroup per 8^3 block)"; echo; grep -v "^#" $O/kernel_stats_moving.md | sed '/^$/d'; echo; echo "## unfused step, column at rest, compact storage (\`--compact --unfused --drift 0,0,0\`, 2 warm-up + 8 timed steps): the stand-alone P2G (\`p2g_tile_kernel\`) and G2P (\`g2p_packed_kernel\`)"; echo; grep -v "^#" $O/kernel_stats_unfused.md | sed '/^$/d'; } > profiles/r06_kernel_stats.md
{ echo "# r06 — PMC counters of the fused step's kernels, moving column (rocprofv3 --pmc, one counter group per run; last three launches; tools/refresh_r06.sh, final build)"; echo; cat $O/pmc_fused.md; echo "# the same kernels, column at rest"; echo; cat $O/pmc_fusedrest.md; } > profiles/r06_pmc_g2p2g.md
{ echo "# r06 — PMC counters of the stand-alone P2G (p2g_tile_kernel<8, 3, 2, true>, compact storage, column at rest; tools/refresh_r06.sh on the final build)"; echo; cat $O/pmc_p2g.md; } > profiles/r06_pmc_p2g.md
for f in n1 n1_at_rest n1_compact_at_rest n1_unfused_at_rest config3_jello_8M soak3000; do cp $O/bench_$f.json profiles/r06_bench_$f.json; done
cp $O/eighth_plain.json profiles/r06_bench_eighth_column.json; cp $O/proxy8.json profiles/r06_bench_rank_proxy8.json; cp $O/proxy8_in_turn.json profiles/r06_bench_rank_proxy8_in_turn.json
cp $O/prims.json profiles/r06_prims.json; cp $O/pmc_p2g.json profiles/pmc_p2g.json; cp $O/pmc_g2p2g.json profiles/pmc_g2p2g.json; cp gpurun_out/r06_gputest.log profiles/r06_gputest.log
python3 - <<'PY'
import sys, json; sys.path.insert(0, "tools"); import kernel_hash
for f in ("profiles/pmc_p2g.json", "profiles/pmc_g2p2g.json"):
    d = json.load(open(f)); print(f, "hash of the measured build", d["code_hash"], "| of this tree", kernel_hash.combined(d["code_object"], d["code_regex"]))
PY
