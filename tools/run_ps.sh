cd $GRAFT_REPO_ROOT
export ZS_ROCM_LIB=$GRAFT_REPO_ROOT/zpc_amd/lib/ablate/libzsrocm_NONE.so
timeout 300 python -m pytest tests/test_mpm_gpu.py -m gpu -x -q -k "fused_g2p2g_matches_reference_golden and sand-8" 2>&1 | tail -5
for v in 0 2 1; do echo "variant $v"; ZS_ROCM_G2P2G_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --checksum 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('checksum',[0]*8)[:6])"; done
