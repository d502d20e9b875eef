cd /root/repo
python - <<'PY'
import numpy as np, torch, zpc_amd as zs
pol=zs.rocm_exec()
a=np.random.default_rng(0).integers(-2**30,2**30,100_000,dtype=np.int32)
d=torch.from_numpy(a).cuda(); out=torch.empty_like(d)
zs.radix_sort(pol,d,out)
o=out.cpu().numpy(); e=np.sort(a)
print('first call mismatches', (o!=e).sum(), 'err', zs.lib().zs_rocm_last_error(-1))
zs.radix_sort(pol,d,out); o=out.cpu().numpy(); print('second call mismatches', (o!=e).sum())
PY
for v in 0 1 2 3 4 7; do echo "== dbg $v"; ZS_ROCM_DEBUG=$v python bench.py --cells 64,128,64 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('p2g ms',j['roofline']['launch_ms'],'g2p ms',j['roofline']['g2p']['launch_ms'],'step ms',j['ms_per_step'])"; done
