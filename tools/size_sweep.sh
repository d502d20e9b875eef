for c in 128,512,128 128,256,128 128,128,128 64,512,64 64,256,64 64,128,64 256,128,256; do
python bench.py --no-cpu-baseline --cells $c --steps 20 --warmup 3 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', '%.3e'%d['value'], '%.3f'%d['ms_per_step'], '%.3f'%d['roofline']['launch_ms'], d['config']['particles'], d['config']['grid_blocks_rank0'])"
done
