#!/bin/bash
# r04 second GPU pass: one-call step A/B, rank proxy, soaks, new full-size tests
O=gpurun_out/r04b; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --no-at-rest --no-cpu-baseline"
timeout 900 python -m pytest tests/test_mpm_gpu.py -x -q -m gpu -k "slotted" > $O/t_slotted.log 2>&1; echo "slotted rc=$?" >> $O/summary.txt
timeout 600 $B --steps 20 --warmup 3 > $O/bench20_onecall.json 2> $O/bench20_onecall.err; echo "onecall rc=$?" >> $O/summary.txt
timeout 600 $B --steps 20 --warmup 3 --py-step > $O/bench20_pystep.json 2> $O/bench20_pystep.err; echo "pystep rc=$?" >> $O/summary.txt
timeout 600 $B --steps 40 --warmup 5 --cells 64,256,64 > $O/eighth_plain.json 2> $O/eighth_plain.err; echo "eighth_plain rc=$?" >> $O/summary.txt
timeout 600 $B --steps 40 --warmup 5 --cells 64,256,64 --rank-proxy 8 > $O/proxy8_onecall.json 2> $O/proxy8_onecall.err; echo "proxy8 rc=$?" >> $O/summary.txt
timeout 600 $B --steps 40 --warmup 5 --cells 64,256,64 --rank-proxy 8 --py-step > $O/proxy8_pystep.json 2> $O/proxy8_pystep.err; echo "proxy8py rc=$?" >> $O/summary.txt
timeout 600 $B --steps 40 --warmup 5 --cells 64,256,64 --rank-proxy 8 --no-overlap > $O/proxy8_nooverlap.json 2> $O/proxy8_nooverlap.err; echo "proxy8noov rc=$?" >> $O/summary.txt
timeout 900 $B --steps 3000 --warmup 0 --checksum > $O/soak3000_closed.json 2> $O/soak3000_closed.err; echo "soak closed rc=$?" >> $O/summary.txt
timeout 900 $B --steps 3000 --warmup 0 --checksum --repartition open > $O/soak3000_open.json 2> $O/soak3000_open.err; echo "soak open rc=$?" >> $O/summary.txt
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "closed_loop or identities" > $O/t_fullsize_new.log 2>&1; echo "fullsize new rc=$?" >> $O/summary.txt
cat $O/summary.txt
