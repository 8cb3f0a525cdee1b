#!/bin/bash
out=gpurun_out/fused_weight_sweep3.txt; : > $out
run() { r=$(PT_SCHED_W=$3 timeout -k 10 160 python bench.py --config $1 --no-extras --no-cpu-baseline --steps $2 --warmup 1 2>>gpurun_out/fused_sweep.err | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])") || exit 1; echo "$1 PT_SCHED_W=$3 : $r" >> $out; }
for w in 6,30,5 6,30,4 6,30,7 7,30,5 5,30,5 6,24,5 6,40,5 7,36,6 6,30,5; do run sphere 5 $w || exit 1; done
for w in 4,12,8 4,12,10 4,10,8 3,12,8 5,14,8 4,14,10 4,12,6 4,12,8; do run instanced 3 $w || exit 1; done
for w in 4,12,8 4,12,10 5,14,8 6,30,5; do run instanced_ext 3 $w || exit 1; done
