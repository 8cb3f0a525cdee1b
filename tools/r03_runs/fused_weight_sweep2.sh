#!/bin/bash
out=gpurun_out/fused_weight_sweep2.txt; : > $out
run() { # config steps weights
  r=$(PT_EXTEND_VARIANT=9 PT_SCHED_W=$3 timeout -k 10 160 python bench.py --config $1 --no-extras --no-cpu-baseline --steps $2 --warmup 1 2>>gpurun_out/fused_sweep.err | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])") || exit 1
  echo "$1 variant 9 PT_SCHED_W=$3 : $r" >> $out
}
for w in 6,30,5 6,60,5 6,400,5 4,30,5 4,30,8 5,24,6 3,30,5; do run sphere 5 $w || exit 1; done
for w in 4,9,5 4,18,5 4,30,5 4,18,8 3,18,5 4,12,8 5,12,8; do run instanced 3 $w || exit 1; done
for w in 3,18,5 3,30,5 3,400,5; do run grid 3 $w || exit 1; done
for w in 6,9,5 6,30,5 4,30,5; do run cornell 5 $w || exit 1; done
for w in 6,9,5 4,18,8 6,30,5; do run instanced_ext 3 $w || exit 1; done
