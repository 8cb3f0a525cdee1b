#!/bin/bash
# weights of the schedule with the fused inner+triangle pass (extend variant 9): PT_SCHED_W="inner,triangle,switch"
out=gpurun_out/fused_weight_sweep.txt; : > $out
run() { # config steps weights
  r=$(PT_EXTEND_VARIANT=9 PT_SCHED_W=$3 timeout -k 10 160 python bench.py --config $1 --no-extras --no-cpu-baseline --steps $2 --warmup 1 2>>gpurun_out/fused_sweep.err | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])") || exit 1
  echo "$1 variant 9 PT_SCHED_W=$3 : $r" >> $out
}
for w in 6,9,5 6,6,5 6,12,5 6,18,5 6,30,5 4,9,5 8,9,5 6,9,9; do run sphere 5 $w || exit 1; done
for w in 6,9,5 6,12,5 6,18,5 4,9,5 6,9,9; do run instanced 3 $w || exit 1; done
for w in 3,9,5 3,6,5 3,12,5 6,9,5 2,9,5 3,4,5; do run grid 3 $w || exit 1; done
