#!/bin/bash
out=gpurun_out/fused_default_check.txt; : > $out
run() { # label config steps  (env passed by caller)
  r=$(timeout -k 10 160 python bench.py --config $2 --no-extras --no-cpu-baseline --steps $3 --warmup 1 2>>gpurun_out/fused_default_check.err | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])") || exit 1
  echo "$2 $1 : $r" >> $out
}
for i in 1 2; do
  run default sphere 5 || exit 1
  PT_SCHED_W=6,30,5 run "PT_SCHED_W=6,30,5" sphere 5 || exit 1
  PT_EXTEND_VARIANT=4 run "variant 4" sphere 5 || exit 1
  run default cornell 8 || exit 1
  PT_EXTEND_VARIANT=4 run "variant 4" cornell 8 || exit 1
done
