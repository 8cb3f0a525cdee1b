#!/bin/bash
out=gpurun_out/staged_fused_check.txt; : > $out
run() { r=$(timeout -k 10 160 python bench.py --config $2 --no-extras --no-cpu-baseline --steps $3 --warmup 2 2>>gpurun_out/staged_fused_check.err | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])") || exit 1; echo "$2 $1 : $r" >> $out; }
PT_STAGED_FUSED=1 timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cornell or variant or staged" --timeout 120 --timeout-method thread > gpurun_out/staged_fused_tests.log 2>&1 || exit 1
for i in 1 2; do
  run "staged, separate passes" cornell 8 || exit 1
  PT_STAGED_FUSED=1 run "staged, fused pass" cornell 8 || exit 1
  PT_STAGED_FUSED=1 PT_SCHED_W=6,30,5 run "staged, fused pass, PT_SCHED_W=6,30,5" cornell 8 || exit 1
  PT_STAGED_FUSED=1 PT_SCHED_W=6,15,5 run "staged, fused pass, PT_SCHED_W=6,15,5" cornell 8 || exit 1
done
