#!/bin/bash
out=gpurun_out/fused3_check.txt; : > $out
run() { r=$(timeout -k 10 160 python bench.py --config $2 --no-extras --no-cpu-baseline --steps $3 --warmup 1 2>>gpurun_out/fused3_check.err | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])") || exit 1; echo "$2 $1 : $r" >> $out; }
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variant" --timeout 120 --timeout-method thread > gpurun_out/fused3_tests.log 2>&1 || exit 1
for i in 1 2; do
  run default sphere 5 || exit 1
  run default instanced 3 || exit 1
  PT_SCHED_W=4,12,5 run "PT_SCHED_W=4,12,5" instanced 3 || exit 1
  PT_SCHED_W=6,30,3 run "PT_SCHED_W=6,30,3" sphere 5 || exit 1
done
run default instanced_ext 3 || exit 1
