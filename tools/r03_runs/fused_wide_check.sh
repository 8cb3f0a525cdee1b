#!/bin/bash
out=gpurun_out/fused_wide_check.txt; : > $out
run() { r=$(timeout -k 10 200 python bench.py --config $2 --fast --no-extras --no-cpu-baseline --steps 3 --warmup 1 2>>gpurun_out/fused_wide_check.err | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); c=d['tolerance_check']
print(d['value'], 'rewalked', c['rays_rewalked_exactly'], 'different', c['different_hit'], 'beyond', c['beyond_1e-6'], 'lost', c['of_those_reference_hit_lost'])") || exit 1; echo "$2 --fast $1 : $r" >> $out; }
PT_EXTEND_VARIANT=9 timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tolerance" --timeout 120 --timeout-method thread > gpurun_out/fused_wide_tests.log 2>&1 || exit 1
for i in 1 2; do
  PT_EXTEND_VARIANT=4 run "variant 4" grid || exit 1
  PT_EXTEND_VARIANT=9 run "variant 9" grid || exit 1
done
PT_EXTEND_VARIANT=9 PT_SCHED_W=3,18,5,3 run "variant 9 PT_SCHED_W=3,18,5,3" grid || exit 1
PT_EXTEND_VARIANT=9 PT_SCHED_W=3,30,5,3 run "variant 9 PT_SCHED_W=3,30,5,3" grid || exit 1
PT_EXTEND_VARIANT=9 PT_SCHED_W=3,9,5,2 run "variant 9 PT_SCHED_W=3,9,5,2" grid || exit 1
PT_EXTEND_VARIANT=4 run "variant 4" instanced || exit 1
PT_EXTEND_VARIANT=9 run "variant 9" instanced || exit 1
