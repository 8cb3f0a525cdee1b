#!/bin/bash
out=gpurun_out/pool_shape_recheck.txt; : > $out
run() { r=$(timeout -k 10 200 python bench.py --config $1 --pools $2 --paths $3 --no-extras --no-cpu-baseline --steps $4 --warmup 1 2>>gpurun_out/pool_shape_recheck.err | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])") || exit 1; echo "$1 pools $2 x $3 : $r" >> $out; }
run sphere 3 750000 5 || exit 1
run sphere 3 600000 5 || exit 1
run sphere 3 900000 5 || exit 1
run sphere 4 600000 5 || exit 1
run sphere 4 750000 5 || exit 1
run sphere 2 1000000 5 || exit 1
run instanced 4 1500000 3 || exit 1
run instanced 3 2000000 3 || exit 1
run instanced 4 1200000 3 || exit 1
run instanced 5 1200000 3 || exit 1
