#!/bin/bash
out=gpurun_out/pool_size_fine.txt; : > $out
run() { r=$(timeout -k 10 200 python bench.py --config $1 --pools $2 --paths $3 --no-extras --no-cpu-baseline --steps $4 --warmup 1 2>>gpurun_out/pool_size_fine.err | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])") || exit 1; echo "$1 pools $2 x $3 : $r" >> $out; }
for p in 650000 691200 720000 750000 780000 820000 750000; do run sphere 3 $p 5 || exit 1; done
for p in 1300000 1400000 1500000 1600000 1800000; do run instanced 4 $p 3 || exit 1; done
