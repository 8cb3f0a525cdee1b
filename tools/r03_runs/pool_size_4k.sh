#!/bin/bash
out=gpurun_out/pool_size_4k.txt; : > $out
run() { r=$(timeout -k 10 200 python bench.py --config $1 --pools $2 --paths $3 $5 --no-extras --no-cpu-baseline --steps $4 --warmup 1 2>>gpurun_out/pool_size_4k.err | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])") || exit 1; echo "$1 $5 pools $2 x $3 : $r" >> $out; }
for p in 1500000 1800000 2000000 2400000; do run instanced 4 $p 3 || exit 1; done
for p in 1500000 1800000 2000000 2400000; do run grid 4 $p 3 || exit 1; done
for p in 1500000 1800000 2000000 2400000; do run grid 4 $p 3 --fast || exit 1; done
for p in 1500000 2000000; do run instanced_ext 4 $p 3 || exit 1; done
