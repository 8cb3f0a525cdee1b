set -e
mkdir -p gpurun_out/r03d
out=gpurun_out/r03d/ab.txt; : > $out
for lib in w8 w7 w8_noilp w7_noilp; do
  for p in 0 1048576 64; do
    PT_AMD_LIB=$PWD/ab/lib_$lib.so PT_SCHED_W=3,9,5,3 timeout -k 10 240 python bench.py --config grid --fast --no-extras --no-cpu-baseline --steps 2 --warmup 1 --check-period $p > /tmp/o.json 2> /tmp/o.err
    python3 -c "import json;d=json.load(open('/tmp/o.json'));print('grid', '$lib', 'period $p', ':', d['value'], d['tolerance_check']['rays_rewalked_exactly'])" >> $out
  done
done
