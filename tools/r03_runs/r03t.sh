set -e
mkdir -p gpurun_out/r03t
out=gpurun_out/r03t/ab3.txt; : > $out
for lib in base sgpr96; do for p in 0 256; do
  PT_AMD_LIB=$PWD/ab/lib_$lib.so timeout -k 10 240 python bench.py --config grid --fast --check-period $p --no-extras --no-cpu-baseline --steps 3 --warmup 1 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('$lib period $p :', d['value'])" >> $out
done; done
