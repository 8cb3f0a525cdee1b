set -e
mkdir -p gpurun_out/r03ao
PT_XCD_CHUNK=16 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_exact_step or pool_sizes or pools_on or full_render or tolerance" --timeout 300 --timeout-method thread > gpurun_out/r03ao/tests_chunk16.txt 2>&1
out=gpurun_out/r03ao/xcd.txt; : > $out
for cfg in "grid" "grid --fast" sphere instanced; do for c in 0 2 8 32 128; do
  PT_XCD_CHUNK=$c timeout -k 10 240 python bench.py --config $cfg --no-extras --no-cpu-baseline --steps 3 --warmup 1 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('$cfg', 'PT_XCD_CHUNK=$c', ':', d['value'])" >> $out
done; done
