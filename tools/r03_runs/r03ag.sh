set -e
mkdir -p gpurun_out/r03ag
PT_COMPACT_SHAPE=3 timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_exact_step or pool_sizes or pools_on or full_render or spp8" --timeout 300 --timeout-method thread > gpurun_out/r03ag/tests_shape3.txt 2>&1
out=gpurun_out/r03ag/compact_shape.txt; : > $out
for cfg in sphere cornell instanced "grid --fast"; do for shape in 0 3; do
  PT_COMPACT_SHAPE=$shape timeout -k 10 240 python bench.py --config $cfg --no-extras --no-cpu-baseline --steps 3 --warmup 1 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('$cfg', 'PT_COMPACT_SHAPE=$shape', ':', d['value'])" >> $out
done; done
