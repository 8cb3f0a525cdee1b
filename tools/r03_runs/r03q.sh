set -e
mkdir -p gpurun_out/r03q
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_extend_variant" --timeout 300 --timeout-method thread > gpurun_out/r03q/tests.txt 2>&1
for cfg in sphere grid; do timeout -k 10 300 python tools/extend_ab.py --config $cfg --variants 4,7 --paths 2000000 --warm-steps 14 > gpurun_out/r03q/extend_ab_soa_$cfg.txt 2>&1; done
out=gpurun_out/r03q/bench_ab_soa.txt; : > $out
for cfg in sphere grid instanced; do for v in 4 7; do
  PT_EXTEND_VARIANT=$v timeout -k 10 240 python bench.py --config $cfg --no-extras --no-cpu-baseline --steps 3 --warmup 1 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('$cfg', 'extend variant $v', ':', d['value'], 'Mrays/s')" >> $out
done; done
