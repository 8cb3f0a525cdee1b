set -e
mkdir -p gpurun_out/r03s
out=gpurun_out/r03s/check_cost.txt; : > $out
for p in 0 1048576 4096 1024 256 64; do
  timeout -k 10 240 python bench.py --config grid --fast --check-period $p --no-extras --no-cpu-baseline --steps 3 --warmup 1 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('grid tolerance mode, check period $p :', d['value'], 'Mrays/s; re-walked', d['tolerance_check']['rays_rewalked_exactly'])" >> $out
done
