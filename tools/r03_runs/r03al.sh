set -e
mkdir -p gpurun_out/r03al
out=gpurun_out/r03al/hwq.txt; : > $out
run() { q=$1; shift
  GPU_MAX_HW_QUEUES=$q timeout -k 10 240 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 "$@" > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('GPU_MAX_HW_QUEUES=$q', '$*', ':', d['value'])" >> $out
}
for shape in "4 1500000" "4 1250000" "4 1750000" "5 1200000" "5 1500000" "6 1000000" "6 1250000" "8 750000"; do set -- $shape; run 8 --config grid --fast --pools $1 --paths $2; done
run 16 --config grid --fast --pools 6 --paths 1000000
run 16 --config grid --fast --pools 8 --paths 750000
for shape in "4 1500000" "5 1200000" "6 1000000"; do set -- $shape; run 8 --config grid --pools $1 --paths $2; done
for shape in "4 1500000" "5 1200000" "6 1000000"; do set -- $shape; run 8 --config instanced --pools $1 --paths $2; done
run 4 --config grid --fast --pools 4 --paths 1500000
