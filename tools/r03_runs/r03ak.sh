set -e
mkdir -p gpurun_out/r03ak
out=gpurun_out/r03ak/hwq.txt; : > $out
run() { q=$1; shift
  GPU_MAX_HW_QUEUES=$q timeout -k 10 240 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 "$@" > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('GPU_MAX_HW_QUEUES=$q', '$*', ':', d['value'])" >> $out
}
run 4 --config sphere --pools 3 --paths 750000
run 8 --config sphere --pools 3 --paths 750000
run 8 --config sphere --pools 4 --paths 560000
run 8 --config sphere --pools 4 --paths 750000
run 8 --config sphere --pools 5 --paths 500000
run 8 --config sphere --pools 6 --paths 400000
run 8 --config grid --fast --pools 4 --paths 1500000
run 8 --config grid --fast --pools 4 --paths 2000000
run 8 --config grid --fast --pools 3 --paths 2000000
