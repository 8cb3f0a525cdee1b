set -e
mkdir -p gpurun_out/r03z
out=gpurun_out/r03z/sweep.txt; : > $out
run() { tag=$1; shift
  timeout -k 10 240 python bench.py --config grid --fast --no-extras --no-cpu-baseline --steps 3 --warmup 1 "$@" > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('grid tolerance mode', '$tag', ':', d['value'])" >> $out
}
for shape in "3 2000000" "3 2500000" "4 2000000" "2 3000000" "3 3000000" "4 1500000"; do set -- $shape; run "pools $1 x $2" --pools $1 --paths $2; done
for w in 3,9,5,2 3,9,5,3 3,9,5,4 3,12,5,3 3,7,5,3 3,9,8,3; do PT_SCHED_W=$w run "weights $w"; done
