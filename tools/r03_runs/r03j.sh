set -e
mkdir -p gpurun_out/r03j
out=gpurun_out/r03j/shapes.txt; : > $out
run() { cfg=$1; mode=$2; pools=$3; paths=$4
  timeout -k 10 240 python bench.py --config $cfg $mode --no-extras --no-cpu-baseline --steps 3 --warmup 1 --check-period 0 --pools $pools --paths $paths > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('$cfg $mode pools $pools x $paths :', d['value'], d['wavefront_iterations_per_render'])" >> $out
}
for rep in 1 2; do
for shape in "3 750000" "2 1250000" "2 1500000" "2 1750000" "2 2000000" "2 2500000" "2 3000000" "3 1500000" "3 2000000" "4 1500000"; do run grid --fast $shape; done
done
for shape in "2 2000000" "2 3000000" "3 2000000"; do run grid "" $shape; done
for shape in "2 2000000" "2 3000000" "3 2000000"; do run instanced "" $shape; done
