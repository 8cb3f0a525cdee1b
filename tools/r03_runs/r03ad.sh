set -e
bash tools/pmc_passes.sh r3_sphere_3x750000 --config sphere --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_sphere_1x2000000 --config sphere --pools 1 --paths 2000000 --steps 1 --warmup 1
