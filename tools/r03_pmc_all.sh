# All counter passes this round's roofline blocks read: headline shape and exclusive one-pool shape, reference-exact and tolerance mode.
set -e
bash tools/pmc_passes.sh r3_sphere_3x750000 --config sphere --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_sphere_1x2000000 --config sphere --pools 1 --paths 2000000 --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_grid_1x2000000 --config grid --pools 1 --paths 2000000 --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_grid_3x2000000 --config grid --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_grid_1x2000000_fast --config grid --fast --check-period 0 --pools 1 --paths 2000000 --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_grid_3x2000000_fast --config grid --fast --steps 1 --warmup 1
