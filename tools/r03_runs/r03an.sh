set -e
bash tools/pmc_passes.sh r3_grid_4x1500000 --config grid --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_grid_4x1500000_fast --config grid --fast --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_instanced_4x1500000 --config instanced --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_instanced_ext_4x1500000 --config instanced_ext --steps 1 --warmup 1
